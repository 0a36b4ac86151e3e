// sweep_mixed.inc.hip -- part of qmcp_kernels.hip (one translation unit; included inside namespace qmcp).
// ------------------------------------------------------------------ general (mixed-span) sweep
// Event-driven form of the canonical rule for arbitrary spans.  Reads are bucketed by start
// and ordered (end desc, index asc) inside a bucket, so the selected reads of a bucket are
// always a prefix; the pool of candidates at position p is the set of bucket heads of the
// last max_span start positions, compared by (end desc, start desc).
// One wave per contig.  Two rings of `ring_size` (power of two > max_span) entries live in
// LDS: the prefix pointer of every bucket still inside the window, and the number of
// selected reads by end position (what stops covering when the sweep passes that end).
// A bucket's pointer is flushed to selend (as an absolute offset) when its slot is recycled.
struct SortedRec { const Rec* r; __device__ uint64_t key(uint32_t j) const { return r[j].key; } };
struct SortedK64 { const uint64_t* k; __device__ uint64_t key(uint32_t j) const { return k[j]; } };

// A stretch that starts behind a cut point (sweep_segments): every read that covers the position
// before its first one is kept, whatever came earlier, and none of them is a candidate any more.
// What the stretch inherits is therefore only their coverage: this returns how many of them reach
// into the stretch and adds them to the expiry ring by end position (ring index = end - a, modulo;
// the ring holds at least max_span entries).  One wave; reads starting in the max_span - 1 positions
// before `a` are looked at (global prefix counts: a previous contig's reads end before `a`).
template <typename Sorted>
__device__ __forceinline__ uint32_t seed_stretch_expiry(Sorted skeys, const uint32_t* __restrict__ boff,
                                                        uint32_t a, uint32_t span_bits, uint32_t max_span,
                                                        uint32_t* s_exp, uint32_t ring, uint32_t lane) {
    const uint64_t code_mask = (1ull << span_bits) - 1;
    const uint32_t lo = a >= max_span ? a - (max_span - 1) : 0u;
    uint32_t count = 0;
    for (uint32_t j = boff[lo] + lane; j < boff[a]; j += 64) {
        const uint64_t kk = skeys.key(j);
        const uint32_t end = (uint32_t)(kk >> span_bits) + (max_span - (uint32_t)(kk & code_mask)) - 1;
        if (end >= a) {
            atomicAdd(&s_exp[(end - a) % ring], 1u);
            ++count;
        }
    }
    return wave_sum_u32(count);
}

// GRING: the two rings live in global memory (`g_rings`, 2 * ring_size words per workgroup) instead
// of LDS -- spans beyond 16 383 (long reads); every ring access is then a write-through / L1-bypassing
// access, ordered by the wave's program order and the workgroup barriers as the LDS ones are.
template <typename Sorted, bool GRING>
__global__ __launch_bounds__(64) void k_sweep_general(const uint32_t* __restrict__ boff,
                                                      const uint32_t* __restrict__ eoff,
                                                      Sorted skeys,
                                                      const uint64_t* __restrict__ contig_pos_off,
                                                      uint32_t span_bits, uint32_t max_span,
                                                      uint32_t M, uint32_t* __restrict__ selend,
                                                      uint32_t ring_size, const uint32_t* __restrict__ seg,
                                                      uint32_t* g_rings) {
    extern __shared__ uint32_t s_ring[];
    uint32_t* const ring = GRING ? g_rings + (size_t)blockIdx.x * 2u * ring_size : s_ring;
    uint32_t* s_ptr = ring;              // [ring_size] bucket prefix pointers
    uint32_t* s_exp = ring + ring_size;  // [ring_size] selected reads by end position
    auto rd = [&](const uint32_t* a, uint32_t i) -> uint32_t {
        if constexpr (GRING) return __hip_atomic_load(a + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else return a[i];
    };
    auto wr = [&](uint32_t* a, uint32_t i, uint32_t v) {
        if constexpr (GRING) __hip_atomic_store(a + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else a[i] = v;
    };
    const uint32_t lane = threadIdx.x;
    const uint32_t c_id = blockIdx.x;
    SweepSeg sg;
    if (!sweep_segment(contig_pos_off, seg, c_id, sg)) return;
    const uint32_t base = sg.base, L = sg.Lrun;  // a stretch never looks past its own end
    const uint32_t rmask = ring_size - 1;
    const uint64_t code_mask = (1ull << span_bits) - 1;
    for (uint32_t i = lane; i < 2 * ring_size; i += 64) wr(ring, i, 0u);
    __syncthreads();
    uint32_t cur = 0;
    if (seg != nullptr) {
        cur = seed_stretch_expiry(skeys, boff, base, span_bits, max_span, s_exp, ring_size, lane);
        __syncthreads();
    }
    for (uint32_t p = 0; p < L; ++p) {
        const uint32_t gp = base + p;
        if (lane == 0) {
            if (p >= ring_size) {
                const uint32_t q = p - ring_size;  // long dead: ring_size > max_span
                selend[base + q] = boff[base + q] + rd(s_ptr, p & rmask);
            }
            wr(s_ptr, p & rmask, 0u);
        }
        __syncthreads();
        const uint32_t cov = boff[gp + 1] - eoff[gp];
        const uint32_t need = min(cov, M);
        uint32_t k = need > cur ? need - cur : 0u;
        while (k > 0) {
            // best head among buckets q in (p - max_span, p]
            uint64_t best = 0;
            for (uint32_t t = lane; t < max_span && t <= p; t += 64) {
                const uint32_t q = p - t;
                const uint32_t gq = base + q;
                const uint32_t b0 = boff[gq], b1 = boff[gq + 1];
                if (b0 == b1) continue;  // nothing starts here (most positions, when reads are long)
                const uint32_t ptr = rd(s_ptr, q & rmask);
                if (b0 + ptr < b1) {
                    const uint64_t key = skeys.key(b0 + ptr);
                    const uint32_t span = max_span - (uint32_t)(key & code_mask);
                    const uint32_t end = q + span - 1;
                    if (end >= p) {
                        const uint64_t pri = ((uint64_t)(end + 1) << 32) | (uint64_t)(q + 1);
                        best = pri > best ? pri : best;
                    }
                }
            }
            best = wave_max_u64(best);
            // need <= cov guarantees a candidate; guard anyway so the loop always ends
            if (best == 0) break;
            const uint32_t bend = (uint32_t)(best >> 32) - 1;
            const uint32_t bq = (uint32_t)(best & 0xFFFFFFFFu) - 1;
            const uint32_t gq = base + bq;
            const uint32_t b0 = boff[gq], b1 = boff[gq + 1];
            const uint32_t ptr = rd(s_ptr, bq & rmask);
            // length of the run of equal-end reads at the head of the winning bucket (<= 64)
            bool same = false;
            const uint32_t j = b0 + ptr + lane;
            if (j < b1) {
                const uint64_t key = skeys.key(j);
                const uint32_t span = max_span - (uint32_t)(key & code_mask);
                same = (bq + span - 1) == bend;
            }
            const uint64_t ball = __ballot(same);
            const uint32_t run = (~ball == 0ull) ? 64u : (uint32_t)(__ffsll((long long)~ball) - 1);
            const uint32_t take = min(k, run);
            __syncthreads();
            if (lane == 0) {
                wr(s_ptr, bq & rmask, ptr + take);
                wr(s_exp, bend & rmask, rd(s_exp, bend & rmask) + take);
            }
            __syncthreads();
            cur += take;
            k -= take;
        }
        // reads ending at p stop covering p+1
        const uint32_t ex = rd(s_exp, p & rmask);
        cur -= ex;
        __syncthreads();
        if (lane == 0) wr(s_exp, p & rmask, 0u);
    }
    __syncthreads();
    // flush the buckets still in the ring
    const uint32_t first = L > ring_size ? L - ring_size : 0u;
    for (uint32_t q = first + lane; q < L; q += 64) selend[base + q] = boff[base + q] + rd(s_ptr, q & rmask);
}
// ------------------------------------------------------------------ mixed-span sweep, LDS-cached
// Same rule as k_sweep_general, organised so that the serial loop touches LDS only:
//   * a preprocessing pass marks run heads of equal composite keys (a "group": reads with the
//     same start and end) and a reverse min-scan turns them into next_head[], so the length of
//     the run starting at j is next_head[j + 1] - j;
//   * positions are taken 64 at a time: bucket bounds, coverage and the first TWO groups of
//     every entering bucket are loaded with wave-wide (not serially dependent) loads into an LDS
//     ring of `ring` slots (power of two >= max_span + 64, so that a slot is only recycled once
//     its previous bucket is dead even for the last position of a chunk), and the previous
//     occupants of those slots flush their selected counts to selend;
//   * a selection event is a wave-wide maximum over the cached bucket heads, key
//     (end - p + 1) << 16 | (0xFFFF - (p - q)): largest end, then largest start; it takes
//     min(deficit, run) reads from the winning group.  Only when a bucket has used up both cached
//     groups is its next group fetched from memory.
// One wave per contig; spans up to kMaxCachedSpan.
struct GenSlots {  // layout of the LDS ring, in 32-bit words per slot
    // G0 / G1: (end + 1, run) of the bucket's head group and of the cached second group,
    // 8 bytes each so one ds_read_b64 fetches both fields
    enum { kG0 = 0, kG1 = 2, kNextJ = 4, kB1 = 5, kTaken = 6, kExp = 7, kWords = 8 };
};

template <typename Sorted>
__global__ __launch_bounds__(256) void k_group_heads(Sorted skeys, uint32_t n,
                                                     uint32_t* __restrict__ next_head) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j <= n; j += stride) {
        uint32_t v = 0xFFFFFFFFu;
        if (j == n) v = n;
        else if (j == 0 || skeys.key(j - 1) != skeys.key(j)) v = j;
        next_head[j] = v;
    }
}

template <typename Sorted>
__global__ __launch_bounds__(64) void k_sweep_general_cached(
    const uint32_t* __restrict__ boff, const uint32_t* __restrict__ eoff, Sorted skeys,
    const uint32_t* __restrict__ next_head, const uint64_t* __restrict__ contig_pos_off,
    uint32_t span_bits, uint32_t max_span, uint32_t M, uint32_t* __restrict__ selend, uint32_t ring,
    const uint32_t* __restrict__ seg) {
    extern __shared__ uint32_t s_gen[];
    uint2* s_g0 = reinterpret_cast<uint2*>(s_gen + GenSlots::kG0 * ring);
    uint2* s_g1 = reinterpret_cast<uint2*>(s_gen + GenSlots::kG1 * ring);
    uint32_t* s_nextj = s_gen + GenSlots::kNextJ * ring;
    uint32_t* s_b1 = s_gen + GenSlots::kB1 * ring;
    uint32_t* s_taken = s_gen + GenSlots::kTaken * ring;
    uint32_t* s_exp = s_gen + GenSlots::kExp * ring;
    const uint32_t lane = threadIdx.x;
    const uint32_t c_id = blockIdx.x;
    SweepSeg sg;
    if (!sweep_segment(contig_pos_off, seg, c_id, sg)) return;
    const uint32_t base = sg.base, L = sg.Lrun;  // a stretch never looks past its own end
    const uint32_t rmask = ring - 1;
    const uint64_t code_mask = (1ull << span_bits) - 1;
    for (uint32_t i = lane; i < GenSlots::kWords * ring; i += 64) s_gen[i] = 0;
    __syncthreads();
    const uint32_t* __restrict__ cb = boff + base;
    const uint32_t* __restrict__ ce = eoff + base;
    uint32_t* __restrict__ csel = selend + base;
    uint32_t cur = 0;
    if (seg != nullptr) cur = seed_stretch_expiry(skeys, boff, base, span_bits, max_span, s_exp, ring, lane);
    // One wave per workgroup: its LDS operations execute in program order, so lane-0 updates
    // are visible to every lane's next read without barriers.
    for (uint32_t p0 = 0; p0 < L; p0 += 64) {
        // ---- enter the chunk's 64 buckets (lane = position p0 + lane)
        const uint32_t q = p0 + lane;
        const uint32_t slot = q & rmask;
        uint32_t need = 0;
        uint32_t exp_c = 0;  // selected reads ending at position p0 + lane
        if (q < L) {
            if (q >= ring) csel[q - ring] = cb[q - ring] + s_taken[slot];  // recycled slot
            exp_c = s_exp[slot];
            s_exp[slot] = 0;
            const uint32_t b0 = cb[q], b1 = cb[q + 1];
            need = min(b1 - ce[q], M);  // cov(q) = boff[q + 1] - eoff[q]
            uint2 g0 = make_uint2(0, 0), g1 = make_uint2(0, 0);
            uint32_t nj = b1;
            if (b0 < b1) {
                const uint64_t k0 = skeys.key(b0);
                g0.y = min(next_head[b0 + 1], b1) - b0;
                g0.x = q + (max_span - (uint32_t)(k0 & code_mask));  // end + 1
                const uint32_t j1 = b0 + g0.y;
                nj = j1;
                if (j1 < b1) {
                    const uint64_t k1 = skeys.key(j1);
                    g1.y = min(next_head[j1 + 1], b1) - j1;
                    g1.x = q + (max_span - (uint32_t)(k1 & code_mask));
                    nj = j1 + g1.y;
                }
            }
            s_g0[slot] = g0;
            s_g1[slot] = g1;
            s_nextj[slot] = nj;
            s_b1[slot] = b1;
            s_taken[slot] = 0;
        }
        // ---- walk the chunk, event by event: lane j holds need(p0 + j) and the selected coverage at
        // p0 + j given the selections so far (exclusive prefix sum of the chunk's expiry counts, then
        // += take on the lanes a selected group covers), so the next position with a deficit is one
        // ballot away and positions without one cost nothing
        uint32_t curv = cur - (wave_incl_scan_add(exp_c) - exp_c);
        for (;;) {
            const unsigned long long pend = __ballot(need > curv);
            if (pend == 0) break;
            const uint32_t j = (uint32_t)__ffsll((long long)pend) - 1;  // first position with a deficit
            const uint32_t p = p0 + j;
            const uint32_t k = __builtin_amdgcn_readlane(need - curv, j);
            // best live head among buckets q' in (p - max_span, p]:
            // key = (end + 1 - p) << 16 | (0xFFFF - (p - q')): largest end, then largest start
            uint32_t best = 0, my_run = 0;
            for (uint32_t t = lane; t < max_span && t <= p; t += 64) {
                const uint2 g = s_g0[(p - t) & rmask];
                if (g.x > p) {
                    const uint32_t key = ((g.x - p) << 16) | (0xFFFFu - t);
                    if (key > best) { best = key; my_run = g.y; }
                }
            }
            uint32_t top = best;
            top = max(top, QMCP_DPP(0u, top, 0x111, 0xF));
            top = max(top, QMCP_DPP(0u, top, 0x112, 0xF));
            top = max(top, QMCP_DPP(0u, top, 0x114, 0xF));
            top = max(top, QMCP_DPP(0u, top, 0x118, 0xF));
            top = max(top, QMCP_DPP(0u, top, 0x142, 0xA));
            top = max(top, QMCP_DPP(0u, top, 0x143, 0xC));
            top = __builtin_amdgcn_readlane(top, 63);
            if (top == 0) break;  // cannot happen (need <= cov); keeps the loop finite
            const uint32_t src = (uint32_t)__ffsll((long long)__ballot(best == top)) - 1;
            const uint32_t run = __builtin_amdgcn_readlane(my_run, src);
            const uint32_t bq = p - (0xFFFFu - (top & 0xFFFFu));
            const uint32_t bslot = bq & rmask;
            const uint32_t bend = p + (top >> 16) - 1;  // end of the winning group (>= p: it is live)
            const uint32_t take = min(k, run);
            // expiry bookkeeping: inside the chunk in the lane register, beyond it in the ring
            if (bend < p0 + 64) {
                exp_c += (lane == bend - p0) ? take : 0u;
            } else if (lane == 0) {
                atomicAdd(&s_exp[bend & rmask], take);
            }
            curv += (lane >= j && lane <= bend - p0) ? take : 0u;
            if (lane == 0) {
                atomicAdd(&s_taken[bslot], take);
                if (take < run) {
                    s_g0[bslot].y = run - take;
                } else {
                    const uint2 g1 = s_g1[bslot];
                    if (g1.y != 0) {
                        // group used up: promote the cached second group (refilled lazily)
                        s_g0[bslot] = g1;
                        s_g1[bslot].y = 0;
                    } else {
                        // both cached groups used: fetch the bucket's next group, if any
                        const uint32_t nj = s_nextj[bslot];
                        const uint32_t b1 = s_b1[bslot];
                        uint2 g0 = make_uint2(0, 0);
                        if (nj < b1) {
                            const uint64_t kk = skeys.key(nj);
                            g0.y = min(next_head[nj + 1], b1) - nj;
                            g0.x = bq + (max_span - (uint32_t)(kk & code_mask));
                            s_nextj[bslot] = nj + g0.y;
                        }
                        s_g0[bslot] = g0;
                    }
                }
            }
        }
        // reads ending at the chunk's last position stop covering the next chunk
        cur = __builtin_amdgcn_readlane(curv - exp_c, 63);
    }
    // flush the buckets still in the ring
    const uint32_t first = L > ring ? L - ring : 0u;
    for (uint32_t qq = first + lane; qq < L; qq += 64) csel[qq] = cb[qq] + s_taken[qq & rmask];
}

// ------------------------------------------------------------------ mixed-span sweep, register-resident
// Register-resident form of the cached event sweep, for max_span + 64 <= 64 * B: the window of
// live buckets is at most 64 * B positions wide, so every lane OWNS B of them (bucket q belongs to
// lane q % 64, slot (q / 64) % B) and keeps their head group, cached second group, read pointers
// and selected count in registers, plus one packed key per bucket:
//     (end + 1 - pbase) << 16 | (start - pbase) << 7 | min(run, 127)      pbase = p0 - 64 (B - 1)
// largest end first, then largest start (the canonical rule; the run bits never decide, starts are
// distinct).  The wave maximum of the live keys therefore names the winning lane, its slot, the
// group's end and (capped) run all at once -- no second lookup.
// The walk over a chunk of 64 positions is event-driven: lane j holds need(p0 + j) and the
// coverage by selected reads at p0 + j given the selections so far (an exclusive prefix sum of the
// chunk's expiry counts at entry, then += take on the lanes a selected group covers), so the next
// position with a deficit is one ballot away and positions without one cost nothing.  A selection
// event is: ballot, lane candidates (register compares), a fused-DPP wave maximum, scalar decode,
// two masked vector updates and the register update in the winning lane -- no LDS round trip on the
// serial path.  Only the expiry counts of reads that end beyond the chunk go through an LDS ring
// (fire-and-forget adds, read back when their chunk enters).  The chunk loop is unrolled B times so
// that the slot a chunk's buckets enter is a compile-time index.
// 1 + kRegLoaders waves: the walker above, and loader waves that run ahead of it -- bucket bounds,
// coverage and the first two groups of every entering bucket are three dependent trips to memory
// (~3 us on sorted records that miss L2), which the walker would otherwise sit out at every chunk's
// entry.  Loader wave w takes the chunks c = w (mod kRegLoaders) and hands each over through its
// own LDS buffer (2 * kRegLoaders buffers in rotation) and a progress word per loader; the walker
// publishes how many chunks it has taken in.  No barrier after the roles split.  The launcher
// gives few workgroups four loaders each and many workgroups (stretches) one.
__device__ __forceinline__ uint32_t reg_sweep_key(uint32_t gx, uint32_t gy, uint32_t qrel, uint32_t pbase) {
    return gy != 0 ? (((gx - pbase) << 16) | (qrel << 7) | min(gy, 127u)) : 0u;
}

static constexpr uint32_t kSpecSnapWords = 512;  // >= the register sweep's window of live buckets

template <typename Sorted, int B, int kRegLoaders>
__global__ __launch_bounds__(64 * (1 + kRegLoaders)) void k_sweep_general_reg(
    const uint32_t* __restrict__ boff, const uint32_t* __restrict__ eoff, Sorted skeys,
    const uint32_t* __restrict__ next_head, const uint64_t* __restrict__ contig_pos_off,
    uint32_t span_bits, uint32_t max_span, uint32_t M, uint32_t* __restrict__ selend,
    const uint32_t* __restrict__ seg,
    uint32_t* __restrict__ selend_odd /* odd stretches' output (speculative tables); or null */,
    const uint32_t* __restrict__ redo_in /* or null: every stretch (else: a later tier of a speculative sweep) */,
    uint32_t n_cand /* speculative tables: entries per column of seg */,
    uint32_t* __restrict__ snap /* speculative tables: kSpecSnapWords per stretch */
#ifdef QMCP_GEN_STAMP
    , unsigned long long* __restrict__ stamps  // lab builds: [0] entry cycles [1] events [2] event cycles [3] fetches [4] fetch cycles [5] walk cycles
#endif
    ) {
    static_assert(B >= 2 && B <= 8, "key fields: 10 bits of end, 9 bits of start");
    constexpr uint32_t kRing = 64 * B;  // >= max_span + 64
    constexpr uint32_t kBack = 64u * (uint32_t)(B - 1);  // p0 - pbase
    __shared__ uint32_t s_exp[kRing];
    // what a loader hands the walker for a chunk: need, bucket start and end, head group, second group,
    // next unread group
    enum { kInNeed, kInB0, kInB1, kInG0x, kInG0y, kInG1x, kInG1y, kInNext, kInWords };
    constexpr uint32_t kBufs = 2u * (uint32_t)kRegLoaders;
    __shared__ uint32_t s_in[kBufs][kInWords][64];
    __shared__ uint32_t s_flag_words[kRegLoaders + 1];  // [w] chunks loader w has handed over, [kRegLoaders] chunks the walker has taken in
    typedef __attribute__((address_space(3))) uint32_t LdsWord;
    volatile LdsWord* const s_flag = (volatile LdsWord*)s_flag_words;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t role = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // 0 walker, 1.. loaders
    const uint32_t c_id = blockIdx.x;
    if (redo_in != nullptr && (c_id >= seg[0] || spec_stretch_idle(seg, n_cand, c_id, redo_in))) return;
    SweepSeg sg;
    if (!sweep_segment(contig_pos_off, seg, c_id, sg)) return;
    const uint32_t base = sg.base, L = sg.Lrun;  // a stretch never looks past its own end
    const uint64_t code_mask = (1ull << span_bits) - 1;
    for (uint32_t i = threadIdx.x; i < kRing; i += blockDim.x) s_exp[i] = 0;
    if (threadIdx.x <= (uint32_t)kRegLoaders) s_flag[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t* __restrict__ cb = boff + base;
    const uint32_t* __restrict__ ce = eoff + base;
    uint32_t* __restrict__ csel = (selend_odd != nullptr && (c_id & 1u) ? selend_odd : selend) + base;
    // per owned bucket: head group (end + 1, run), cached second group, next unread group, bucket
    // start and end, reads selected so far
    uint32_t g0x[B], g0y[B], g1x[B], g1y[B], nextj[B], bstart[B], bend1[B], taken[B];
#pragma unroll
    for (int b = 0; b < B; ++b) { g0x[b] = g0y[b] = g1x[b] = g1y[b] = nextj[b] = bstart[b] = bend1[b] = taken[b] = 0; }
    uint32_t cur = 0;  // selected reads covering the chunk's first position, before its own selections
    if (seg != nullptr) {
        if (role == 0) cur = seed_stretch_expiry(skeys, boff, base, span_bits, max_span, s_exp, kRing, lane);
        __syncthreads();
    }
    const uint32_t n_chunks = (L + 63) / 64;
    // A speculative stretch owns the positions from `own` on and runs in from `base` (a multiple of 64
    // positions earlier).  When it reaches `own` it records, for the buckets still alive, how many reads
    // each has given so far: the whole state of the walk at that point, which k_spec_verify_mixed compares
    // with what the stretch before it -- which ends there -- left behind.
    uint32_t snap_chunk = 0xFFFFFFFFu;
    if (selend_odd != nullptr && seg != nullptr) {
        const uint32_t own = seg[1 + 3 * n_cand + c_id];
        if (own != base) snap_chunk = (own - base) / 64u;
    }
    uint32_t* const my_snap = snap + (size_t)c_id * kSpecSnapWords;

    auto load_group = [&](uint32_t j, uint32_t b1, uint32_t q, uint32_t& gx, uint32_t& gy) {
        // group starting at sorted index j of the bucket of position q ending at b1 (gy = 0: none)
        gx = 0; gy = 0;
        if (j < b1) {
            const uint64_t kk = skeys.key(j);
            gy = min(next_head[j + 1], b1) - j;
            gx = q + (max_span - (uint32_t)(kk & code_mask));  // end + 1
        }
    };

    if (role != 0) {
        // ---- loader w: chunks w - 1, w - 1 + kRegLoaders, ...; buffer c % kBufs is free once the walker
        // has taken chunk c - kBufs in
        uint32_t handed = 0, walker_at = 0;
        for (uint32_t c = role - 1u; c < n_chunks; c += (uint32_t)kRegLoaders) {
            const uint32_t q = c * 64 + lane, buf = c % kBufs;
            uint32_t need = 0, b0 = 0, b1 = 0, x0 = 0, y0 = 0, x1 = 0, y1 = 0, nj = 0;
            if (q < L) {
                b0 = cb[q];
                b1 = cb[q + 1];
                need = min(b1 - ce[q], M);  // cov(q) = boff[q + 1] - eoff[q]
                load_group(b0, b1, q, x0, y0);
                load_group(b0 + y0, b1, q, x1, y1);
                nj = b0 + y0 + y1;
            }
            while (c >= kBufs && walker_at + kBufs <= c) {
                walker_at = s_flag[kRegLoaders];
                if (walker_at + kBufs <= c) __builtin_amdgcn_s_sleep(8);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            s_in[buf][kInNeed][lane] = need; s_in[buf][kInB0][lane] = b0; s_in[buf][kInB1][lane] = b1;
            s_in[buf][kInG0x][lane] = x0; s_in[buf][kInG0y][lane] = y0;
            s_in[buf][kInG1x][lane] = x1; s_in[buf][kInG1y][lane] = y1;
            s_in[buf][kInNext][lane] = nj;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            handed += 1;
            if (lane == 0) s_flag[role - 1u] = handed;
        }
        return;
    }

    for (uint32_t c0 = 0; c0 < n_chunks; c0 += B) {
#pragma unroll
        for (int e = 0; e < B; ++e) {
            const uint32_t c = c0 + e;
            if (c >= n_chunks) break;
            const uint32_t p0 = c * 64;
#ifdef QMCP_GEN_STAMP
            const unsigned long long st_e0 = __builtin_amdgcn_s_memtime();
#endif
            // ---- the chunk's 64 buckets enter slot e (lane = position p0 + lane): the slot's previous
            // buckets hand their selected counts to the loader, which hands this chunk's buckets over
            const uint32_t q = p0 + lane;
            uint32_t need = 0, exp_c = 0;
            {
                const uint32_t qq = q - kRing;
                if (c >= (uint32_t)B && qq < L) csel[qq] = bstart[e] + taken[e];
            }
            if (c == snap_chunk) {
#pragma unroll
                for (int b = 0; b < B; ++b) {
                    if (b != e) {
                        const uint32_t back = 64u * (uint32_t)((e - b + B) % B);  // chunks ago, in positions
                        if (p0 >= back) my_snap[(base + p0 - back + lane) % kSpecSnapWords] = bstart[b] + taken[b];
                    }
                }
            }
            {
                // chunk c is the (c / kRegLoaders + 1)-th of loader c % kRegLoaders
                const uint32_t w = c % (uint32_t)kRegLoaders, want = c / (uint32_t)kRegLoaders + 1u;
                while (s_flag[w] < want) __builtin_amdgcn_s_sleep(2);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                const uint32_t buf = c % kBufs;
                need = s_in[buf][kInNeed][lane];
                bstart[e] = s_in[buf][kInB0][lane];
                bend1[e] = s_in[buf][kInB1][lane];
                g0x[e] = s_in[buf][kInG0x][lane]; g0y[e] = s_in[buf][kInG0y][lane];
                g1x[e] = s_in[buf][kInG1x][lane]; g1y[e] = s_in[buf][kInG1y][lane];
                nextj[e] = s_in[buf][kInNext][lane];
                taken[e] = 0;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) s_flag[kRegLoaders] = c + 1u;
            }
            if (q < L) {
                exp_c = s_exp[q % kRing];
                s_exp[q % kRing] = 0;
            }
#ifdef QMCP_GEN_STAMP
            __builtin_amdgcn_s_waitcnt(0);
            const unsigned long long st_e1 = __builtin_amdgcn_s_memtime();
            unsigned long long st_ev = 0, st_nev = 0, st_fe = 0, st_nfe = 0;
#endif
            // ---- keys of the owned buckets, relative to this chunk (the first chunks' pbase wraps:
            // consistently, every difference below is small and positive)
            const uint32_t pbase = p0 - kBack;
            uint32_t key[B];
#pragma unroll
            for (int b = 0; b < B; ++b) {
                const uint32_t back = 64u * (uint32_t)((e - b + B) % B);  // chunks ago, in positions
                key[b] = reg_sweep_key(g0x[b], g0y[b], kBack - back + lane, pbase);
            }
            // ---- selected coverage at every position of the chunk, before the selections made there
            uint32_t curv = cur - (wave_incl_scan_add(exp_c) - exp_c);
            for (;;) {
                const unsigned long long pend = __ballot(need > curv);
                if (pend == 0) break;
#ifdef QMCP_GEN_STAMP
                const unsigned long long st_v0 = __builtin_amdgcn_s_memtime();
#endif
                const uint32_t j = (uint32_t)__ffsll((long long)pend) - 1;  // first position with a deficit
                const uint32_t k = __builtin_amdgcn_readlane(need - curv, j);
                // this lane's best head: the buckets it owns, the entering one once its position is reached
                uint32_t m = lane <= j ? key[e] : 0u;
#pragma unroll
                for (int b = 0; b < B; ++b)
                    if (b != e) m = max(m, key[b]);
                // live at p iff end >= p; if the lane's best is dead so is all else it owns (smaller ends)
                uint32_t top = (m >> 16) > kBack + j ? m : 0u;
                top = max(top, QMCP_DPP(0u, top, 0x111, 0xF));
                top = max(top, QMCP_DPP(0u, top, 0x112, 0xF));
                top = max(top, QMCP_DPP(0u, top, 0x114, 0xF));
                top = max(top, QMCP_DPP(0u, top, 0x118, 0xF));
                top = max(top, QMCP_DPP(0u, top, 0x142, 0xA));
                top = max(top, QMCP_DPP(0u, top, 0x143, 0xC));
                top = __builtin_amdgcn_readlane(top, 63);
                if (top == 0) break;  // cannot happen (need <= cov); keeps the loop finite
                const uint32_t qrel = (top >> 7) & 511u;
                const uint32_t src = qrel & 63u;
                uint32_t wslot = (uint32_t)e + 1u + (qrel >> 6);  // slot of the chunk that bucket entered in
                wslot = wslot >= (uint32_t)B ? wslot - (uint32_t)B : wslot;
                const uint32_t take = min(k, top & 127u);        // <= the group's true run
                const uint32_t bend_rel = (top >> 16) - 1u - kBack;  // group's end - p0  (>= j: it is live)
                // expiry bookkeeping: inside the chunk in the lane register, beyond it in the ring
                if (bend_rel < 64u) {
                    exp_c += (lane == bend_rel) ? take : 0u;
                } else if (lane == 0) {
                    atomicAdd(&s_exp[(p0 + bend_rel) % kRing], take);
                }
                curv += (lane >= j && lane <= bend_rel) ? take : 0u;
                // the winning lane updates its own bucket in registers (compile-time slot: one copy of
                // this code per slot, all but one skipped)
#pragma unroll
                for (int b = 0; b < B; ++b) {
                    if (wslot == (uint32_t)b) {
                        if (lane == src) {
                            const uint32_t back = 64u * (uint32_t)((e - b + B) % B);
                            const uint32_t run = g0y[b];
                            taken[b] += take;
                            if (take < run) {
                                g0y[b] = run - take;
                            } else if (g1y[b] != 0) {
                                g0x[b] = g1x[b]; g0y[b] = g1y[b]; g1y[b] = 0;   // promote the cached group
                            } else {
                                // both cached groups used: fetch the bucket's next group, if any
#ifdef QMCP_GEN_STAMP
                                const unsigned long long st_f0 = __builtin_amdgcn_s_memtime();
#endif
                                load_group(nextj[b], bend1[b], p0 + lane - back, g0x[b], g0y[b]);
                                nextj[b] += g0y[b];
#ifdef QMCP_GEN_STAMP
                                __builtin_amdgcn_s_waitcnt(0);
                                st_fe += __builtin_amdgcn_s_memtime() - st_f0;
                                st_nfe += 1;
#endif
                            }
                            key[b] = reg_sweep_key(g0x[b], g0y[b], kBack - back + lane, pbase);
                        }
                    }
                }
#ifdef QMCP_GEN_STAMP
                st_ev += __builtin_amdgcn_s_memtime() - st_v0;
                st_nev += 1;
#endif
            }
            // reads ending at the chunk's last position stop covering the next chunk
            cur = __builtin_amdgcn_readlane(curv - exp_c, 63);
#ifdef QMCP_GEN_STAMP
            {
                const unsigned long long st_w = __builtin_amdgcn_s_memtime() - st_e1;
                // fetch counters live in the winning lanes: reduce over the wave
                unsigned long long fe = 0, nfe = 0;
                for (int l = 0; l < 64; ++l) {
                    fe += __shfl((unsigned long long)st_fe, l, 64);
                    nfe += __shfl((unsigned long long)st_nfe, l, 64);
                }
                if (lane == 0 && stamps) {
                    atomicAdd(&stamps[0], st_e1 - st_e0);
                    atomicAdd(&stamps[1], st_nev);
                    atomicAdd(&stamps[2], st_ev);
                    atomicAdd(&stamps[3], nfe);
                    atomicAdd(&stamps[4], fe);
                    atomicAdd(&stamps[5], st_w);
                }
            }
#endif
        }
    }
    // flush the buckets still owned
    const uint32_t last_c = n_chunks - 1;
#pragma unroll
    for (int b = 0; b < B; ++b) {
        // the most recent chunk that filled slot b
        if (last_c >= (uint32_t)b) {
            const uint32_t cc = last_c - ((last_c - (uint32_t)b) % (uint32_t)B);
            const uint32_t qq = cc * 64 + lane;
            if (qq < L) csel[qq] = bstart[b] + taken[b];
        }
    }
}
