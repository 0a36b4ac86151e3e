// near_uniform.inc.hip -- part of qmcp_kernels.hip (one translation unit; included inside namespace qmcp).
// ------------------------------------------------------------------ near-uniform route
// Calls whose reads have one dominant span ell and a small share of SHORTER ones (soft clips, insertions: BamApi
// takes a read's span from its CIGAR, libs/bam-api/src/read.cpp:11-13).  The mixed-span walk is one wave's serial
// chain per contig at ~165 cycles per position; the one-span sweep is ~1.2.  This route keeps the one-span machinery
// and treats the short reads as EXCEPTIONS (tests/near_uniform_model.py restates it on the host, checked against the
// oracle before these kernels were written):
//
//   The canonical greedy (oracle/qmcp_oracle.c, standing in for SimpleMaxFlow::Solve at
//   quasi_mcp_cpu_max_flow_solver.cpp:19-20: at a position with a deficit take the unselected covering reads with the
//   largest end, then the largest start, then the smallest index) files every read under its END: bucket
//   v = end - ell + 1.  A regular read starts at its bucket; a shorter one is released late, at start > v, and from
//   then on goes before its bucket's regular members.  A read selected at time t covers [t, end] whatever its start,
//   so as long as no exception is wanted the selection is the one-span sweep over the regular reads with
//   need'(p) = min(cov_all(p), M) - (exceptions selected so far that cover p, from the time they were selected).
//
//   k_pm_prepare_sort   (pass_major.inc.hip, ell_reg != 0) leaves the exceptions out of the sorted passes and lists them;
//   k_nu_exc_diff + scan + k_nu_need_adjust   coverage of the exceptions -> nadj[p] = need'(p) - min(cov_regular(p), M);
//   k_nu_prepick        an exception that starts where everything is kept (cov_all <= M) is selected when released;
//   [ k_sweep_pack / k_sweep_uniform_ev / k_sweep_expand with nadj  ->  S(p) of the regular reads
//     k_nu_verify       lists the exceptions that need a look: every selected one, and every unselected x = (v, s, e)
//                       that can be reached at all -- that needs the regular members of the buckets (v, s] used up in
//                       the sweep's FINAL counts: one look at bucket s clears nearly all
//     k_nu_replay       one wave per listed exception: the sweep's time-resolved picks over the run of used-up buckets
//                       around s, rebuilt from final counts (below the run's anchor nothing is picked late), and with
//                       them the test "wanted at t:  raw(t) + k(t) > A(t) + r(t)" -- the sweep's demand before its clamp,
//                       the exceptions selected at exactly t, what the buckets above x still offer, the listed
//                       exceptions above x that are candidates at t.  A selected exception must be wanted at its time
//                       and not before, an unselected one never: the first time that fails is its open question
//     k_nu_select_apply settles every open question (select at that time / move earlier / unselect): nadj follows ]
//   repeated until a round leaves no open question -- then the selection is the greedy's (per contig the
//   earliest question is settled for good each round, the others tentatively and nearly always rightly) -- then
//   k_pm_rank_mark for the regular reads and k_nu_mark_selected.
// The host gives up (and takes the mixed-span route) when a listed exception has no anchor within ell buckets and
// nothing earlier in its contig is open, when the rounds outrun their budget, when more than a tenth of the reads are
// exceptions, or when any read is LONGER than the dominant span.
static constexpr uint32_t kNuUnpicked = 0xFFFFFFFFu;
static constexpr unsigned long long kNuNoKey = ~0ull;
// An open question's key: time << 19 | (511 - (end - time)) << 10 | (time - start) << 1 | kind -- earlier time first,
// then the larger end, the larger start; kind 0: the exception is wanted (select it at that time), 1: a selected
// exception is not wanted at its time (unselect it).  kNuUnresolvedLow: the low 19 bits of a read whose run has no
// anchor (no event key has them: time - start < 256).
static constexpr unsigned long long kNuUnresolvedLow = 0x7FFFFull;
static constexpr int kNuKeyShift = 19;
static constexpr int kNuOthers = 1024;  // listed exceptions of one contig whose lives overlap a suspect's and that outrank it or are selected (more: the route gives up)

struct NuExc {  // the exception list: three arrays of cap slots + the route's own two; slots come in groups of 128
    const uint32_t* gs; const uint32_t* ge; const uint32_t* idx; uint32_t* pick; unsigned long long* key;
    const uint32_t* cnt;   // how many of a group's slots hold an exception (k_pm_prepare_sort: one group per wave and pass)
    uint32_t cap;          // all slots: the groups' and, behind them, the overflow region's kNuOverflow
    const uint32_t* n_over;  // entries of the overflow region (the producer's stats[6])
    uint32_t* goff;        // exclusive scan of cnt
    uint32_t* dense;       // the filled slots, in list order (what the per-round kernels walk)
    uint32_t n_dense;      // how many there are (the host's count)
};
__device__ __forceinline__ uint32_t nu_count(const NuExc& x) { return x.cap; }
static constexpr uint32_t kNuGroup = kPmExcPerWave;  // slots per group (a power of two)
__device__ __forceinline__ bool nu_valid(const NuExc& x, uint32_t i) {
    const uint32_t gcap = x.cap - kNuOverflow;
    return i < gcap ? (i & (kNuGroup - 1u)) < x.cnt[i / kNuGroup] : i - gcap < min(*x.n_over, kNuOverflow);
}

__global__ __launch_bounds__(256) void k_nu_count_span(const uint32_t* __restrict__ starts, const uint32_t* __restrict__ ends,
                                                       uint32_t n, uint32_t span, uint32_t* __restrict__ out) {
    uint32_t local = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        local += (ends[i] - starts[i] + 1u == span) ? 1u : 0u;
    local = wave_sum_u32(local);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(out, local);
}

// How uniform are the spans?  A strided sample of the reads into 512 span bins (LDS), then the fullest bin's share:
// out[0] = reads sampled, out[1] = reads in the fullest bin, out[2] = its span.  One workgroup: the sample is 64 Ki reads at most.
__global__ __launch_bounds__(1024) void k_nu_span_mode_share(const uint32_t* __restrict__ starts, const uint32_t* __restrict__ ends,
                                                             uint32_t n, uint32_t stride, uint32_t* __restrict__ out) {
    __shared__ uint32_t s_bin[512];
    __shared__ uint32_t s_red[16];
    for (uint32_t i = threadIdx.x; i < 512; i += blockDim.x) s_bin[i] = 0;
    __syncthreads();
    uint32_t sampled = 0;
    for (uint64_t j = threadIdx.x; j * stride < n && j < 65536u; j += blockDim.x) {
        const uint32_t i = (uint32_t)(j * stride);
        atomicAdd(&s_bin[min(ends[i] - starts[i] + 1u, 511u)], 1u);
        ++sampled;
    }
    __syncthreads();
    uint32_t best = 0;
    for (uint32_t i = threadIdx.x; i < 512; i += blockDim.x) best = max(best, s_bin[i]);
    best = wave_max_u32(best);
    sampled = wave_sum_u32(sampled);
    if ((threadIdx.x & 63) == 0) { s_red[threadIdx.x >> 6] = best; atomicAdd(&out[0], sampled); }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t m = 0;
        for (uint32_t w = 0; w < blockDim.x / 64; ++w) m = max(m, s_red[w]);
        out[1] = m;
        for (uint32_t i = 0; i < 512; ++i)
            if (s_bin[i] == m) { out[2] = i; break; }  // the fullest bin's span (511: that or longer)
    }
}
void launch_span_mode_share(hipStream_t st, const uint32_t* starts, const uint32_t* ends, uint32_t n, uint32_t* out /* 3 words, zeroed here */) {
    (void)hipMemsetAsync(out, 0, 3 * sizeof(uint32_t), st);
    const uint32_t stride = n > 65536u ? n / 65536u : 1u;
    hipLaunchKernelGGL(k_nu_span_mode_share, dim3(1), dim3(1024), 0, st, starts, ends, n, stride, out);
}

// +1 at an exception's start, -1 behind its end, and its flags reset
__global__ __launch_bounds__(256) void k_nu_exc_diff(NuExc x, uint32_t* __restrict__ diff) {
    const uint32_t n = nu_count(x);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (!nu_valid(x, i)) continue;
        const uint32_t gcap = x.cap - kNuOverflow;  // (the overflow region's entries follow the groups' in the dense order)
        const uint32_t at = i < gcap ? x.goff[i / kNuGroup] + (i & (kNuGroup - 1u)) : x.goff[gcap / kNuGroup] + (i - gcap);
        if (at < x.n_dense) x.dense[at] = i;
        atomicAdd(&diff[x.gs[i]], 1u);
        atomicAdd(&diff[x.ge[i] + 1u], 0xFFFFFFFFu);
        x.pick[i] = kNuUnpicked;
        x.key[i] = kNuNoKey;
    }
}

// ce = exclusive scan of diff: exceptions covering p = ce[p + 1]
__device__ __forceinline__ uint32_t nu_cov_regular(const uint32_t* __restrict__ boff, uint32_t p, uint32_t ell) {
    return boff[p + 1] - boff[p + 1 >= ell ? p + 1 - ell : 0u];  // (the global prefix needs no clamp at contig borders)
}
__global__ __launch_bounds__(256) void k_nu_need_adjust(const uint32_t* __restrict__ boff, const uint32_t* __restrict__ ce,
                                                        uint32_t ltot, uint32_t ell, uint32_t M, int32_t* __restrict__ nadj) {
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p <= ltot; p += gridDim.x * blockDim.x) {
        int32_t a = 0;
        if (p < ltot) {
            const uint32_t cr = nu_cov_regular(boff, p, ell);
            a = (int32_t)min(cr + ce[p + 1], M) - (int32_t)min(cr, M);
        }
        nadj[p] = a;
    }
}

__global__ __launch_bounds__(256) void k_nu_prepick(NuExc x, const uint32_t* __restrict__ boff, const uint32_t* __restrict__ ce,
                                                    uint32_t ell, uint32_t M, int32_t* __restrict__ nadj,
                                                    uint32_t* __restrict__ state) {
    const uint32_t n = nu_count(x);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (!nu_valid(x, i)) continue;
        const uint32_t s = x.gs[i], e = x.ge[i];
        if (nu_cov_regular(boff, s, ell) + ce[s + 1] <= M) {
            x.pick[i] = s;
            for (uint32_t p = s; p <= e; ++p) atomicSub(&nadj[p], 1);
            atomicAdd(&state[3], 1u);
        }
    }
}

struct NuView {  // what verify and replay read of the sweep's result
    const uint32_t* boff; const uint32_t* selend; const int32_t* nadj; const uint64_t* poff; uint32_t n_contigs, ell, M;
    const uint32_t* ce;  // exclusive scan of the exceptions' coverage differences: exceptions covering p = ce[p + 1]
};
__device__ __forceinline__ uint32_t nu_c(const NuView& v, uint32_t u) { return v.boff[u + 1] - v.boff[u]; }
__device__ __forceinline__ uint32_t nu_S(const NuView& v, uint32_t u) { return v.selend[u] - v.boff[u]; }
__device__ __forceinline__ uint32_t nu_contig_of(const NuView& v, uint32_t pos) {
    uint32_t lo = 0, hi = v.n_contigs;  // last c with poff[c] <= pos
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if ((uint32_t)v.poff[mid] <= pos) lo = mid; else hi = mid;
    }
    return lo;
}
// a bucket before the contig's first position holds nothing (and nothing is kept across a contig's border)
__device__ __forceinline__ bool nu_exhausted(const NuView& v, int32_t u, int32_t c0) {
    return u < c0 || nu_S(v, (uint32_t)u) == nu_c(v, (uint32_t)u);
}

// The round's suspects binned by start position (round 4: a replay needs the listed exceptions whose lives overlap its
// own, and looked through the whole list for them -- 16 ms a round at 20 k suspects on two 10^7-position contigs): cells
// of 2^shift positions, counted by k_nu_verify, scanned, filled by k_nu_bin; a replay reads the cells over (s - ell, e].
struct NuBins {
    uint32_t* count;   // n_cells + 1 words: suspects per cell, then (zeroed again) the fill counters
    uint32_t* start;   // n_cells + 2: exclusive scan of count
    uint32_t* sorted;  // suspects_cap: slots of the suspects list, by cell
    uint32_t shift, n_cells;
    // Sweeps in stretches (long contigs: a contig-wide "swept from" says little): cells where something a replay reads has
    // changed since the round before -- the sweep's counts (k_nu_diff) or the need (k_nu_select_apply).  A listed
    // exception is replayed again only if a cell of [s - (kNuReach + 1) ell, e + 1] is dirty (or its run had no anchor
    // last time); the others keep their answer and are listed for their neighbours' sake.  null: replay every one.
    const uint32_t* dirty;   // this round's
    uint32_t* dirty_next;    // the next round's (zeroed): written by k_nu_select_apply, then by the next k_nu_diff
};
__device__ __forceinline__ uint32_t nu_cell(const NuBins& b, uint32_t pos) { return min(pos >> b.shift, b.n_cells - 1u); }
__host__ inline uint32_t nu_bin_shift(uint32_t ltot) {
    uint32_t shift = 8;
    while (((uint64_t)ltot >> shift) + 1 > (1u << 17)) ++shift;
    return shift;
}
static constexpr uint32_t kNuMaxCells = (1u << 17) + 1;

// How far below a read the replay looks for a state to start from (an anchor or a cut point), in spans.  One span nearly
// always does; shallow data (1.5 x M: stretches where nearly everything is kept) has runs of used-up buckets longer than
// that -- the replay is the same from any distance (nothing below an anchor is picked at or after the anchor's time), so a
// read whose run has nothing within one span is staged again with kNuReach spans (round 4; tests/near_uniform_model.py
// REACH: at 1.5 x M two of ten cases settled with one span, all ten with six).
static constexpr int kNuReach = 6;
// suspects: {exception, contig} pairs; replay_list: the slots of those that are replayed this round (the others are listed
// for their neighbours' sake).  Slots are handed out a workgroup at a time: one atomic per counter and workgroup (one per
// suspect was 25 k atomics on one word per round on long shallow contigs, most of the round's time).
__global__ __launch_bounds__(256) void k_nu_verify(NuExc x, NuView v, uint2* __restrict__ suspects, uint32_t suspects_cap,
                                                   uint32_t* __restrict__ state,
                                                   const uint32_t* __restrict__ swept_from /* per contig: the first block this round swept */,
                                                   NuBins bins, uint32_t* __restrict__ replay_list) {
    __shared__ uint32_t s_w[4][2], s_base[2];
    if (state[7] == 0u) return;  // (no contig was swept this round: every one is settled)
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    for (uint32_t j0 = blockIdx.x * blockDim.x; j0 < x.n_dense; j0 += gridDim.x * blockDim.x) {
        const uint32_t j = j0 + threadIdx.x;
        bool listed = false, replay = false;
        uint32_t i = 0, contig = 0;
        int32_t s = 0;
        if (j < x.n_dense) {
            i = x.dense[j];
            s = (int32_t)x.gs[i];
            const int32_t e = (int32_t)x.ge[i];
            const int32_t b = e - (int32_t)v.ell + 1;
            contig = nu_contig_of(v, (uint32_t)s);
            const int32_t c0 = (int32_t)(uint32_t)v.poff[contig];
            // what an earlier round settled stays settled: nothing at or below e has changed if the round swept only from
            // a later block on (a contig that is settled sweeps nothing).  One span of slack: an exception listed for its
            // own sake needs the ones whose lives overlap its own beside it, and those may end a span earlier.
            const uint32_t from = swept_from[contig];
            listed = !(from == 0xFFFFFFFFu || (uint64_t)(uint32_t)(e - c0) + v.ell < (uint64_t)from * v.ell);
            if (listed && x.pick[i] == kNuUnpicked) {  // (a selected exception is always listed: its selection is checked)
                bool reach = true;
                for (int32_t u = s; u > b && reach; --u) reach = nu_exhausted(v, u, c0);
                listed = reach;
            }
            if (listed) {
                // replayed: unless it ends before what the round swept, or nothing it reads has changed since it was last
                // replayed (its answer -- no question -- stands); a read whose run had no anchor keeps that key until a
                // replay says otherwise, and is replayed whatever changed
                replay = (uint64_t)(uint32_t)(e - c0) >= (uint64_t)from * v.ell;
                const unsigned long long k0 = x.key[i];
                const bool was_unresolved = bins.dirty != nullptr && k0 != kNuNoKey && (k0 & kNuUnresolvedLow) == kNuUnresolvedLow;
                if (replay && bins.dirty != nullptr && !was_unresolved) {
                    const uint32_t c_lo = nu_cell(bins, (uint32_t)max(s - (kNuReach + 1) * (int32_t)v.ell, 0));
                    const uint32_t c_hi = nu_cell(bins, (uint32_t)e + 1u);
                    bool d = false;
                    for (uint32_t cc = c_lo; cc <= c_hi; ++cc) d |= bins.dirty[cc] != 0u;
                    replay = d;
                }
                if (!was_unresolved || replay) x.key[i] = kNuNoKey;
            }
        }
        const uint64_t ml = __builtin_amdgcn_ballot_w64(listed), mr = __builtin_amdgcn_ballot_w64(replay);
        if (lane == 0) { s_w[w][0] = (uint32_t)__builtin_popcountll(ml); s_w[w][1] = (uint32_t)__builtin_popcountll(mr); }
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t t0 = s_w[0][0] + s_w[1][0] + s_w[2][0] + s_w[3][0], t1 = s_w[0][1] + s_w[1][1] + s_w[2][1] + s_w[3][1];
            s_base[0] = t0 ? atomicAdd(&state[4], t0) : 0u;
            s_base[1] = t1 ? atomicAdd(&state[14], t1) : 0u;
        }
        __syncthreads();
        if (listed) {
            uint32_t slot = s_base[0] + (uint32_t)__builtin_popcountll(ml & ((1ull << lane) - 1ull));
            uint32_t rslot = s_base[1] + (uint32_t)__builtin_popcountll(mr & ((1ull << lane) - 1ull));
            for (uint32_t ww = 0; ww < w; ++ww) { slot += s_w[ww][0]; rslot += s_w[ww][1]; }
            if (slot < suspects_cap) {
                suspects[slot] = make_uint2(i, contig);
                atomicAdd(&bins.count[nu_cell(bins, (uint32_t)s)], 1u);
                if (replay && rslot < suspects_cap) replay_list[rslot] = slot;
            } else {
                atomicOr(&state[2], 2u);  // more suspects than the list holds: the route gives up
            }
        }
        __syncthreads();  // (s_w and s_base are written again by the next strip)
    }
}

// One wave per suspect.  Everything the replay reads -- bucket offsets, the sweep's result and nadj over
// [s - 2 ell, e + 1] -- is staged in LDS first (a trip to memory per replayed position would make an exception in a
// contig's last ell positions, where every bucket is empty and the run is as long as the read, cost half a millisecond).
// after a sweep in stretches: which cells' kept counts differ from the round before (and the copy for the next round)
__global__ __launch_bounds__(256) void k_nu_diff(const uint32_t* __restrict__ selend, uint32_t* __restrict__ prev, uint32_t ltot,
                                                 uint32_t shift, uint32_t n_cells, uint32_t* __restrict__ dirty, uint32_t first_round) {
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < ltot; p += gridDim.x * blockDim.x) {
        const uint32_t a = selend[p];
        if (!first_round && a != prev[p]) dirty[min(p >> shift, n_cells - 1u)] = 1u;
        prev[p] = a;
    }
}

__global__ __launch_bounds__(256) void k_nu_bin(NuExc x, const uint2* __restrict__ suspects, uint32_t suspects_cap,
                                                const uint32_t* __restrict__ state, NuBins bins) {
    const uint32_t n_sus = min(state[4], suspects_cap);
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < n_sus; q += gridDim.x * blockDim.x) {
        const uint32_t cell = nu_cell(bins, x.gs[suspects[q].x]);
        bins.sorted[bins.start[cell] + atomicAdd(&bins.count[cell], 1u)] = q;
    }
}

// staged positions [s - (reach + 1) ell, e + 1] and replayed buckets: (reach + 2) ell + 16 entries each, dynamic LDS.
// Two launches: every listed exception with one span of reach (11 KB of LDS at ell = 150: six workgroups per CU; with
// kNuReach spans for all of them it was two, and the replays of a round are one wave each -- 1.7 ms for 25 k of them),
// then the few whose run had nothing within a span (the far list) with kNuReach.
__host__ __device__ constexpr uint32_t nu_stage_entries(uint32_t ell, bool far) { return (uint32_t)((far ? kNuReach : 1) + 2) * ell + 16u; }
template <bool kFar>
__global__ __launch_bounds__(64) void k_nu_replay(NuExc x, NuView v, const uint2* __restrict__ suspects, uint32_t suspects_cap,
                                                  uint32_t* __restrict__ state, unsigned long long* __restrict__ viol_key,
                                                  const uint32_t* __restrict__ swept_from, NuBins bins,
                                                  const uint32_t* __restrict__ replay_list /* kFar: the far list */,
                                                  uint32_t* __restrict__ far_list /* !kFar: out */) {
    extern __shared__ uint32_t s_nu_dyn[];
    const int32_t kNuStage = (int32_t)nu_stage_entries(v.ell, kFar), kNuCur = kNuStage;
    uint32_t* const s_b = s_nu_dyn;
    uint32_t* const s_e = s_b + kNuStage;
    uint32_t* const s_x = s_e + kNuStage;
    int32_t* const s_a = reinterpret_cast<int32_t*>(s_x + kNuStage);
    int32_t* const s_cur = s_a + kNuStage;
    int32_t* const s_stack = s_cur + kNuStage;
    // (the others' starts, ends and selection times relative to s - ell: lives that overlap [s, e] lie within 3 spans of it)
    // (one word and a half per neighbour: start | end << 16 | outranks << 31, and the selection time -- with four arrays
    //  of 2 048 the workgroup held 25 KB and six fitted a compute unit; a replay is one wave, mostly waiting)
    __shared__ uint32_t o_se[kNuOthers];
    __shared__ short o_t[kNuOthers];
    __shared__ uint32_t o_n;
    const int32_t lane = (int32_t)threadIdx.x;
    const uint32_t n_replay = min(state[kFar ? 15 : 14], suspects_cap);
    const int32_t ell = (int32_t)v.ell;
    for (uint32_t rq = blockIdx.x; rq < n_replay; rq += gridDim.x) {
        const uint32_t sus_slot = replay_list[rq];
        const uint2 su = suspects[sus_slot];
        const uint32_t i = su.x, contig = su.y;
        const int32_t s = (int32_t)x.gs[i], e = (int32_t)x.ge[i];
        const int32_t b = e - ell + 1;
        const int32_t c0 = (int32_t)(uint32_t)v.poff[contig];
        // (the exceptions listed for the others' sake only -- they end before what the round swept, or nothing they read has
        //  changed -- are not in the replay list: k_nu_verify)
        int32_t r0 = 0;
        auto stage = [&](int32_t reach) {
            r0 = max(s - (reach + 1) * ell, 0);
            const int32_t count = e + 1 - r0 + 1;  // <= (reach + 2) ell + 1
            __syncthreads();  // (the previous suspect's -- or the first go's -- reads of the stage are done)
            for (int32_t k = lane; k < count && k < kNuStage; k += 64) {
                s_b[k] = v.boff[r0 + k];
                s_e[k] = v.selend[r0 + k];
                s_a[k] = v.nadj[r0 + k];
                s_x[k] = v.ce[r0 + k + 1];
                s_cur[k] = 0;
            }
            for (int32_t k = count + lane; k < count + 16 && k < kNuCur; k += 64) s_cur[k] = 0;
            __syncthreads();
        };
        stage(kFar ? kNuReach : 1);
        if (lane == 0) o_n = 0;
        __syncthreads();
        // the other listed exceptions of the contig whose lives overlap this one's: {start, end, selection time or -1,
        // outranks this one}
        const uint32_t my_idx = x.idx[i];
        const int32_t sel = x.pick[i] == kNuUnpicked ? -1 : (int32_t)x.pick[i];
        // (an overlapping life starts in (s - ell, e]: spans are at most ell)
        const uint32_t j0 = bins.start[nu_cell(bins, (uint32_t)max(s - ell + 1, 0))];
        const uint32_t j1 = bins.start[nu_cell(bins, (uint32_t)e) + 1u];
        for (uint32_t j = j0 + (uint32_t)lane; j < j1; j += 64u) {
            const uint2 z = suspects[bins.sorted[j]];
            if (z.y != contig || z.x == i) continue;
            const int32_t zs = (int32_t)x.gs[z.x], ze = (int32_t)x.ge[z.x];
            if (ze < s || zs > e) continue;
            // (only a neighbour that outranks this one, or that is selected, enters the test below)
            const bool above = ze > e || (ze == e && (zs > s || (zs == s && x.idx[z.x] < my_idx)));
            const uint32_t zp = x.pick[z.x];
            if (!above && zp == kNuUnpicked) continue;
            const uint32_t slot = atomicAdd(&o_n, 1u);
            if (slot < (uint32_t)kNuOthers) {
                const int32_t rel0 = s - ell;  // (lives that overlap [s, e] lie within three spans of it: 15 bits do)
                o_se[slot] = (uint32_t)(zs - rel0) | ((uint32_t)(ze - rel0) << 16) | (above ? 0x80000000u : 0u);
                o_t[slot] = zp == kNuUnpicked ? (short)-1 : (short)((int32_t)zp - rel0);
            }
        }
        __syncthreads();
        const int32_t n_others = (int32_t)min(o_n, (uint32_t)kNuOthers);
        if (o_n > (uint32_t)kNuOthers) {
            if (lane == 0) atomicOr(&state[2], 4u);  // more neighbours than the list holds: the route gives up
            continue;
        }
        auto C = [&](int32_t u) { return (int32_t)(s_b[u + 1 - r0] - s_b[u - r0]); };
        auto S = [&](int32_t u) { return (int32_t)(s_e[u - r0] - s_b[u - r0]); };
        auto exhausted = [&](int32_t u) { return u < c0 || S(u) == C(u); };
        auto need = [&](int32_t t) {
            const int32_t from = max(t + 1 - ell, 0);  // (>= r0: t > s - ell)
            return (int32_t)min(s_b[t + 1 - r0] - s_b[from - r0], v.M) + s_a[t - r0];
        };
        // the run of exhausted buckets downwards from b; its anchor u1 - 1 is the first bucket that keeps members for
        // good (or lies before the contig): nothing below it is picked at or after its own time
        constexpr int32_t reach = kFar ? kNuReach : 1;
        int32_t u1 = b + 1, cut = -1;
        while (u1 - 1 >= c0 && exhausted(u1 - 1) && s - (u1 - 1) < reach * ell) --u1;
        // No anchor within reach: a CUT POINT does as well.  At a position p with cov_all(p) <= M every read
        // covering it is kept, so at time p every bucket in (p - ell, p] is used up -- a known state to start from.
        if (u1 - 1 >= c0 && exhausted(u1 - 1)) {
            for (int32_t p = s - 1; p > s - reach * ell && p >= c0; --p) {
                const uint32_t cov_reg = s_b[p + 1 - r0] - s_b[max(p + 1 - ell, 0) - r0];
                if (cov_reg + s_x[p - r0] <= v.M) { cut = p; break; }
            }
        }
        if (!kFar && cut < 0 && u1 - 1 >= c0 && exhausted(u1 - 1)) {
            // nothing within a span (rare): the far launch stages kNuReach spans for it
            if (lane == 0) {
                const uint32_t f = atomicAdd(&state[15], 1u);
                if (f < suspects_cap) far_list[f] = sus_slot;
            }
            continue;
        }
        if (cut < 0 && u1 - 1 >= c0 && exhausted(u1 - 1)) {
            // kNuReach spans of exhausted buckets in a row below the read: not modelled.  That only matters if nothing EARLIER in the
            // contig is wanted: behind a wanted exception the sweep ran on a need it could not meet, and what it left
            // there (often every bucket used up) is replaced by the next round's sweep anyway.  So the read enters the
            // contig's contest with a key of its own -- its release time, lowest priority -- and the route gives up
            // only if that key wins.
            if (lane == 0) {
                const unsigned long long k = ((unsigned long long)(uint32_t)s << kNuKeyShift) | kNuUnresolvedLow;
                x.key[i] = k;
                atomicMin(&viol_key[contig], k);
                state[5] = x.idx[i];  // (which read: for the host's debug line)
                state[8] = (uint32_t)s; state[9] = (uint32_t)u1; state[10] = (uint32_t)S(u1 - 1); state[11] = (uint32_t)C(u1 - 1);
                state[12] = (uint32_t)b; state[13] = (uint32_t)c0;
            }
            continue;
        }
        u1 = cut >= 0 ? cut + 1 : max(u1, c0);
        const int32_t anchor = u1 - 1;
        const bool has_anchor = cut < 0 && anchor >= c0;
        // replayed buckets [base, t) keep time-resolved counts in s_cur
        const int32_t base = cut >= 0 ? max(cut - ell + 1, c0) : has_anchor ? anchor : c0;
        auto wave_sum_S = [&](int32_t lo, int32_t hi) {  // final counts of buckets [lo, hi)
            int32_t acc = 0;
            for (int32_t u = lo + lane; u < hi; u += 64) acc += S(u);
            return (int32_t)wave_sum_u32((uint32_t)acc);
        };
        // From here on every lane runs the same scalar program on the staged arrays (writes of one value to one
        // address by all lanes): the window is kept as two running sums -- `fixed`, the final counts of the buckets
        // below the replayed ones, and `repl`, the replayed buckets' counts so far -- and the replayed buckets that
        // still offer members as a stack (a bucket enters once, at its own time, and leaves when it is used up).
        int32_t repl = 0, n_stack = 0;
        if (has_anchor) {
            const int32_t d = need(anchor) - wave_sum_S(max(anchor - ell + 1, 0), anchor);
            repl = min(max(d, 0), C(anchor));
            s_cur[0] = repl;
            s_stack[n_stack++] = anchor;  // (never used up: that is what makes it the anchor)
        }
        int32_t fixed = cut >= 0 ? 0 : wave_sum_S(max(u1 - ell + 1, 0), base);
        if (cut >= 0) {
            // behind a cut: every bucket of (cut - ell, cut] is used up, nothing older is in any later window
            int32_t acc = 0;
            for (int32_t u = base + lane; u <= cut; u += 64) {
                const int32_t cu = C(u);
                s_cur[u - base] = cu;
                if (u > cut + 1 - ell) acc += cu;     // the window of time cut + 1: (cut + 1 - ell, cut + 1)
            }
            repl = (int32_t)wave_sum_u32((uint32_t)acc);
            __syncthreads();
        }
        int32_t avail = 0;   // what the regular members of the buckets (b, t] still offer
        unsigned long long key = kNuNoKey;
        // A selected exception is checked up to its time (wanted there, and not before); an unselected one over its life.
        // Wanted at t:  raw(t) + k(t) > avail(t) + r(t)  -- raw: the sweep's own demand, before the clamp at zero (a
        // selection that was not needed leaves a surplus); k: exceptions selected at exactly t (their unit is back in the
        // demand the greedy saw); r: listed exceptions above this one that are candidates at t.
        const int32_t last = sel >= 0 ? sel : e;
        for (int32_t t = u1; t <= last; ++t) {
            const int32_t raw = need(t) - fixed - repl;
            int32_t d = max(raw, 0);
            const int32_t ct = C(t);
            if (t > b) avail += ct;
            if (t >= s) {
                int32_t k = sel == t ? 1 : 0, r = 0;
                if (n_others > 0) {  // (uniform)
                    int32_t kk = 0, rr = 0;
                    const int32_t tr = t - (s - ell);  // (>= ell: t >= s)
                    for (int32_t j = lane; j < n_others; j += 64) {
                        const int32_t zt = o_t[j];
                        const uint32_t se = o_se[j];
                        kk += zt == tr ? 1 : 0;
                        rr += ((se >> 31) != 0 && (int32_t)(se & 0xFFFFu) <= tr && tr <= (int32_t)((se >> 16) & 0x7FFFu) &&
                               (zt < 0 || zt >= tr)) ? 1 : 0;
                    }
                    k += (int32_t)wave_sum_u32((uint32_t)kk);
                    r = (int32_t)wave_sum_u32((uint32_t)rr);
                }
                const bool wanted = raw + k > avail + r;
                const bool is_open = (sel < 0 || t < sel) ? wanted : !wanted;
                if (is_open) {
                    key = ((unsigned long long)(uint32_t)t << kNuKeyShift) | ((unsigned long long)(511 - (e - t)) << 10) |
                          ((unsigned long long)(t - s) << 1) | ((sel < 0 || t < sel) ? 0ull : 1ull);
                    break;
                }
            }
            avail -= min(d, avail);
            if (ct > 0 && n_stack < kNuCur) s_stack[n_stack++] = t;
            while (d > 0 && n_stack > 0) {  // top-down through what the replayed buckets still offer
                const int32_t u = s_stack[n_stack - 1];
                const int32_t have = C(u) - s_cur[u - base];
                const int32_t k = min(d, have);
                s_cur[u - base] += k;
                repl += k;
                d -= k;
                if (k == have) --n_stack;
            }
            if (sel < 0 && t >= s && !exhausted(t)) break;  // bucket t keeps members for good: x is never reached later
            // the window moves on: bucket t - ell + 1 leaves it
            const int32_t out = t - ell + 1;
            if (out >= base) repl -= s_cur[out - base];
            else if (out >= 0 && cut < 0) fixed -= S(out);
        }
        if (lane == 0) {
            x.key[i] = key;
            if (key != kNuNoKey) atomicMin(&viol_key[contig], key);
        }
    }
}

__global__ __launch_bounds__(256) void k_nu_round_reset(uint32_t* __restrict__ state, unsigned long long* __restrict__ viol_key,
                                                        uint32_t* __restrict__ viol_idx, uint32_t* __restrict__ sweep_from_next,
                                                        uint32_t n_contigs, uint32_t first_round) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        state[7] = first_round | state[1];  // contigs this round's sweep covered: those that selected in the round before
        state[6] += state[1] != 0u ? 1u : 0u;  // rounds that selected something
        state[1] = 0;
        state[4] = 0;
        state[14] = 0;  // exceptions to replay this round
        state[15] = 0;  // ... of them, those whose run has nothing within one span
    }
    if (i < n_contigs) { viol_key[i] = kNuNoKey; viol_idx[i] = 0xFFFFFFFFu; sweep_from_next[i] = 0xFFFFFFFFu; }
}
// Settles every open question of the round: an exception the sweep wants is selected at that time (or
// earlier than it was), one that is not wanted at its time is unselected; nadj follows, and every contig's next sweep
// starts two blocks before its earliest change.  Only a contig's EARLIEST question (then the highest priority) is known
// to be settled for good -- everything before it is certified -- the others are settled tentatively and checked again
// by the next round (tests/near_uniform_model.py: solve_near_uniform_batched).  A contig whose earliest question is a
// read without an anchor ends the attempt.
__global__ __launch_bounds__(256) void k_nu_select_apply(NuExc x, const uint2* __restrict__ suspects, uint32_t suspects_cap,
                                                          uint32_t* __restrict__ state, const unsigned long long* __restrict__ viol_key,
                                                          int32_t* __restrict__ nadj,
                                                          const uint64_t* __restrict__ poff, uint32_t ell,
                                                          uint32_t* __restrict__ sweep_from /* per contig, preset to "settled" */,
                                                          const uint32_t* __restrict__ seg_exact /* sweeps in stretches: the exact
                                                              table [count, {start, end, contig end}...]; or null */,
                                                          uint32_t* __restrict__ marks_next /* per exact stretch, zeroed: 1 = the
                                                              next round's sweep must cover it */,
                                                          NuBins bins,
                                                          const uint32_t* __restrict__ seg_fine /* the first speculative tier's
                                                              table, or null */, uint32_t n_cand,
                                                          uint32_t* __restrict__ fine_next /* per stretch of it, zeroed */) {
    // one wave per listed exception (the lanes share the walks over [t, e]: a round on shallow data settles thousands of
    // questions, a span of atomics each -- 0.35 ms with a thread apiece); what happens once per question is lane 0's
    const uint32_t n_sus = min(state[4], suspects_cap);
    const uint32_t lane = threadIdx.x & 63u, n_waves = gridDim.x * (blockDim.x >> 6);
    uint32_t applied = 0;   // (lane 0's: questions settled, and the change in the number of selected exceptions -- one
    int32_t selected = 0;   //  atomic per wave at the end; one per question was thousands on one word)
    for (uint32_t q = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); q < n_sus; q += n_waves) {
        const uint2 su = suspects[q];
        const unsigned long long k = x.key[su.x];
        if (k == kNuNoKey) continue;
        if ((k & kNuUnresolvedLow) == kNuUnresolvedLow) {
            if (lane == 0 && k == viol_key[su.y]) atomicOr(&state[2], 1u);  // the contig's earliest open question has no answer
            continue;
        }
        const uint32_t t = (uint32_t)(k >> kNuKeyShift), e = x.ge[su.x];
        const uint32_t old = x.pick[su.x];
        __builtin_amdgcn_wave_barrier();  // (every lane has read pick before lane 0 writes it)
        if ((k & 1ull) == 0ull) {  // wanted at t: select it there (it may have been selected later)
            const uint32_t hi = old != kNuUnpicked ? old - 1u : e;
            for (uint32_t p = t + lane; p <= hi; p += 64u) atomicSub(&nadj[p], 1);
            if (lane == 0) {
                x.pick[su.x] = t;
                if (old == kNuUnpicked) ++selected;
            }
        } else {                   // selected at `old`, not wanted there
            for (uint32_t p = old + lane; p <= e; p += 64u) atomicAdd(&nadj[p], 1);
            if (lane == 0) {
                x.pick[su.x] = kNuUnpicked;
                --selected;
            }
        }
        if (lane != 0) continue;
        // the next sweep of this contig: what happens from t on changes buckets above t - ell only, so the state
        // entering the block two before t's is still the chain's (the chain keeps it at every 64th block)
        const uint32_t kt = (t - (uint32_t)poff[su.y]) / ell;
        atomicMin(&sweep_from[su.y], (kt >= 2u ? kt - 2u : 0u) & ~63u);
        if (bins.dirty_next != nullptr) {  // the need changes on [min(t, old), e]
            for (uint32_t cc = nu_cell(bins, min(t, old)); cc <= nu_cell(bins, e); ++cc) bins.dirty_next[cc] = 1u;
        }
        if (seg_exact != nullptr) {
            // the need changes on [min(t, old), e] and the kept counts from the bucket above t - ell on, up to the end of
            // the exact stretch (behind a cut point the sweep starts afresh): every exact stretch that meets
            // [t - ell + 1, e] is swept again
            const uint32_t c0 = (uint32_t)poff[su.y];
            const uint32_t first = min(t, old), lo = first - c0 >= ell ? first - ell + 1u : c0;
            const uint32_t count = seg_exact[0];
            uint32_t a = 0, b = count;  // last stretch with start <= lo (the table is in position order)
            while (b - a > 1) {
                const uint32_t mid = (a + b) >> 1;
                if (seg_exact[1 + 3 * mid] <= lo) a = mid; else b = mid;
            }
            for (uint32_t r = a; r < count && seg_exact[1 + 3 * r] <= e; ++r) marks_next[r] = 1u;
            if (seg_fine != nullptr) {
                // ... and of the speculative stretches only those whose sweep READS a position whose need changes or
                // WRITES a bucket that may keep another count: a stretch looks at [start, end + ell] (run-in included)
                // and a selection at t moves buckets from t - ell + 1 on; two spans of margin either side.  What changes
                // further on changes through the state at a boundary, and that is compared wherever the stretch on either
                // side was swept (k_spec_verify) -- a sweep forgets a change within a run-in, so later rounds, which settle
                // a few questions each, sweep a few stretches instead of the genome (one GPU's share of configs[4] with
                // 1 % clipped reads: rounds 4 to 6 were 2 ms of sweep each for 37, 6 and 0 selections).
                const uint32_t fc = seg_fine[0];
                const uint32_t from = lo - c0 >= 2u * ell ? lo - 2u * ell : c0, upto = e + 2u * ell;
                uint32_t fa = 0, fb = fc;  // first stretch whose end + 2 ell >= from (ends are in position order)
                while (fa < fb) {
                    const uint32_t mid = (fa + fb) >> 1;
                    if (seg_fine[1 + 3 * mid + 1] + 2u * ell < from) fa = mid + 1; else fb = mid;
                }
                for (uint32_t r = fa; r < fc && seg_fine[1 + 3 * r] <= upto; ++r) fine_next[r] = 1u;
            }
        }
        ++applied;
    }
    if (lane == 0 && applied != 0) {
        atomicAdd(&state[1], applied);
        if (selected != 0) atomicAdd(&state[3], (uint32_t)selected);
    }
}

__global__ __launch_bounds__(256) void k_nu_mark_selected(NuExc x, unsigned long long* __restrict__ mask,
                                                          unsigned long long* __restrict__ kept_total) {
    uint32_t kept = 0;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < x.n_dense; j += gridDim.x * blockDim.x) {
        const uint32_t i = x.dense[j];
        if (x.pick[i] == kNuUnpicked) continue;
        const uint64_t bit = (uint64_t)x.idx[i];
        atomicOr(&mask[bit >> 6], 1ull << (bit & 63));
        ++kept;
    }
    kept = wave_sum_u32(kept);
    if ((threadIdx.x & 63) == 0 && kept) atomicAdd(kept_total, (unsigned long long)kept);
}

// ---- launchers.  Device layout of the route's own buffers (all sized by the host):
//   exc      gs, ge, idx (cap words each, written by k_pm_prepare_sort), pick (cap words), key (64-bit, cap entries), then the
//            groups' counts (cap / 128 words, also the producer's), their scan, and the filled slots' indices (cap words)
//   state    8 words: [1] exceptions selected this round, [2] give-up flags, [3] selected in all, [4] suspects,
//            [6] earlier rounds that selected something
static NuExc nu_exc_view(uint32_t* exc, uint32_t cap, uint32_t n_dense, const uint32_t* n_over) {
    NuExc x;
    x.gs = exc; x.ge = exc + cap; x.idx = exc + 2 * (size_t)cap; x.pick = exc + 3 * (size_t)cap;
    x.key = reinterpret_cast<unsigned long long*>(exc + 4 * (size_t)cap);
    x.cnt = exc + 6 * (size_t)cap;
    x.cap = cap;
    x.goff = exc + 6 * (size_t)cap + cap / kNuGroup + 4;
    x.dense = x.goff + cap / kNuGroup + 4;
    x.n_dense = n_dense;
    x.n_over = n_over;
    return x;
}
uint32_t* nu_exc_counts(uint32_t* exc, uint32_t cap) { return exc + 6 * (size_t)cap; }
size_t nu_exc_bytes(uint32_t cap) { return ((size_t)cap * 7 + 2 * (cap / kNuGroup + 4) + 8) * sizeof(uint32_t); }
void launch_nu_count_span(hipStream_t st, const uint32_t* starts, const uint32_t* ends, uint32_t n, uint32_t span, uint32_t* out) {
    hipLaunchKernelGGL(k_nu_count_span, dim3(grid_for(n, 256)), dim3(256), 0, st, starts, ends, n, span, out);
}
void launch_nu_setup(hipStream_t st, uint32_t* exc, uint32_t cap, uint32_t n_exc, const uint32_t* n_over,
                     const uint32_t* boff, uint32_t ltot, uint32_t ell, uint32_t M, uint32_t* ce /* ltot + 3 words */,
                     uint32_t* spine, int32_t* nadj /* ltot + 1 */, uint32_t* state) {
    const NuExc x = nu_exc_view(exc, cap, n_exc, n_over);
    (void)hipMemsetAsync(ce, 0, ((size_t)ltot + 3) * sizeof(uint32_t), st);
    (void)hipMemsetAsync(state, 0, 8 * sizeof(uint32_t), st);
    launch_exclusive_scan(st, x.cnt, (cap - kNuOverflow) / kNuGroup, x.goff, spine, true);
    hipLaunchKernelGGL(k_nu_exc_diff, dim3(grid_for(cap ? cap : 1, 256)), dim3(256), 0, st, x, ce);
    launch_exclusive_scan(st, ce, ltot + 2, ce, spine, false);
    hipLaunchKernelGGL(k_nu_need_adjust, dim3(grid_for((uint64_t)ltot + 1, 256)), dim3(256), 0, st, boff, ce, ltot, ell, M, nadj);
    hipLaunchKernelGGL(k_nu_prepick, dim3(grid_for(cap ? cap : 1, 256)), dim3(256), 0, st, x, boff, ce, ell, M, nadj, state);
}
size_t nu_suspect_bytes(uint32_t suspects_cap) {  // the list, its slots by cell, the cells' counts and starts
    return (size_t)suspects_cap * (sizeof(uint2) + 3 * sizeof(uint32_t)) + 2 * ((size_t)kNuMaxCells + 3) * sizeof(uint32_t);
}
size_t nu_cells_bytes() { return ((size_t)kNuMaxCells + 3) * sizeof(uint32_t); }  // one word per cell (the dirty flags)
void launch_nu_round(hipStream_t st, uint32_t* exc, uint32_t cap, uint32_t n_exc, const uint32_t* n_over, bool first_round,
                     const uint32_t* boff, const uint32_t* selend, int32_t* nadj, const uint32_t* ce, const uint64_t* d_poff, uint32_t n_contigs,
                     uint32_t ell, uint32_t M, uint2* suspects, uint32_t suspects_cap, uint32_t* state,
                     unsigned long long* viol_key, uint32_t* viol_idx, const uint32_t* swept_from, uint32_t* sweep_from_next,
                     uint32_t ltot, uint32_t* spine, const uint32_t* seg_exact, uint32_t n_cand, uint32_t* marks_next,
                     uint32_t* selend_prev, uint32_t* dirty, uint32_t* dirty_next, const uint32_t* seg_fine, uint32_t* fine_next) {
    const NuExc x = nu_exc_view(exc, cap, n_exc, n_over);
    NuView v;
    v.boff = boff; v.selend = selend; v.nadj = nadj; v.poff = d_poff; v.n_contigs = n_contigs; v.ell = ell; v.M = M;
    v.ce = ce;
    NuBins bins;
    bins.shift = nu_bin_shift(ltot);
    bins.n_cells = (uint32_t)(((uint64_t)ltot >> bins.shift) + 1);
    bins.sorted = reinterpret_cast<uint32_t*>(suspects + suspects_cap);
    uint32_t* const replay_list = bins.sorted + suspects_cap;
    uint32_t* const far_list = replay_list + suspects_cap;
    bins.count = far_list + suspects_cap;
    bins.start = bins.count + kNuMaxCells + 3;
    bins.dirty = first_round ? nullptr : dirty;
    bins.dirty_next = dirty_next;
    if (selend_prev != nullptr) {
        hipLaunchKernelGGL(k_nu_diff, dim3(grid_for(ltot ? ltot : 1, 256)), dim3(256), 0, st, selend, selend_prev, ltot, bins.shift,
                           bins.n_cells, dirty, first_round ? 1u : 0u);
        (void)hipMemsetAsync(dirty_next, 0, ((size_t)bins.n_cells + 1) * sizeof(uint32_t), st);
    } else {
        bins.dirty = nullptr;
        bins.dirty_next = nullptr;
    }
    hipLaunchKernelGGL(k_nu_round_reset, dim3((n_contigs + 255) / 256), dim3(256), 0, st, state, viol_key, viol_idx, sweep_from_next, n_contigs,
                       first_round ? 1u : 0u);
    (void)hipMemsetAsync(bins.count, 0, ((size_t)bins.n_cells + 1) * sizeof(uint32_t), st);
    if (seg_exact != nullptr) (void)hipMemsetAsync(marks_next, 0, (size_t)n_cand * sizeof(uint32_t), st);
    if (seg_exact != nullptr && seg_fine != nullptr) (void)hipMemsetAsync(fine_next, 0, (size_t)n_cand * sizeof(uint32_t), st);
    hipLaunchKernelGGL(k_nu_verify, dim3(grid_for(n_exc ? n_exc : 1, 256)), dim3(256), 0, st, x, v, suspects, suspects_cap, state,
                       swept_from, bins, replay_list);
    launch_exclusive_scan(st, bins.count, bins.n_cells + 1, bins.start, spine, false);
    (void)hipMemsetAsync(bins.count, 0, ((size_t)bins.n_cells + 1) * sizeof(uint32_t), st);
    hipLaunchKernelGGL(k_nu_bin, dim3(64), dim3(256), 0, st, x, suspects, suspects_cap, state, bins);
    {
        const size_t lds = 6 * (size_t)nu_stage_entries(ell, false) * sizeof(uint32_t);  // (ell <= 256: the sweeps' limit)
        const size_t lds_far = 6 * (size_t)nu_stage_entries(ell, true) * sizeof(uint32_t);
        (void)hipFuncSetAttribute((const void*)k_nu_replay<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_far);
        hipLaunchKernelGGL(k_nu_replay<false>, dim3(2304), dim3(64), lds, st, x, v, suspects, suspects_cap, state, viol_key, swept_from,
                           bins, replay_list, far_list);
        hipLaunchKernelGGL(k_nu_replay<true>, dim3(256), dim3(64), lds_far, st, x, v, suspects, suspects_cap, state, viol_key, swept_from,
                           bins, far_list, (uint32_t*)nullptr);
    }
    hipLaunchKernelGGL(k_nu_select_apply, dim3(128), dim3(256), 0, st, x, suspects, suspects_cap, state, viol_key, nadj, d_poff, ell,
                       sweep_from_next, seg_exact, marks_next, bins, seg_exact != nullptr ? seg_fine : nullptr, n_cand, fine_next);
}
void launch_nu_mark_selected(hipStream_t st, uint32_t* exc, uint32_t cap, uint32_t n_exc, const uint32_t* n_over,
                             unsigned long long* mask, unsigned long long* kept_total) {
    const NuExc x = nu_exc_view(exc, cap, n_exc, n_over);
    hipLaunchKernelGGL(k_nu_mark_selected, dim3(grid_for(n_exc ? n_exc : 1, 256)), dim3(256), 0, st, x, mask, kept_total);
}
