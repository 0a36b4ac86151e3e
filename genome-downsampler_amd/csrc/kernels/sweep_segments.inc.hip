// sweep_segments -- cut points: where a contig's sweep can be split exactly.
//
// A position p with cov(p) <= M forces every read covering p to be kept, so in
// the sweep's terms the distances d are equal over [p - ell, p] and nothing behind p matters to
// what follows: the sweep may start afresh at p + 1 with the initial state of a contig's start
// (d == 0, h = ex of the first block -- with the coverage counted from the true prefix sums, reads
// from before p + 1 included).  Deep data has no such positions; shallow or gapped data (exomes,
// amplicon panels, low-pass genomes) is full of them, and there a contig's serial chain becomes many.
//
// k_find_cuts: the position axis is cut into windows; every window reports the first position q in
// it at which a stretch may start (cov(q - 1) <= M, q not a contig's first position).
// k_build_segments: contig starts and window cuts, merged in position order, become the table the
// sweep kernels index by workgroup: [count, then {start, end, contig end} per stretch].

static constexpr uint32_t kNoCut = 0xFFFFFFFFu;
static constexpr int kSegMaxCandidates = 4096;  // contigs + windows (the one-span sweeps ask for at most 1024)

__global__ __launch_bounds__(256) void k_find_cuts(const uint32_t* __restrict__ boff,
                                                   const uint32_t* __restrict__ eoff,  // null: one span, ell
                                                   const uint64_t* __restrict__ contig_pos_off,
                                                   uint32_t n_contigs, uint32_t ltot, uint32_t ell, uint32_t M,
                                                   uint32_t win, uint32_t* __restrict__ cut) {
    __shared__ uint32_t s_first;
    __shared__ uint32_t s_cstart[256];  // contig starts (fewer than 256 contigs when this runs)
    const uint32_t w = blockIdx.x;
    const uint32_t lo = max(w * win, 1u);
    const uint32_t hi = (uint32_t)min((uint64_t)(w + 1) * win, (uint64_t)ltot);
    if (threadIdx.x == 0) s_first = kNoCut;
    if (threadIdx.x < n_contigs) s_cstart[threadIdx.x] = (uint32_t)contig_pos_off[threadIdx.x];
    __syncthreads();
    // sixteen strips of blockDim.x positions per round (all their loads in flight before the round's barrier: a
    // window without a cut is scanned to its end, 244 k positions on a genome-sized share): the first
    // position wins, whichever strip it is in
    for (uint32_t q0 = lo; q0 < hi; q0 += 16 * blockDim.x) {
#pragma unroll
        for (uint32_t k = 0; k < 16; ++k) {
            const uint32_t q = q0 + k * blockDim.x + threadIdx.x;
            if (q < hi) {
                // coverage of position q - 1: starts up to it minus ends before it
                const uint32_t cov = boff[q] - (eoff != nullptr ? eoff[q - 1] : boff[q >= ell ? q - ell : 0u]);
                if (cov <= M) {
                    bool contig_start = false;
                    for (uint32_t c = 0; c < n_contigs; ++c) contig_start |= s_cstart[c] == q;
                    if (!contig_start) atomicMin(&s_first, q);
                }
            }
        }
        __syncthreads();
        if (s_first != kNoCut) break;  // uniform: read after the barrier
        __syncthreads();
    }
    if (threadIdx.x == 0) cut[w] = s_first;
}

// Speculative boundaries (burn != 0).  On data a few times deeper than M cut points are rare, but the
// greedy FORGETS where it started: two sweeps of the same positions from different states select the
// same reads from some point on (measured: within 25-60 blocks at a depth of 2 x M, hundreds at 4 x M,
// never on deep data), because the state is only the kept counts of the last ell start positions and the
// sparse counts keep forcing it.  So a window without a cut still gets a boundary at its first position b:
// its stretch starts `burn` positions earlier from the state of a cut point (every read over the start
// kept) and OWNS the positions from b on (only every stride-th window is a candidate, so that stretches stay
// several run-ins long); the stretch before it runs up to b as before.  The two have
// then both computed [b - burn, b) -- a stretch's run-in is stored apart from what it owns -- and
// k_spec_verify compares the last ell positions before b: equal kept counts there are equal states, so
// everything the speculative stretch selected from b on is what the serial sweep selects.  A mismatch
// marks the part of the genome it lies in for another go (see "Tiers" below).
// Table: [count, {start, end, contig end} per stretch, then the position each stretch owns from, then the
// index of the exact table's stretch it lies in].
static constexpr int kSegThreads = 1024;
static constexpr int kSegPerThread = kSegMaxCandidates / kSegThreads;  // candidates a thread looks after

__global__ __launch_bounds__(kSegThreads) void k_build_segments(const uint32_t* __restrict__ cut,
                                                                uint32_t n_windows,
                                                                const uint64_t* __restrict__ contig_pos_off,
                                                                uint32_t n_contigs, uint32_t ltot,
                                                                uint32_t win, uint32_t burn, uint32_t stride,
                                                                uint32_t* __restrict__ seg,
                                                                uint32_t* __restrict__ n_speculative /* = ; or null */) {
    __shared__ __attribute__((aligned(16))) uint32_t s_pos[kSegMaxCandidates];
    __shared__ __attribute__((aligned(16))) uint32_t s_exact[kSegMaxCandidates];
    __shared__ uint32_t s_sorted[kSegMaxCandidates];
    __shared__ uint32_t s_count, s_spec;
    __shared__ uint32_t s_cpos[257];  // contig starts (+ the end of the last): read once, not in every thread's loops
    const uint32_t n_cand = n_contigs + n_windows;
    const uint32_t n_pad = (n_cand + 3u) & ~3u;  // the loops below read four candidates at a time
    if (threadIdx.x <= n_contigs && threadIdx.x < 257) s_cpos[threadIdx.x] = (uint32_t)contig_pos_off[threadIdx.x];
    if (threadIdx.x == 0) { s_count = 0; s_spec = 0; }
    __syncthreads();
    // candidate t = threadIdx.x + j * kSegThreads: contig starts first, then the windows
    uint32_t mine[kSegPerThread];
    bool spec[kSegPerThread];
#pragma unroll
    for (int j = 0; j < kSegPerThread; ++j) {
        const uint32_t t = threadIdx.x + (uint32_t)j * kSegThreads;
        mine[j] = kNoCut;
        spec[j] = false;
        if (t < n_contigs) {
            if (s_cpos[t + 1] > s_cpos[t]) mine[j] = s_cpos[t];  // empty contigs: no work
        } else if (t < n_cand) {
            const uint32_t w = t - n_contigs;
            mine[j] = cut[w];
            if (mine[j] == kNoCut && burn != 0 && w >= 1 && w % stride == 0 && (uint64_t)w * win < ltot) {
                mine[j] = w * win;
                spec[j] = true;
            }
        }
        if (t < n_pad) {
            s_pos[t] = spec[j] ? kNoCut : mine[j];  // the exact boundaries first
            s_exact[t] = spec[j] ? kNoCut : mine[j];
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kSegPerThread; ++j) {
        if (spec[j]) {
            // the run-in must lie inside one stretch: no exact boundary in (mine - burn, mine]
            const uint32_t m = mine[j];
            bool ok = m >= burn;
            for (uint32_t k = 0; k < n_pad; k += 4) {
                const uint4 v = *reinterpret_cast<const uint4*>(&s_exact[k]);
                const uint32_t q4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (q4[i] != kNoCut && q4[i] <= m && q4[i] + burn > m) ok = false;
            }
            if (!ok) { mine[j] = kNoCut; spec[j] = false; }
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kSegPerThread; ++j) {
        const uint32_t t = threadIdx.x + (uint32_t)j * kSegThreads;
        if (t < n_pad) s_pos[t] = mine[j];
    }
    __syncthreads();
    uint32_t rank[kSegPerThread];
#pragma unroll
    for (int j = 0; j < kSegPerThread; ++j) {
        rank[j] = 0;
        if (mine[j] != kNoCut) {
            // candidates are distinct: non-empty contigs start at distinct positions, windows are
            // disjoint, a cut is never a contig's first position, and a speculative boundary has no exact one
            // within `burn` before it
            // (four candidates per LDS read: entries from n_cand on hold kNoCut, which is below nothing)
            const uint32_t m = mine[j];
            uint32_t r = 0;
#pragma unroll 4
            for (uint32_t k = 0; k < n_pad; k += 4) {
                const uint4 v = *reinterpret_cast<const uint4*>(&s_pos[k]);
                r += (v.x < m ? 1u : 0u) + (v.y < m ? 1u : 0u) + (v.z < m ? 1u : 0u) + (v.w < m ? 1u : 0u);
            }
            rank[j] = r;
            s_sorted[r] = m;
            atomicAdd(&s_count, 1u);
            if (spec[j]) atomicAdd(&s_spec, 1u);
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kSegPerThread; ++j) {
        if (mine[j] != kNoCut) {
            const uint32_t m = mine[j], r = rank[j];
            const uint32_t count = s_count;
            uint32_t cend = ltot;
            for (uint32_t c = 0; c < n_contigs; ++c) {
                const uint32_t a = s_cpos[c], b = s_cpos[c + 1];
                if (a <= m && m < b) cend = b;
            }
            const uint32_t next = r + 1 < count ? s_sorted[r + 1] : ltot;
            seg[1 + 3 * r + 0] = spec[j] ? m - burn : m;
            seg[1 + 3 * r + 1] = min(next, cend);
            seg[1 + 3 * r + 2] = cend;
            seg[1 + 3 * n_cand + r] = m;
            // which stretch of the EXACT table (contig starts and cut points only) this one lies in: what a
            // disagreement marks for another go, and what the later tiers look up to see whether they have work
            uint32_t exact_before = 0;
#pragma unroll 4
            for (uint32_t k = 0; k < n_pad; k += 4) {  // (kNoCut is the largest value: never <= a position)
                const uint4 v = *reinterpret_cast<const uint4*>(&s_exact[k]);
                exact_before += (v.x <= m ? 1u : 0u) + (v.y <= m ? 1u : 0u) + (v.z <= m ? 1u : 0u) + (v.w <= m ? 1u : 0u);
            }
            seg[1 + 4 * n_cand + r] = exact_before - 1u;  // (>= 1: every position lies behind its contig's start)
            if (r == 0) seg[0] = count;
        }
    }
    if (threadIdx.x == 0 && n_speculative != nullptr) *n_speculative = s_spec;
}

// Tiers.  A disagreement marks the EXACT stretch (between two cut points / contig starts) it lies in:
// `redo_out[x] = 1`.  The next tier's kernels -- sweep, check, merge -- look their stretch's exact stretch up
// in `redo_in` and return at once when it is not marked, so only the marked parts of the genome are swept
// again (first speculatively with a longer run-in, at last exactly), and everything is queued at once.
__device__ __forceinline__ bool spec_stretch_idle(const uint32_t* __restrict__ seg, uint32_t n_cand, uint32_t r,
                                                  const uint32_t* __restrict__ redo_in) {
    return redo_in != nullptr && redo_in[seg[1 + 4 * n_cand + r]] == 0;
}

// one workgroup per stretch: a speculative one compares the last ell positions of its run-in (`run_in`: where
// the sweep stored them) with what the stretch that owns those positions stored in `owned`
__global__ __launch_bounds__(256) void k_spec_verify(const uint32_t* __restrict__ seg, uint32_t n_cand, uint32_t ell,
                                                     const uint32_t* __restrict__ owned,
                                                     const uint32_t* __restrict__ run_in,
                                                     uint32_t* __restrict__ mismatches,
                                                     const uint32_t* __restrict__ redo_in /* or null: every stretch */,
                                                     uint32_t* __restrict__ redo_out) {
    const uint32_t r = blockIdx.x;
    if (r >= seg[0] || spec_stretch_idle(seg, n_cand, r, redo_in)) return;
    const uint32_t start = seg[1 + 3 * r], own = seg[1 + 3 * n_cand + r];
    if (own == start) return;  // an exact boundary
    bool differs = false;
    for (uint32_t i = threadIdx.x; i < ell; i += blockDim.x) differs |= run_in[own - ell + i] != owned[own - ell + i];
    if (__syncthreads_or(differs ? 1 : 0) && threadIdx.x == 0) {
        atomicAdd(mismatches, 1u);
        redo_out[seg[1 + 4 * n_cand + r]] = 1u;
    }
}

// Mixed spans (the register-resident event sweep): a bucket keeps giving reads while any of them is alive,
// so what the stretch before a speculative boundary left in its output for the last W = max_span start
// positions is the count "so far", and the speculative stretch recorded its own counts so far when it
// reached the boundary (k_sweep_general_reg, `snap`): equal counts are equal states of the walk.
__global__ __launch_bounds__(256) void k_spec_verify_mixed(const uint32_t* __restrict__ seg, uint32_t n_cand, uint32_t W,
                                                           const uint32_t* __restrict__ out_even,
                                                           const uint32_t* __restrict__ out_odd,
                                                           const uint32_t* __restrict__ snap, uint32_t snap_words,
                                                           uint32_t* __restrict__ mismatches,
                                                           const uint32_t* __restrict__ redo_in,
                                                           uint32_t* __restrict__ redo_out) {
    const uint32_t r = blockIdx.x;
    if (r >= seg[0] || spec_stretch_idle(seg, n_cand, r, redo_in)) return;
    const uint32_t start = seg[1 + 3 * r], own = seg[1 + 3 * n_cand + r];
    if (own == start) return;  // an exact boundary
    const uint32_t* prev = (r & 1u) ? out_even : out_odd;
    const uint32_t* mine = snap + (size_t)r * snap_words;
    bool differs = false;
    for (uint32_t i = threadIdx.x; i < W; i += blockDim.x) {
        const uint32_t p = own - W + i;
        differs |= mine[p % snap_words] != prev[p];
    }
    if (__syncthreads_or(differs ? 1 : 0) && threadIdx.x == 0) {
        atomicAdd(mismatches, 1u);
        redo_out[seg[1 + 4 * n_cand + r]] = 1u;
    }
}

// ... and the final counts of those W positions are the speculative stretch's: an odd stretch's range
// moves to the even output from W before the position it owns from (if it is speculative) up to W
// before its end (if the next one is)
__global__ __launch_bounds__(256) void k_spec_merge_mixed(const uint32_t* __restrict__ seg, uint32_t n_cand, uint32_t W,
                                                          uint32_t* __restrict__ out_even,
                                                          const uint32_t* __restrict__ out_odd,
                                                          const uint32_t* __restrict__ redo_in) {
    const uint32_t r = blockIdx.x;
    const uint32_t count = seg[0];
    if ((r & 1u) == 0 || r >= count || spec_stretch_idle(seg, n_cand, r, redo_in)) return;
    const uint32_t start = seg[1 + 3 * r], own = seg[1 + 3 * n_cand + r];
    uint32_t from = own != start ? own - W : own;
    uint32_t end = seg[1 + 3 * r + 1];
    if (r + 1 < count) {
        const uint32_t nstart = seg[1 + 3 * (r + 1)], nown = seg[1 + 3 * n_cand + r + 1];
        if (nown != nstart && nown == end) end -= W;
    }
    for (uint32_t p = from + blockIdx.y * blockDim.x + threadIdx.x; p < end; p += gridDim.y * blockDim.x) out_even[p] = out_odd[p];
}
