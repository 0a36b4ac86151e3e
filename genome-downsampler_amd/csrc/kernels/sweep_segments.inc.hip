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
// the host asks for at most kMaxSweepWindows windows and splits only calls of fewer than 256 contigs
static_assert(kMaxSweepWindows + 256 <= (uint32_t)kSegMaxCandidates, "stretch tables: windows + contigs must fit k_build_segments' LDS arrays");
// k_build_segments declares five arrays of kSegMaxCandidates words (~80 KiB of static LDS): fine on gfx950's
// 160 KiB per workgroup, beyond the 64 KiB of gfx90a / gfx942 -- this library is written for gfx950 only
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "qmcp kernels are written for gfx950 (MI355X): k_build_segments alone needs more than 64 KiB of LDS per workgroup"
#endif

__global__ __launch_bounds__(256) void k_find_cuts(const uint32_t* __restrict__ boff,
                                                   const uint32_t* __restrict__ eoff,  // null: one span, ell
                                                   const uint64_t* __restrict__ contig_pos_off,
                                                   uint32_t n_contigs, uint32_t ltot, uint32_t ell, uint32_t M,
                                                   uint32_t win, uint32_t* __restrict__ cut,
                                                   const uint32_t* __restrict__ other_cov /* near-uniform route: reads that are
                                                       not in boff (the listed exceptions) covering position q - 1 =
                                                       other_cov[q]; a cut needs the coverage of ALL reads <= M.  Or null */) {
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
                const uint32_t cov = boff[q] - (eoff != nullptr ? eoff[q - 1] : boff[q >= ell ? q - ell : 0u]) +
                                     (other_cov != nullptr ? other_cov[q] : 0u);
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
static constexpr int kSegPerThread = kSegMaxCandidates / kSegThreads;  // consecutive windows a thread looks after

// exclusive scan over the workgroup of kSegPerThread flags per thread (thread t owns entries t * kSegPerThread ...):
// pre[j] = flags set before entry j; returns the total.  s_ws: one word per wave.
__device__ __forceinline__ uint32_t seg_block_scan(const bool (&flag)[kSegPerThread], uint32_t (&pre)[kSegPerThread],
                                                   uint32_t* __restrict__ s_ws) {
    uint32_t local = 0;
#pragma unroll
    for (int j = 0; j < kSegPerThread; ++j) { pre[j] = local; local += flag[j] ? 1u : 0u; }
    const uint32_t inc = wave_incl_scan_add(local);
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    __syncthreads();  // (s_ws may still be read from the scan before)
    if (lane == 63) s_ws[w] = inc;
    __syncthreads();
    uint32_t before = inc - local, total = 0;
    for (uint32_t x = 0; x < (uint32_t)(kSegThreads / 64); ++x) {
        const uint32_t v = s_ws[x];
        before += x < w ? v : 0u;
        total += v;
    }
#pragma unroll
    for (int j = 0; j < kSegPerThread; ++j) pre[j] += before;
    return total;
}

// The window candidates come in position order (window w's cut, or its speculative boundary, lies in window w),
// so ranks are prefix counts over the windows plus a look at the (< 256) contig starts: no all-pairs compare.
__global__ __launch_bounds__(kSegThreads) void k_build_segments(const uint32_t* __restrict__ cut,
                                                                uint32_t n_windows,
                                                                const uint64_t* __restrict__ contig_pos_off,
                                                                uint32_t n_contigs, uint32_t ltot,
                                                                uint32_t win, uint32_t burn, uint32_t stride,
                                                                uint32_t* __restrict__ seg,
                                                                uint32_t* __restrict__ n_speculative /* = ; or null */) {
    __shared__ uint32_t s_wexact[kSegMaxCandidates];      // window w's real cut (kNoCut: none)
    __shared__ uint32_t s_wpos[kSegMaxCandidates];        // window w's boundary: the cut, a speculative one, or kNoCut
    __shared__ uint32_t s_pe[kSegMaxCandidates + 1];      // real cuts in the windows before w
    __shared__ uint32_t s_pv[kSegMaxCandidates + 1];      // boundaries in the windows before w
    __shared__ uint32_t s_sorted[kSegMaxCandidates];
    __shared__ uint32_t s_cpos[257];                      // contig starts (+ the end of the last)
    __shared__ uint32_t s_ws[kSegThreads / 64];
    __shared__ uint32_t s_spec;
    const uint32_t n_cand = n_contigs + n_windows;
    const uint32_t w0 = threadIdx.x * (uint32_t)kSegPerThread;
    if (threadIdx.x <= n_contigs && threadIdx.x < 257) s_cpos[threadIdx.x] = (uint32_t)contig_pos_off[threadIdx.x];
    if (threadIdx.x == 0) s_spec = 0;
    uint32_t ex[kSegPerThread];
    bool flag[kSegPerThread];
#pragma unroll
    for (int j = 0; j < kSegPerThread; ++j) {
        const uint32_t w = w0 + (uint32_t)j;
        ex[j] = w < n_windows ? cut[w] : kNoCut;
        s_wexact[w] = ex[j];
        flag[j] = ex[j] != kNoCut;
    }
    uint32_t pre[kSegPerThread];
    const uint32_t total_exact = seg_block_scan(flag, pre, s_ws);  // (its first barrier publishes s_cpos, s_wexact)
#pragma unroll
    for (int j = 0; j < kSegPerThread; ++j) s_pe[w0 + (uint32_t)j] = pre[j];
    if (threadIdx.x == 0) s_pe[kSegMaxCandidates] = total_exact;
    __syncthreads();
    // non-empty contig starts before / up to a position
    auto contigs_before = [&](uint32_t p, bool inclusive) {
        uint32_t k = 0;
        for (uint32_t c = 0; c < n_contigs; ++c) {
            const uint32_t a = s_cpos[c];
            k += (s_cpos[c + 1] > a && (inclusive ? a <= p : a < p)) ? 1u : 0u;
        }
        return k;
    };
    // speculative boundaries: window w's first position, with a run-in that no exact boundary may fall into
    uint32_t pos[kSegPerThread];
    bool spec[kSegPerThread];
    uint32_t my_spec = 0;
#pragma unroll
    for (int j = 0; j < kSegPerThread; ++j) {
        const uint32_t w = w0 + (uint32_t)j;
        pos[j] = ex[j];
        spec[j] = false;
        if (w < n_windows && ex[j] == kNoCut && burn != 0 && w >= 1 && w % stride == 0 && (uint64_t)w * win < ltot) {
            const uint32_t b = w * win;
            bool ok = b >= burn;
            if (ok) {
                const uint32_t lo = b - burn;        // an exact boundary q with lo < q <= b spoils it
                const uint32_t wlo = lo / win;        // (window w itself holds no cut)
                if (wlo < w) {
                    if (s_pe[w] - s_pe[wlo + 1] != 0) ok = false;              // windows wholly inside (lo, b)
                    const uint32_t q = s_wexact[wlo];
                    if (q != kNoCut && q > lo) ok = false;
                }
                for (uint32_t c = 0; c < n_contigs; ++c) {
                    const uint32_t q = s_cpos[c];
                    if (s_cpos[c + 1] > q && q <= b && q > lo) ok = false;
                }
            }
            if (ok) { pos[j] = b; spec[j] = true; ++my_spec; }
        }
        s_wpos[w] = pos[j];
        flag[j] = pos[j] != kNoCut;
    }
    if (my_spec != 0) atomicAdd(&s_spec, my_spec);
    const uint32_t total_windows = seg_block_scan(flag, pre, s_ws);  // (publishes s_wpos)
#pragma unroll
    for (int j = 0; j < kSegPerThread; ++j) s_pv[w0 + (uint32_t)j] = pre[j];
    if (threadIdx.x == 0) s_pv[kSegMaxCandidates] = total_windows;
    __syncthreads();
    const uint32_t count = total_windows + contigs_before(0xFFFFFFFFu, true);
    // ranks: window boundaries, then (threads below n_contigs) the contig starts
    uint32_t rank[kSegPerThread];
#pragma unroll
    for (int j = 0; j < kSegPerThread; ++j) {
        rank[j] = 0;
        if (pos[j] != kNoCut) {
            rank[j] = s_pv[w0 + (uint32_t)j] + contigs_before(pos[j], false);
            s_sorted[rank[j]] = pos[j];
        }
    }
    uint32_t c_pos = kNoCut, c_rank = 0, c_wc = 0;
    if (threadIdx.x < n_contigs && s_cpos[threadIdx.x + 1] > s_cpos[threadIdx.x]) {
        c_pos = s_cpos[threadIdx.x];
        c_wc = min(c_pos / win, n_windows);  // the window the start lies in (n_windows: none)
        const uint32_t in_window = (c_wc < n_windows && s_wpos[c_wc] != kNoCut && s_wpos[c_wc] < c_pos) ? 1u : 0u;
        c_rank = contigs_before(c_pos, false) + (c_wc < n_windows ? s_pv[c_wc] : total_windows) + in_window;
        s_sorted[c_rank] = c_pos;
    }
    __syncthreads();
    auto emit = [&](uint32_t m, uint32_t r, bool is_spec, uint32_t exact_upto) {
        uint32_t cend = ltot;
        for (uint32_t c = 0; c < n_contigs; ++c) {
            const uint32_t a = s_cpos[c], b = s_cpos[c + 1];
            if (a <= m && m < b) cend = b;
        }
        const uint32_t next = r + 1 < count ? s_sorted[r + 1] : ltot;
        seg[1 + 3 * r + 0] = is_spec ? m - burn : m;
        seg[1 + 3 * r + 1] = min(next, cend);
        seg[1 + 3 * r + 2] = cend;
        seg[1 + 3 * n_cand + r] = m;
        // which stretch of the EXACT table (contig starts and cut points only) this one lies in: what a
        // disagreement marks for another go, and what the later tiers look up to see whether they have work
        seg[1 + 4 * n_cand + r] = exact_upto + contigs_before(m, true) - 1u;  // (>= 1: every position lies behind its contig's start)
        if (r == 0) seg[0] = count;
    };
#pragma unroll
    for (int j = 0; j < kSegPerThread; ++j)
        if (pos[j] != kNoCut)  // real cuts up to it: those of the windows before, and its own if it is one
            emit(pos[j], rank[j], spec[j], s_pe[w0 + (uint32_t)j] + (ex[j] != kNoCut ? 1u : 0u));
    if (c_pos != kNoCut) {
        const uint32_t own = (c_wc < n_windows && s_wexact[c_wc] != kNoCut && s_wexact[c_wc] <= c_pos) ? 1u : 0u;
        emit(c_pos, c_rank, false, (c_wc < n_windows ? s_pe[c_wc] : total_exact) + own);
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
                                                     uint32_t* __restrict__ redo_out,
                                                     const uint32_t* __restrict__ own_marks /* or null; else per stretch of
                                                         this table: 1 = swept this time.  A boundary is looked at when the
                                                         stretch on EITHER side of it was: the other side's output is the
                                                         sweep before's, and equal counts there say that what lies behind
                                                         the boundary has not changed either */) {
    const uint32_t r = blockIdx.x;
    if (r >= seg[0] || spec_stretch_idle(seg, n_cand, r, redo_in)) return;
    if (own_marks != nullptr && own_marks[r] == 0 && (r == 0 || own_marks[r - 1] == 0)) return;
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
