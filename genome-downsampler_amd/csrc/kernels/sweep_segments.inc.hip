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
static constexpr int kSegMaxCandidates = 1024;  // contigs + windows

__global__ __launch_bounds__(256) void k_find_cuts(const uint32_t* __restrict__ boff,
                                                   const uint32_t* __restrict__ eoff,  // null: one span, ell
                                                   const uint64_t* __restrict__ contig_pos_off,
                                                   uint32_t n_contigs, uint32_t ltot, uint32_t ell, uint32_t M,
                                                   uint32_t win, uint32_t* __restrict__ cut) {
    __shared__ uint32_t s_first;
    const uint32_t w = blockIdx.x;
    const uint32_t lo = max(w * win, 1u);
    const uint32_t hi = (uint32_t)min((uint64_t)(w + 1) * win, (uint64_t)ltot);
    if (threadIdx.x == 0) s_first = kNoCut;
    __syncthreads();
    // four strips of 256 positions per round: the first position wins, whichever strip it is in
    for (uint32_t q0 = lo; q0 < hi; q0 += 1024) {
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
            const uint32_t q = q0 + k * 256 + threadIdx.x;
            if (q < hi) {
                // coverage of position q - 1: starts up to it minus ends before it
                const uint32_t cov = boff[q] - (eoff != nullptr ? eoff[q - 1] : boff[q >= ell ? q - ell : 0u]);
                if (cov <= M) {
                    bool contig_start = false;
                    for (uint32_t c = 0; c < n_contigs; ++c) contig_start |= (uint32_t)contig_pos_off[c] == q;
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

__global__ __launch_bounds__(kSegMaxCandidates) void k_build_segments(const uint32_t* __restrict__ cut,
                                                                      uint32_t n_windows,
                                                                      const uint64_t* __restrict__ contig_pos_off,
                                                                      uint32_t n_contigs, uint32_t ltot,
                                                                      uint32_t* __restrict__ seg) {
    __shared__ uint32_t s_pos[kSegMaxCandidates];
    __shared__ uint32_t s_sorted[kSegMaxCandidates];
    __shared__ uint32_t s_count;
    const uint32_t t = threadIdx.x;
    uint32_t mine = kNoCut;
    if (t < n_contigs) {
        if (contig_pos_off[t + 1] > contig_pos_off[t]) mine = (uint32_t)contig_pos_off[t];  // empty contigs: no work
    } else if (t - n_contigs < n_windows) {
        mine = cut[t - n_contigs];
    }
    s_pos[t] = mine;
    if (t == 0) s_count = 0;
    __syncthreads();
    uint32_t rank = 0;
    if (mine != kNoCut) {
        // candidates are distinct: non-empty contigs start at distinct positions, windows are
        // disjoint, and a cut is never a contig's first position
        for (uint32_t k = 0; k < n_contigs + n_windows; ++k) rank += s_pos[k] < mine ? 1u : 0u;
        s_sorted[rank] = mine;
        atomicAdd(&s_count, 1u);
    }
    __syncthreads();
    if (mine != kNoCut) {
        const uint32_t count = s_count;
        uint32_t cend = ltot;
        for (uint32_t c = 0; c < n_contigs; ++c) {
            const uint32_t a = (uint32_t)contig_pos_off[c], b = (uint32_t)contig_pos_off[c + 1];
            if (a <= mine && mine < b) cend = b;
        }
        const uint32_t next = rank + 1 < count ? s_sorted[rank + 1] : ltot;
        seg[1 + 3 * rank + 0] = mine;
        seg[1 + 3 * rank + 1] = min(next, cend);
        seg[1 + 3 * rank + 2] = cend;
        if (rank == 0) seg[0] = count;
    }
}
