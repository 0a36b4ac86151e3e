// mark_and_next_rows.inc.hip -- part of qmcp_kernels.hip (one translation unit; included inside namespace qmcp).
// ------------------------------------------------------------------ keep-mask emission
// sorted entry j (bucket = its start position) is kept iff j < selend[bucket].
// obtain_sequence counterpart (quasi_mcp_cpu_max_flow_solver.cpp:89-100).
// One thread per start position walks that bucket's selected prefix [boff[q], selend[q]) --
// at most M entries, usually 0..2 -- and sets the kept reads' bits: the sorted records of the
// other ~95 % of the reads are never touched.
template <typename Keys>
__global__ __launch_bounds__(256) void k_mark(Keys keys, uint32_t ltot,
                                              const uint32_t* __restrict__ boff,
                                              const uint32_t* __restrict__ selend,
                                              uint32_t* __restrict__ mask32,
                                              unsigned long long* __restrict__ n_kept) {
    const uint32_t stride = gridDim.x * blockDim.x;
    uint32_t mine = 0;
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < ltot; q += stride) {
        const uint32_t j0 = boff[q], j1 = selend[q];
        for (uint32_t j = j0; j < j1; ++j) {
            const uint32_t idx = keys.idx(j);
            atomicOr(&mask32[idx >> 5], 1u << (idx & 31));
        }
        mine += j1 - j0;
    }
    mine = wave_sum_u32(mine);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(n_kept, (unsigned long long)mine);
}

// cov[p] = #reads started at or before p - #reads ended before p
// (what BamApi::find_input_cover builds with per-base increments, bam_api.cpp:275-286)
__global__ __launch_bounds__(256) void k_coverage(const uint32_t* __restrict__ boff,
                                                  const uint32_t* __restrict__ eoff,
                                                  uint32_t ltot, uint32_t* __restrict__ cov) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < ltot; p += stride)
        cov[p] = boff[p + 1] - eoff[p];
}

// b and d of the reference's flow network from the coverage (stage probe): create_b_function's result
// after its cap loop (quasi_mcp_cpu_max_flow_solver.cpp:58-73) is b[0] = 0, b[p + 1] = min(cov[p], M);
// create_demand_function (:75-87) turns it, in place and ascending, into d[0] = -b[1],
// d[i] = b[i] - b[i + 1] for 1 <= i < n, and leaves d[n] = b[n] (its loop stops at i < n).
__global__ __launch_bounds__(256) void k_b_and_demand(const uint32_t* __restrict__ cov, uint32_t n, uint32_t M,
                                                      int32_t* __restrict__ b, int32_t* __restrict__ d) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += stride) {
        const int32_t bi = i == 0 ? 0 : (int32_t)min(cov[i - 1], M);
        const int32_t bn = i < n ? (int32_t)min(cov[i], M) : 0;  // b[i + 1]
        b[i] = bi;
        d[i] = i == 0 ? -bn : (i < n ? bi - bn : bi);
    }
}

// ------------------------------------------------------------------ "next" rows
// BamApi::find_pairs on the bitmask (bam_api.cpp:239-273): mates are (2q, 2q+1).
__global__ __launch_bounds__(256) void k_complete_pairs(uint64_t* __restrict__ mask,
                                                        uint32_t n_words, uint64_t n_reads) {
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint64_t even = 0x5555555555555555ull;
    for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += stride) {
        uint64_t m = mask[w];
        m |= ((m & even) << 1) | ((m >> 1) & even);
        // an unpaired trailing read (odd n_reads) has no mate: never set bits past n_reads
        const uint64_t first = (uint64_t)w * 64;
        if (first + 64 > n_reads) {
            const uint32_t live = (uint32_t)(n_reads - first);
            m &= live >= 64 ? ~0ull : ((1ull << live) - 1ull);
        }
        mask[w] = m;
    }
}

// Amplicon FILTER predicate per pair (bam_api.cpp:311-327, amplicon.cpp:5-7,
// amplicon_set.cpp:5-9); one wave emits one 64-pair word with a ballot.
__global__ __launch_bounds__(256) void k_amplicon_filter(const uint32_t* __restrict__ starts,
                                                         const uint32_t* __restrict__ ends,
                                                         const uint32_t* __restrict__ seq_lengths,
                                                         const uint32_t* __restrict__ qualities,
                                                         uint64_t n_pairs,
                                                         const uint32_t* __restrict__ amp_starts,
                                                         const uint32_t* __restrict__ amp_ends,
                                                         uint32_t n_amp, uint32_t min_length,
                                                         uint32_t min_mapq,
                                                         uint64_t* __restrict__ pair_keep) {
    extern __shared__ uint32_t s_amp[];  // [2 * n_cached]
    const uint32_t n_cached = min(n_amp, 4096u);
    for (uint32_t i = threadIdx.x; i < n_cached; i += blockDim.x) {
        s_amp[i] = amp_starts[i];
        s_amp[n_cached + i] = amp_ends[i];
    }
    __syncthreads();
    const uint64_t n_words = (n_pairs + 63) / 64;
    const uint64_t wave_global = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint32_t lane = threadIdx.x & 63;
    for (uint64_t w = wave_global; w < n_words; w += n_waves) {
        const uint64_t q = w * 64 + lane;
        bool ok = false;
        if (q < n_pairs) {
            const uint64_t i = 2 * q, j = i + 1;
            const uint32_t s1 = starts[i], e1 = ends[i], s2 = starts[j], e2 = ends[j];
            bool pass = true;
            if (qualities) pass = pass && qualities[i] >= min_mapq && qualities[j] >= min_mapq;
            if (seq_lengths) pass = pass && seq_lengths[i] >= min_length && seq_lengths[j] >= min_length;
            bool in_one = false;
            for (uint32_t a = 0; a < n_cached && !in_one; ++a) {
                const uint32_t as = s_amp[a], ae = s_amp[n_cached + a];
                in_one = as <= s1 && e1 <= ae && as <= s2 && e2 <= ae;
            }
            for (uint32_t a = n_cached; a < n_amp && !in_one; ++a) {
                const uint32_t as = amp_starts[a], ae = amp_ends[a];
                in_one = as <= s1 && e1 <= ae && as <= s2 && e2 <= ae;
            }
            ok = pass && in_one;
        }
        const uint64_t word = __ballot(ok);
        if (lane == 0) pair_keep[w] = word;
    }
}

// ------------------------------------------------------------------ filter -> solve pipeline glue
// Stream compaction of the pairs that survive the FILTER (pairs stay adjacent: survivor q'
// becomes reads 2q', 2q'+1) and the map back to original read indices -- the device-resident
// equivalent of what BamApi does while ingesting (bam_api.cpp:434-461: only accepted pairs are
// appended) and of the id bookkeeping around the solver in App::execute (src/app.cpp:134-142).
__global__ __launch_bounds__(256) void k_word_popcounts(const uint64_t* __restrict__ words,
                                                        uint32_t n_words,
                                                        uint32_t* __restrict__ counts) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += stride)
        counts[w] = __popcll(words[w]);
}

// keep mask -> ascending ReadIndex list (what obtain_sequence returns, quasi_mcp_cpu_max_flow_solver.cpp:89-100):
// word w writes its set bits from word_base[w] on (exclusive scan of the words' popcounts)
__global__ __launch_bounds__(256) void k_mask_to_indices(const uint64_t* __restrict__ mask, uint32_t n_words,
                                                         const uint32_t* __restrict__ word_base,
                                                         unsigned long long* __restrict__ out) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += stride) {
        uint64_t bits = mask[w];
        uint32_t dst = word_base[w];
        while (bits != 0) {
            out[dst++] = (unsigned long long)w * 64ull + (unsigned long long)(__ffsll((long long)bits) - 1);
            bits &= bits - 1;
        }
    }
}

__global__ __launch_bounds__(256) void k_compact_pairs(const uint32_t* __restrict__ starts,
                                                       const uint32_t* __restrict__ ends,
                                                       const uint64_t* __restrict__ pair_keep,
                                                       const uint32_t* __restrict__ word_base,
                                                       uint64_t n_pairs,
                                                       uint32_t* __restrict__ starts_c,
                                                       uint32_t* __restrict__ ends_c,
                                                       uint32_t* __restrict__ orig_pair) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n_pairs; q += stride) {
        const uint64_t word = pair_keep[q >> 6];
        const uint32_t bit = (uint32_t)(q & 63);
        if ((word >> bit) & 1ull) {
            const uint32_t dst = word_base[q >> 6] + (uint32_t)__popcll(word & ((1ull << bit) - 1ull));
            starts_c[2 * dst] = starts[2 * q];
            starts_c[2 * dst + 1] = starts[2 * q + 1];
            ends_c[2 * dst] = ends[2 * q];
            ends_c[2 * dst + 1] = ends[2 * q + 1];
            orig_pair[dst] = (uint32_t)q;
        }
    }
}

__global__ __launch_bounds__(256) void k_expand_mask(const uint64_t* __restrict__ mask_c,
                                                     const uint32_t* __restrict__ orig_pair,
                                                     uint32_t n_reads_c,
                                                     uint32_t* __restrict__ mask32) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_reads_c; i += stride) {
        if ((mask_c[i >> 6] >> (i & 63)) & 1ull) {
            const uint32_t orig = 2u * orig_pair[i >> 1] + (i & 1u);
            atomicOr(&mask32[orig >> 5], 1u << (orig & 31));
        }
    }
}
