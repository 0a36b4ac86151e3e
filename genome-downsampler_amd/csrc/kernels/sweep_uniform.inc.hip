// sweep_uniform.inc.hip -- part of qmcp_kernels.hip (one translation unit; included inside namespace qmcp).
// ------------------------------------------------------------------ uniform-span sweep
// All reads of the call have span `ell`.  With W(p) = #dropped reads with start <= p the
// canonical greedy (oracle/qmcp_oracle.c) is the pointwise-maximal W under
//     0 <= W(p) - W(p-1) <= c(p)          c(p)  = reads starting at p
//     W(p) - W(p-ell) <= ex(p)            ex(p) = cov(p) - min(cov(p), M)
// i.e. single-source shortest paths on a line graph with edges p-1 -> p (c(p)),
// p-ell -> p (ex(p)) and p -> p-1 (0).  Distances obey
//     d(p) = min( d(p-1) + c(p),  min_{j in [p-ell, p-1]} ( d(j) + ex(j+ell) ) )
// Positions are processed in blocks of `ell`.  The window minimum splits into
//     A(p) = suffix minimum over the previous block of h(j) = d(j) + ex(j+ell)
//            (held in registers: same lane, same slot as p), and
//     m(p) = running minimum of h over the current block before p.
// Carrying (d, m) turns one position into the min-plus map
//     d' = min(d + c, m, A)          m' = min(m, d' + ex) = min(d + c + ex, m, A + ex)
// and maps of the form  d' = min(d + a, m, u),  m' = min(d + b, m, v)  (b >= a, v >= u) are
// closed under composition:
//     a = min(a1 + a2, b1)   b = min(a1 + b2, b1)
//     u = min(u1 + a2, v1, u2)   v = min(u1 + b2, v1, v2)
// so a block is ONE wave-wide inclusive scan of 4-tuples (DPP row shifts + row broadcasts),
// a lane-local replay, and a suffix-min for the next block -- no iteration, no LDS.
// Selected count at p: S(p) = c(p) - (d(p) - d(p-1)); the kept reads of a start position are
// its S(p) lowest read indices (all ends are equal, so the rule's tie-break is the index).
//
// One wave per contig; local index i = lane * E + r, valid while i < ell.  Loads of block
// b+1 are issued before block b is computed (the chain is latency-bound).
struct Map4 { uint32_t a, b, u, v; };
__device__ __forceinline__ Map4 map_identity() { return Map4{0u, kInf, kInf, kInf}; }
__device__ __forceinline__ Map4 map_compose(const Map4& f, const Map4& g) {  // f first, then g
    Map4 r;
    r.a = min(f.a + g.a, f.b);
    r.b = min(f.a + g.b, f.b);
    r.u = min(min(f.u + g.a, f.v), g.u);
    r.v = min(min(f.u + g.b, f.v), g.v);
    return r;
}
__device__ __forceinline__ Map4 wave_incl_scan_map(Map4 x) {
#define QMCP_STEP(ctrl, rmask)                                  \
    {                                                           \
        Map4 p;                                                 \
        p.a = QMCP_DPP(0u, x.a, ctrl, rmask);                   \
        p.b = QMCP_DPP(kInf, x.b, ctrl, rmask);                 \
        p.u = QMCP_DPP(kInf, x.u, ctrl, rmask);                 \
        p.v = QMCP_DPP(kInf, x.v, ctrl, rmask);                 \
        x = map_compose(p, x);                                  \
    }
    QMCP_STEP(0x111, 0xF)
    QMCP_STEP(0x112, 0xF)
    QMCP_STEP(0x114, 0xF)
    QMCP_STEP(0x118, 0xF)
    QMCP_STEP(0x142, 0xA)
    QMCP_STEP(0x143, 0xC)
#undef QMCP_STEP
    return x;
}
// min over lanes strictly above this lane (>= kInf for lane 63).  Row totals are read
// with v_readlane and merged with per-lane masks (all-ones = "row does not count"), so there
// is no divergent control flow.
__device__ __forceinline__ uint32_t wave_excl_suffix_min(uint32_t t) {
    uint32_t s = t;
    s = min(s, QMCP_DPP(kInf, s, 0x101, 0xF));  // row_shl:1
    s = min(s, QMCP_DPP(kInf, s, 0x102, 0xF));
    s = min(s, QMCP_DPP(kInf, s, 0x104, 0xF));
    s = min(s, QMCP_DPP(kInf, s, 0x108, 0xF));
    const uint32_t r1 = __builtin_amdgcn_readlane(s, 16);
    const uint32_t r2 = __builtin_amdgcn_readlane(s, 32);
    const uint32_t r3 = __builtin_amdgcn_readlane(s, 48);
    const uint32_t row = (threadIdx.x & 63) >> 4;
    const uint32_t off1 = row < 1 ? 0u : 0xFFFFFFFFu;  // rows that lie above this lane's row
    const uint32_t off2 = row < 2 ? 0u : 0xFFFFFFFFu;
    const uint32_t off3 = row < 3 ? 0u : 0xFFFFFFFFu;
    s = min(min(s, r1 | off1), min(r2 | off2, r3 | off3));  // inclusive suffix min
    return QMCP_DPP(kInf, s, 0x130, 0xF);                   // wave_shl:1 -> exclusive
}

template <int E>
struct SweepLoads { uint32_t x0[E], x1[E], x2[E]; };

// Unconditional loads with clamped addresses (boff has ltot + 1 entries, base + L is always
// in range): every lane issues the same number of loads, so the compiler can keep the next
// block's loads in flight behind a counted s_waitcnt instead of draining them.
template <int E>
__device__ __forceinline__ void sweep_load(const uint32_t* __restrict__ cb /* boff + base */,
                                           uint32_t a, uint32_t ell, uint32_t L, uint32_t lane,
                                           SweepLoads<E>& o) {
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const uint32_t p = a + lane * E + r;
        o.x0[r] = cb[min(p, L)];
        o.x1[r] = cb[min(p + 1, L)];
        o.x2[r] = cb[min(p + ell + 1, L)];
    }
}

// DPP reads whose `old` operand is the operator's TRUE identity (INT32_MAX for signed min,
// 0xFFFFFFFF for unsigned min, 0 for add): LLVM's DPP combiner then folds the move into the
// operation (v_min_i32_dpp / v_min_u32_dpp / v_add_u32_dpp), one instruction per scan step.
#define QMCP_DPP_IMIN(v, ctrl, rmask) \
    __builtin_amdgcn_update_dpp((int)0x7FFFFFFF, (int)(v), (ctrl), (rmask), 0xF, false)
#define QMCP_DPP_UMIN(v, ctrl, rmask) \
    (uint32_t) __builtin_amdgcn_update_dpp((int)0xFFFFFFFF, (int)(v), (ctrl), (rmask), 0xF, false)

// One block of `ell` positions starting at contig position a.  State carried between blocks:
// h (the previous block's h(j) = d(j) + ex(j + ell), slot-aligned) and d_last.
//
// Fast form.  Without intra-block jumps the block recurrence is
//     d'(i) = min( d'(i-1) + c(i), A(i) ),   A(i) = min_{j >= i} h(j)   (previous block)
// Unrolling it and using that the inclusive count prefix C(i) is non-decreasing gives
//     d'(i) = min( d_last + C(i),  C(i) + min_{j <= i} (h(j) - C(j)),  min_{j > i} h(j) )
// i.e. two INDEPENDENT wave scans over the previous block's h (a prefix-min and a suffix-min,
// one value each) plus a prefix sum of counts that does not depend on the chain at all.
// Then the block's own running minimum m(i) = min_{j<i} h'(j) is compared with d'(i): if it
// never undercuts, d' also satisfies the full recurrence (with intra-block jumps) position by
// position and is exact.  The function returns whether some lane saw an undercut; the caller
// then redoes the group with sweep_block_full.  On deep data the binding jumps come from the
// previous block, so that is rare; either way the result is the same distances.
//
// A lone wave pays two wait states between dependent DPP operations, so the scans are written
// pairwise interleaved (prefix-min with suffix-min; the verification scan with the NEXT block's
// count prefix, which is why the per-block terms are prepared one block ahead).
template <int E>
struct BlockTerms { uint32_t cnt[E], exj[E]; };

template <int E>
struct BlockPrep {       // everything about a block that does not depend on the chain
    uint32_t x0[E];      // bucket offset of each slot (for the store)
    uint32_t cnt[E];     // reads starting at the slot
    uint32_t exj[E];     // ex at the landing position of the slot's jump (kInf: none)
    uint32_t C[E];       // inclusive prefix of cnt over the block
    uint32_t before;     // sum of cnt over all lower lanes
};

template <int E>
__device__ __forceinline__ void block_terms(const SweepLoads<E>& cur, uint32_t a, uint32_t ell,
                                            uint32_t L, uint32_t M, uint32_t lane, BlockTerms<E>& t) {
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const uint32_t i = lane * E + r;
        const uint32_t p = a + i;
        const bool valid = i < ell && p < L;
        const uint32_t cov = cur.x2[r] - cur.x1[r];
        t.cnt[r] = valid ? cur.x1[r] - cur.x0[r] : 0u;
        t.exj[r] = (valid && p + ell < L) ? (cov > M ? cov - M : 0u) : kInf;
    }
}

// The near-uniform route's sweep in stretches (round 4): the need at the landing position is moved by nadj and capped at
// what the regular reads can give -- need' = min(min(cov, M) + nadj, cov), ex' = cov - need' -- exactly what the
// event-driven form does with it (k_sweep_uniform_ev); adj[r] = nadj at the slot's landing position p + ell.
__device__ __forceinline__ uint32_t ex_adjusted(uint32_t cov, uint32_t M, int32_t adj) {
    return (uint32_t)((int32_t)cov - min((int32_t)min(cov, M) + adj, (int32_t)cov));
}
template <int E>
__device__ __forceinline__ void block_terms_adj(const SweepLoads<E>& cur, const int32_t (&adj)[E], uint32_t a, uint32_t ell,
                                                uint32_t L, uint32_t M, uint32_t lane, BlockTerms<E>& t) {
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const uint32_t i = lane * E + r;
        const uint32_t p = a + i;
        const bool valid = i < ell && p < L;
        const uint32_t cov = cur.x2[r] - cur.x1[r];
        t.cnt[r] = valid ? cur.x1[r] - cur.x0[r] : 0u;
        t.exj[r] = (valid && p + ell < L) ? ex_adjusted(cov, M, adj[r]) : kInf;
    }
}

// chain-independent part of a block, up to (not including) the wave scan of the lane sums
template <int E>
__device__ __forceinline__ uint32_t prep_local(const SweepLoads<E>& ld, uint32_t a, uint32_t ell,
                                               uint32_t L, uint32_t M, uint32_t lane,
                                               BlockPrep<E>& pr) {
    uint32_t lsum = 0;
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const uint32_t i = lane * E + r;
        const uint32_t p = a + i;
        const bool valid = i < ell && p < L;
        const uint32_t cov = ld.x2[r] - ld.x1[r];
        pr.x0[r] = ld.x0[r];
        pr.cnt[r] = valid ? ld.x1[r] - ld.x0[r] : 0u;
        pr.exj[r] = (valid && p + ell < L) ? (cov > M ? cov - M : 0u) : kInf;
        lsum += pr.cnt[r];
        pr.C[r] = lsum;
    }
    return lsum;
}
template <int E>
__device__ __forceinline__ void prep_finish(BlockPrep<E>& pr, uint32_t incl_lane_sums) {
    pr.before = QMCP_DPP(0u, incl_lane_sums, 0x138, 0xF);  // wave_shr:1, lane 0 gets 0
#pragma unroll
    for (int r = 0; r < E; ++r) pr.C[r] += pr.before;
}
template <int E>
__device__ __forceinline__ void prep_block(const SweepLoads<E>& ld, uint32_t a, uint32_t ell,
                                           uint32_t L, uint32_t M, uint32_t lane, BlockPrep<E>& pr) {
    prep_finish<E>(pr, wave_incl_scan_add(prep_local<E>(ld, a, ell, L, M, lane, pr)));
}

template <int E>
__device__ __forceinline__ void block_emit(const uint32_t (&x0)[E], const uint32_t (&cnt)[E],
                                           const uint32_t (&dn)[E], const uint32_t (&hn)[E],
                                           uint32_t d_in, uint32_t a, uint32_t trash, uint32_t ell,
                                           uint32_t Lrun, uint32_t lane, uint32_t last_lane,
                                           uint32_t last_r, uint32_t (&h)[E], uint32_t& d_last,
                                           uint32_t* __restrict__ csel) {
    uint32_t prev = d_in, pick = 0;
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const uint32_t i = lane * E + r;
        const uint32_t p = a + i;
        // unconditional store: slots outside the contig write the spare entry selend[ltot]
        csel[(i < ell && p < Lrun) ? p : trash] = x0[r] + (cnt[r] - (dn[r] - prev));
        prev = dn[r];
        if ((uint32_t)r == last_r) pick = dn[r];
        h[r] = hn[r];
    }
    d_last = __builtin_amdgcn_readlane(pick, last_lane);
}

// `pr` describes the block being solved; `nxt_ld` / `a_next` the block after it, whose terms
// are prepared here (into `nx`) in the shadow of this block's verification scan.
template <int E>
__device__ __forceinline__ bool sweep_block_fast(const BlockPrep<E>& pr, uint32_t a,
                                                 const SweepLoads<E>& nxt_ld, uint32_t a_next,
                                                 BlockPrep<E>& nx, uint32_t trash, uint32_t ell,
                                                 uint32_t L, uint32_t Lrun, uint32_t M, uint32_t lane,
                                                 uint32_t last_lane, uint32_t last_r,
                                                 uint32_t (&h)[E], uint32_t& d_last,
                                                 uint32_t* __restrict__ csel) {
    uint32_t dn[E], hn[E];
    // prefix-min of (h - C) and exclusive suffix-min of h over the previous block.
    // h < 2^31 and (h - C) >= -2^28, so the signed arithmetic cannot overflow.
    int32_t lp[E];
    int32_t pm = 0x7FFFFFFF;
    uint32_t sx[E];
    uint32_t sm = 0xFFFFFFFFu;
#pragma unroll
    for (int r = 0; r < E; ++r) {
        pm = min(pm, (int32_t)h[r] - (int32_t)pr.C[r]);
        lp[r] = pm;
    }
#pragma unroll
    for (int r = E - 1; r >= 0; --r) {
        sx[r] = sm;  // min over slots r' > r of this lane
        sm = min(sm, h[r]);
    }
    const uint32_t srun = sm;
    // interleaved: inclusive prefix-min of pm (row_shr ...) and inclusive suffix-min of sm (row_shl ...)
    pm = min(pm, QMCP_DPP_IMIN(pm, 0x111, 0xF));  sm = min(sm, QMCP_DPP_UMIN(sm, 0x101, 0xF));
    pm = min(pm, QMCP_DPP_IMIN(pm, 0x112, 0xF));  sm = min(sm, QMCP_DPP_UMIN(sm, 0x102, 0xF));
    pm = min(pm, QMCP_DPP_IMIN(pm, 0x114, 0xF));  sm = min(sm, QMCP_DPP_UMIN(sm, 0x104, 0xF));
    pm = min(pm, QMCP_DPP_IMIN(pm, 0x118, 0xF));  sm = min(sm, QMCP_DPP_UMIN(sm, 0x108, 0xF));
    const uint32_t r1 = __builtin_amdgcn_readlane(sm, 16);
    pm = min(pm, QMCP_DPP_IMIN(pm, 0x142, 0xA));
    const uint32_t r2 = __builtin_amdgcn_readlane(sm, 32);
    const uint32_t r3 = __builtin_amdgcn_readlane(sm, 48);
    pm = min(pm, QMCP_DPP_IMIN(pm, 0x143, 0xC));
    {
        const uint32_t row = lane >> 4;
        const uint32_t off1 = row < 1 ? 0u : 0xFFFFFFFFu;  // rows that lie above this lane's row
        const uint32_t off2 = row < 2 ? 0u : 0xFFFFFFFFu;
        const uint32_t off3 = row < 3 ? 0u : 0xFFFFFFFFu;
        sm = min(min(sm, r1 | off1), min(r2 | off2, r3 | off3));  // inclusive suffix min
    }
    const int32_t pp = __builtin_amdgcn_update_dpp((int)0x7FFFFFFF, (int)pm, 0x138, 0xF, 0xF, false);
    const uint32_t after = QMCP_DPP(0xFFFFFFFFu, sm, 0x130, 0xF);  // lanes above (all-ones: none)

    // d entering the lane = d' at the last slot of the lane below (C = before, prefix-min = pp,
    // suffix = everything from this lane's first slot on); for lane 0 it is d_last.
    uint32_t d_in = min(min(d_last + pr.before, (uint32_t)(pp + (int32_t)pr.before)), min(srun, after));
    d_in = lane == 0 ? d_last : d_in;
    uint32_t vm = 0xFFFFFFFFu;
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const uint32_t viaP = (uint32_t)((int32_t)pr.C[r] + min(pp, lp[r]));
        dn[r] = min(min(d_last + pr.C[r], viaP), min(sx[r], after));
        hn[r] = dn[r] + pr.exj[r];
        vm = min(vm, hn[r]);
    }
    // interleaved: verification scan (inclusive prefix-min of the lanes' min h') and the next
    // block's count prefix (inclusive prefix-sum of its lane sums)
    uint32_t cs = prep_local<E>(nxt_ld, a_next, ell, L, M, lane, nx);
    vm = min(vm, QMCP_DPP_UMIN(vm, 0x111, 0xF));  cs += QMCP_DPP(0u, cs, 0x111, 0xF);
    vm = min(vm, QMCP_DPP_UMIN(vm, 0x112, 0xF));  cs += QMCP_DPP(0u, cs, 0x112, 0xF);
    vm = min(vm, QMCP_DPP_UMIN(vm, 0x114, 0xF));  cs += QMCP_DPP(0u, cs, 0x114, 0xF);
    vm = min(vm, QMCP_DPP_UMIN(vm, 0x118, 0xF));  cs += QMCP_DPP(0u, cs, 0x118, 0xF);
    vm = min(vm, QMCP_DPP_UMIN(vm, 0x142, 0xA));  cs += QMCP_DPP(0u, cs, 0x142, 0xA);
    vm = min(vm, QMCP_DPP_UMIN(vm, 0x143, 0xC));  cs += QMCP_DPP(0u, cs, 0x143, 0xC);
    prep_finish<E>(nx, cs);
    // m entering this lane = min of h' over all lower lanes
    uint32_t run = QMCP_DPP(0xFFFFFFFFu, vm, 0x138, 0xF);
    bool undercut = false;
#pragma unroll
    for (int r = 0; r < E; ++r) {
        undercut |= run < dn[r];
        run = min(run, hn[r]);
    }
    block_emit<E>(pr.x0, pr.cnt, dn, hn, d_in, a, trash, ell, Lrun, lane, last_lane, last_r, h, d_last, csel);
    return __any(undercut);
}

// General form of the same block: maps carrying (d, m) -- see Map4 above.
template <int E>
__device__ __forceinline__ void sweep_block_full(const SweepLoads<E>& cur, uint32_t a,
                                                 uint32_t trash, uint32_t ell, uint32_t L, uint32_t Lrun,
                                                 uint32_t M, uint32_t lane, uint32_t last_lane,
                                                 uint32_t last_r, uint32_t (&h)[E],
                                                 uint32_t& d_last, uint32_t* __restrict__ csel,
                                                 const int32_t* __restrict__ nb = nullptr /* nadj + base, or null */) {
    BlockTerms<E> t;
    if (nb != nullptr) {
        int32_t adj[E];
#pragma unroll
        for (int r = 0; r < E; ++r) adj[r] = nb[min(a + lane * E + r + ell, L)];
        block_terms_adj<E>(cur, adj, a, ell, L, M, lane, t);
    } else {
        block_terms<E>(cur, a, ell, L, M, lane, t);
    }
    uint32_t sufA[E], dn[E], hn[E];
    {
        uint32_t srun = kInf;
#pragma unroll
        for (int r = E - 1; r >= 0; --r) { srun = min(srun, h[r]); sufA[r] = srun; }
        const uint32_t after = min(wave_excl_suffix_min(srun), kInf);
#pragma unroll
        for (int r = 0; r < E; ++r) sufA[r] = min(sufA[r], after);  // min_{j >= i} h(j)
    }
    Map4 acc = map_identity();
#pragma unroll
    for (int r = 0; r < E; ++r) {
        Map4 e;
        e.a = t.cnt[r];
        e.b = t.cnt[r] + t.exj[r];
        e.u = sufA[r];
        e.v = sufA[r] + t.exj[r];
        acc = map_compose(acc, e);
    }
    Map4 inc = wave_incl_scan_map(acc);
    Map4 pre;  // composition of all lower lanes (identity for lane 0)
    pre.a = QMCP_DPP(0u, inc.a, 0x138, 0xF);
    pre.b = QMCP_DPP(kInf, inc.b, 0x138, 0xF);
    pre.u = QMCP_DPP(kInf, inc.u, 0x138, 0xF);
    pre.v = QMCP_DPP(kInf, inc.v, 0x138, 0xF);
    // state entering this lane: (d, m) = pre applied to (d_last, +inf)
    const uint32_t d_in = min(d_last + pre.a, pre.u);
    uint32_t dd = d_in;
    uint32_t m = min(d_last + pre.b, pre.v);
#pragma unroll
    for (int r = 0; r < E; ++r) {
        dd = min(min(dd + t.cnt[r], m), sufA[r]);
        dn[r] = dd;
        hn[r] = dd + t.exj[r];
        m = min(m, hn[r]);
    }
    block_emit<E>(cur.x0, t.cnt, dn, hn, d_in, a, trash, ell, Lrun, lane, last_lane, last_r, h, d_last, csel);
}

// blocks [b_begin, b_end) in the general form, loads of block b+1 in flight under block b
template <int E>
__device__ __forceinline__ void sweep_full_run(const uint32_t* __restrict__ cb, uint32_t b_begin,
                                               uint32_t b_end, uint32_t trash, uint32_t ell,
                                               uint32_t L, uint32_t Lrun, uint32_t M, uint32_t lane,
                                               uint32_t last_lane, uint32_t last_r,
                                               uint32_t (&h)[E], uint32_t& d_last,
                                               uint32_t* __restrict__ csel,
                                               const int32_t* __restrict__ nb = nullptr /* nadj + base, or null */) {
    SweepLoads<E> T0, T1;
    sweep_load<E>(cb, b_begin * ell, ell, L, lane, T0);
    for (uint32_t b = b_begin; b < b_end; b += 2) {
        sweep_load<E>(cb, (b + 1) * ell, ell, L, lane, T1);
        sweep_block_full<E>(T0, b * ell, trash, ell, L, Lrun, M, lane, last_lane, last_r, h, d_last, csel, nb);
        sweep_load<E>(cb, (b + 2) * ell, ell, L, lane, T0);
        if (b + 1 < b_end)
            sweep_block_full<E>(T1, (b + 1) * ell, trash, ell, L, Lrun, M, lane, last_lane, last_r, h, d_last, csel, nb);
    }
}

// What one sweep workgroup works on: a whole contig, or -- when the host found cut points
// (positions whose coverage is <= M: every read covering them is kept, so the sweep's state behind
// them does not depend on what came before) -- the stretch of a contig between two cut points.
//   base   global position where the stretch starts
//   Lrun   its length: blocks are solved, and results stored, only up to here
//   L      distance to the CONTIG's end: loads and jump landings are valid up to here (the last block
//          of a stretch looks past its end)
// `seg` (null: one workgroup per contig) = [count, then {start, end, contig end} of each stretch].
struct SweepSeg { uint32_t base, Lrun, L; };
__device__ __forceinline__ bool sweep_segment(const uint64_t* __restrict__ contig_pos_off,
                                              const uint32_t* __restrict__ seg, uint32_t idx, SweepSeg& s) {
    if (seg == nullptr) {
        s.base = (uint32_t)contig_pos_off[idx];
        s.L = (uint32_t)(contig_pos_off[idx + 1] - contig_pos_off[idx]);
        s.Lrun = s.L;
        return s.L != 0;
    }
    if (idx >= seg[0]) return false;  // the grid is an upper bound
    const uint32_t* q = seg + 1 + 3 * idx;
    s.base = q[0];
    s.Lrun = q[1] - q[0];
    s.L = q[2] - q[0];
    return s.Lrun != 0;
}
// ex = cov - min(cov, M) at the first block's positions: the h of the (virtual) block before it,
// whose distances are all equal.  The window of starts may reach back past the stretch, and even past
// the contig's start: no read starts within ell - 1 of a contig's end (it would cross it), so the
// global prefix counts need no clamp at contig borders.
template <int E>
__device__ __forceinline__ void sweep_initial_h(const uint32_t* __restrict__ boff, const SweepSeg& sg,
                                                uint32_t ell, uint32_t M, uint32_t lane, uint32_t (&h)[E],
                                                const int32_t* __restrict__ nadj = nullptr /* near-uniform route, or null */) {
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const uint32_t i = lane * E + r;
        const uint32_t hi = sg.base + min(i + 1, sg.L);  // prefix index p + 1
        const uint32_t cov = boff[hi] - boff[hi >= ell ? hi - ell : 0u];
        uint32_t ex = cov > M ? cov - M : 0u;
        if (nadj != nullptr) ex = ex_adjusted(cov, M, nadj[sg.base + min(i, sg.L)]);
        h[r] = (i < ell && i < sg.L) ? ex : kInf;
    }
}

template <int E>
__global__ __launch_bounds__(64) void k_sweep_uniform(const uint32_t* __restrict__ boff,
                                                      const uint64_t* __restrict__ contig_pos_off,
                                                      uint32_t ell, uint32_t M, uint32_t ltot,
                                                      uint32_t* __restrict__ selend,
                                                      uint32_t* __restrict__ iter_stats,
                                                      const uint32_t* __restrict__ seg) {
    const uint32_t lane = threadIdx.x;
    const uint32_t c_id = blockIdx.x;
    SweepSeg sg;
    if (!sweep_segment(contig_pos_off, seg, c_id, sg)) return;
    const uint32_t base = sg.base, L = sg.L, Lrun = sg.Lrun;
    const uint32_t n_blocks = (Lrun + ell - 1) / ell;
    // this wave is a serial dependency chain that may share its SIMD with streaming kernels:
    // win the issue arbitration
    __builtin_amdgcn_s_setprio(3);

    uint32_t h[E];  // previous block's h(j) = d(j) + ex(j + ell), aligned with this block's slots
    // virtual block -1: d == 0 and the jump from j = i - ell lands on p = i
    sweep_initial_h<E>(boff, sg, ell, M, lane, h);
    uint32_t d_last = 0;
    uint32_t n_full = 0;  // blocks that needed the 4-component form
    const uint32_t last_lane = (ell - 1) / E, last_r = (ell - 1) % E;

    // Four register sets in rotation: the loads of block b+3 are issued before block b is
    // computed, so three blocks of work (~1.5 us) cover the load latency even when the radix
    // passes on the other stream keep HBM busy.  No register copies between iterations; loads
    // past the contig end clamp to a valid address.
    const uint32_t* __restrict__ cb = boff + base;
    uint32_t* __restrict__ csel = selend + base;
    const uint32_t trash = ltot - base;  // csel[trash] == selend[ltot], the spare entry
    // Steady state: groups of four blocks in the fast form, four register sets in rotation
    // (loads of block b+3 are issued before block b is computed), no branch around any load
    // or store so the compiler keeps counted waits.  A group in which some block reports an
    // undercut leaves the loop, is redone from the saved state in the general form, and the
    // pipeline restarts behind it.
    const uint32_t n_groups = n_blocks / 4;
    uint32_t g = 0;
    uint32_t penalty = 0;  // groups to run in the general form after a failed fast attempt
    while (g < n_groups) {
        if (penalty > 0) {
            // sparse / low-coverage stretch: the fast form keeps failing here, do not try it
            const uint32_t run = min(penalty, n_groups - g);
            sweep_full_run<E>(cb, g * 4, (g + run) * 4, trash, ell, L, Lrun, M, lane, last_lane, last_r, h,
                              d_last, csel);
            n_full += run * 4;
            g += run;
            if (g >= n_groups) break;
        }
        // pipeline start: loads for the group's first four blocks, terms of its first block
        SweepLoads<E> S0, S1, S2, S3;
        BlockPrep<E> PA, PB;
        sweep_load<E>(cb, g * 4 * ell, ell, L, lane, S0);
        sweep_load<E>(cb, (g * 4 + 1) * ell, ell, L, lane, S1);
        sweep_load<E>(cb, (g * 4 + 2) * ell, ell, L, lane, S2);
        sweep_load<E>(cb, (g * 4 + 3) * ell, ell, L, lane, S3);
        prep_block<E>(S0, g * 4 * ell, ell, L, M, lane, PA);
        uint32_t h_save[E];
        uint32_t d_save = d_last;
        bool bad = false;
        uint32_t good = 0;
        for (; g < n_groups; ++g) {
            const uint32_t a = g * 4 * ell;
#pragma unroll
            for (int r = 0; r < E; ++r) h_save[r] = h[r];
            d_save = d_last;
            // block k solves with terms prepared during block k-1 and prepares block k+1 from
            // loads issued three blocks earlier; S_k is re-loaded for block k+4 once consumed
#define QMCP_FAST(PR, pos, LD_NEXT, PR_NEXT)                                                        \
    sweep_block_fast<E>(PR, pos, LD_NEXT, (pos) + ell, PR_NEXT, trash, ell, L, Lrun, M, lane, last_lane,  \
                        last_r, h, d_last, csel)
            sweep_load<E>(cb, a + 4 * ell, ell, L, lane, S0);
            bad = QMCP_FAST(PA, a, S1, PB);
            sweep_load<E>(cb, a + 5 * ell, ell, L, lane, S1);
            bad |= QMCP_FAST(PB, a + ell, S2, PA);
            sweep_load<E>(cb, a + 6 * ell, ell, L, lane, S2);
            bad |= QMCP_FAST(PA, a + 2 * ell, S3, PB);
            sweep_load<E>(cb, a + 7 * ell, ell, L, lane, S3);
            bad |= QMCP_FAST(PB, a + 3 * ell, S0, PA);
#undef QMCP_FAST
            if (bad) break;
            ++good;
        }
        if (bad) {
            // redo group g in the general form (its stores overwrite the speculative ones);
            // back off from the fast form: 1, 3, 7, ... 63 groups, reset by a fast success
#pragma unroll
            for (int r = 0; r < E; ++r) h[r] = h_save[r];
            d_last = d_save;
            sweep_full_run<E>(cb, g * 4, g * 4 + 4, trash, ell, L, Lrun, M, lane, last_lane, last_r, h, d_last, csel);
            n_full += 4;
            ++g;
            penalty = good > 0 ? 1u : min(2 * penalty + 1, 63u);
        } else {
            penalty = 0;
        }
    }
    // tail: at most three blocks, general form
    if (n_groups * 4 < n_blocks)
        sweep_full_run<E>(cb, n_groups * 4, n_blocks, trash, ell, L, Lrun, M, lane, last_lane, last_r, h, d_last, csel);
    if (iter_stats && lane == 0) {
        atomicAdd(&iter_stats[0], n_full);
        atomicAdd(&iter_stats[1], n_blocks);
        atomicAdd(&iter_stats[2], 1u);  // stretches swept
    }
}
