// sweep_uniform_pipelines.inc.hip -- part of qmcp_kernels.hip (one translation unit; included inside namespace qmcp).
// ------------------------------------------------------------------ uniform sweep: seven-wave pipelines
// Same algorithm as k_sweep_uniform, with the work of a group of four blocks split over three
// waves of one workgroup (one workgroup per contig) that advance in lockstep, one
// __syncthreads() per group, all hand-offs through LDS (no spinning, uniform control flow):
//   wave 0  PREP    loads bucket offsets and prepares the chain-independent terms of group t
//   wave 1  CHAIN   solves group t-1 in the fast form: only the two min-scans and the combine
//                   remain on the serial path
//   wave 2  CHECK   verifies group t-2 (running-minimum test) and stores its results
// If CHECK finds an undercut in group f, every wave sees the flag after the barrier; the chain
// wave restores its state from the start of group f, redoes f (and, while the fast form keeps
// failing, a growing run of following groups) alone in the general form, and the pipeline
// restarts behind it.  Nothing of a failed group is stored by CHECK (it verifies all four blocks
// before storing), and the chain wave is at most one group ahead of it, so no speculative value
// ever reaches memory.
// LDS: three group slots x four blocks x (6E + 2) words x 64 lanes, word-major (conflict-free).
// A "row" is one block's bucket offsets in slot layout: X[r] = cb[min(a + lane*E + r, L)].
// Each lane's E slots are adjacent in memory, so a row is ONE vector load per lane (dwordx2/x3/
// x4) instead of E scalar ones; rows that poke past the end of the table fall back to clamped
// scalar loads.  The three views a block needs are then built in registers:
//   x0 = X(b),  x1 = X(b) shifted by one slot,  x2 = X(b+1) shifted by one slot,
// where "shifted" takes slot r+1 of the same lane, the next lane's slot 0 (DPP wave_shl:1) for
// the lane's last slot, and the following row's very first entry for the block's last slot.
template <int E> struct RowVec;
template <> struct RowVec<1> { typedef uint32_t type; };
template <> struct RowVec<2> { typedef uint32_t type __attribute__((ext_vector_type(2), aligned(4))); };
template <> struct RowVec<3> { typedef uint32_t type __attribute__((ext_vector_type(3), aligned(4))); };
template <> struct RowVec<4> { typedef uint32_t type __attribute__((ext_vector_type(4), aligned(4))); };

// Split in two so that the select fix-up (which needs the data) can sit a whole stage after the
// load was issued: row_issue starts the load, row_finish turns it into slot values.
template <int E>
struct RowRaw { typename RowVec<E>::type v; uint32_t sh; };

template <int E>
__device__ __forceinline__ void row_issue(const uint32_t* __restrict__ cb, uint32_t a, uint32_t L,
                                          uint32_t lane, RowRaw<E>& raw) {
    // branch-free (a load inside a divergent branch makes the compiler drain vmcnt at the join):
    // the vector is read at a base clamped so that it ends at cb[L] at the latest
    const uint32_t p = a + lane * E;
    const uint32_t pe = min(p, L + 1 - E);  // L >= ell > E always
    raw.sh = p - pe;
    raw.v = *reinterpret_cast<const typename RowVec<E>::type*>(cb + pe);
}
template <int E>
__device__ __forceinline__ void row_finish(const RowRaw<E>& raw, uint32_t (&X)[E]) {
    // lanes whose base moved pick their entries with selects: X[r] = v[min(sh + r, E-1)]
    if constexpr (E == 1) {
        X[0] = raw.v;
    } else {
#pragma unroll
        for (int r = 0; r < E; ++r) {
            uint32_t x = raw.v[E - 1];
#pragma unroll
            for (int q = E - 2; q >= r; --q) x = (raw.sh + r <= (uint32_t)q) ? raw.v[q] : x;
            X[r] = x;
        }
    }
}
template <int E>
__device__ __forceinline__ void row_load(const uint32_t* __restrict__ cb, uint32_t a, uint32_t L,
                                         uint32_t lane, uint32_t (&X)[E]) {
    RowRaw<E> raw;
    row_issue<E>(cb, a, L, lane, raw);
    row_finish<E>(raw, X);
}

template <int E>
__device__ __forceinline__ void rows_to_loads(const uint32_t (&Xa)[E], const uint32_t (&Xb)[E],
                                              const uint32_t (&Xc)[E], uint32_t lane,
                                              uint32_t last_lane, uint32_t last_r, SweepLoads<E>& o) {
    const uint32_t nb0 = QMCP_DPP(0u, Xa[0], 0x130, 0xF);  // next lane's first slot (wave_shl:1)
    const uint32_t nb1 = QMCP_DPP(0u, Xb[0], 0x130, 0xF);
    const uint32_t tail1 = __builtin_amdgcn_readlane(Xb[0], 0);  // cb[a + ell]
    const uint32_t tail2 = __builtin_amdgcn_readlane(Xc[0], 0);  // cb[a + 2 ell]
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const bool is_last = lane == last_lane && (uint32_t)r == last_r;
        o.x0[r] = Xa[r];
        o.x1[r] = is_last ? tail1 : (r + 1 < E ? Xa[r + 1 < E ? r + 1 : r] : nb0);
        o.x2[r] = is_last ? tail2 : (r + 1 < E ? Xb[r + 1 < E ? r + 1 : r] : nb1);
    }
}

template <int E>
struct MwLayout {
    static constexpr int kG = 8;            // blocks per group (one pipeline stage)
    static constexpr int kC = 0;            // [E]  inclusive count prefix       PREP -> CHAIN
    static constexpr int kEx = E;           // [E]  ex at the jump landing       PREP -> CHAIN, CHECK
    static constexpr int kX0 = 2 * E;       // [E]  bucket offsets               PREP -> CHECK
    static constexpr int kCnt = 3 * E;      // [E]  counts                       PREP -> CHECK
    static constexpr int kDn = 4 * E;       // [E]  distances                    CHAIN -> CHECK
    static constexpr int kH0 = 5 * E;       // [E]  block 0 only: h entering the group (rollback state)
    static constexpr int kDin = 6 * E;      // [1]  block 0 only: d entering the group   CHAIN -> CHECK
    static constexpr int kWords = 6 * E + 1;
    static constexpr int kSlots = 3;
    static constexpr size_t kBytes = (size_t)kSlots * kG * kWords * 64 * sizeof(uint32_t) + 64;
};

// Pipelined form of the sweep: one workgroup of seven waves per contig.
//   waves 0,1,2,4  PREP   two blocks of the group each: bucket-offset rows -> counts, prefix, ex
//   wave  3        CHAIN  the serial part: two min-scans + combine per block (alone on its SIMD:
//                         waves are placed round-robin on the four SIMDs)
//   waves 5,6      CHECK  four blocks of the group each: undercut check, selected counts, stores
// A lone wave issues an instruction every 5-8 cycles (lab/issue_lab.hip), so everything that does
// not depend on the chain is kept off the chain wave.  Blocks go in groups of kG; stage t has PREP
// on group g0+t, CHAIN on g0+t-1, CHECK on g0+t-2, one barrier per stage, all hand-offs through
// three LDS slots.  Results reach memory only after the verdict: every block before the first
// failed one of a group is exact and is stored.  On a failed check every wave but CHAIN leaves the
// pipeline right after the barrier; CHAIN, which reads the flag without waiting for it, notices
// at the end of the stage it has already started, rebuilds the state entering the failed block
// from LDS (it publishes h and d at every group entry; inside a group the state is the previous
// block's distances + ex), redoes the rest of that group in the general form, and the pipeline
// restarts behind it -- warm: what PREP made for the next two groups is still in LDS.  Only when
// the first group of a run fails again does CHAIN back off to runs of 1, 3, 7 ... 63 groups in the
// general form (sparse data, where the fast form rarely holds).
template <int E>
__global__ __launch_bounds__(448) void k_sweep_uniform_mw(const uint32_t* __restrict__ boff,
                                                          const uint64_t* __restrict__ contig_pos_off,
                                                          uint32_t ell, uint32_t M, uint32_t ltot,
                                                          uint32_t* __restrict__ selend,
                                                          uint32_t* __restrict__ iter_stats,
                                                          const uint32_t* __restrict__ seg) {
    using Ly = MwLayout<E>;
    constexpr int kG = Ly::kG;
    extern __shared__ uint32_t s_mw[];
    uint32_t* s_flag = s_mw + (size_t)Ly::kSlots * kG * Ly::kWords * 64;
    const uint32_t lane = threadIdx.x & 63;
    // wave-uniform in the compiler's eyes too: role branches are scalar branches, so the PREP
    // waves can run their own copy of the stage loop
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t role = wv == 3 ? 1u : (wv >= 5 ? 2u : 0u);  // 0 PREP, 1 CHAIN, 2 CHECK
    const uint32_t pblk = 2 * (wv == 4 ? 3u : wv);            // PREP: first of its two blocks
    const uint32_t cblk = 4 * (wv - 5);                       // CHECK: first of its four blocks
    const uint32_t c_id = blockIdx.x;
    SweepSeg sg;
    if (!sweep_segment(contig_pos_off, seg, c_id, sg)) return;  // uniform over the workgroup
    const uint32_t base = sg.base, L = sg.L, Lrun = sg.Lrun;
    const uint32_t n_blocks = (Lrun + ell - 1) / ell;
    const uint32_t n_groups = n_blocks / kG;
    const uint32_t* __restrict__ cb = boff + base;
    uint32_t* __restrict__ csel = selend + base;
    const uint32_t trash = ltot - base;  // E spare words behind the table absorb idle lanes' stores
    const uint32_t last_lane = (ell - 1) / E, last_r = (ell - 1) % E;
    // the waves form one serial pipeline: each must win issue arbitration against the
    // streaming kernels that may share their SIMDs
    __builtin_amdgcn_s_setprio(3);
    // s_flag[0]: first block of the checked group whose check failed (kNoFail: none)
    constexpr uint32_t kNoFail = 0xFFFFFFFFu;
    if (threadIdx.x == 0) s_flag[0] = kNoFail;

    // chain state (meaningful in the CHAIN wave only)
    uint32_t h[E];
    sweep_initial_h<E>(boff, sg, ell, M, lane, h);
    uint32_t d_last = 0;
    uint32_t n_full = 0;
    uint32_t penalty = 0;
    uint32_t g0 = 0;
    bool warm = false;  // the PREP data of groups g0 and g0+1 survive from the failed run
#ifdef QMCP_MW_STAMP
    unsigned long long stamp_work = 0, stamp_wait = 0, stamp_fail = 0, stamp_fails = 0, stamp_post = 0, stamp_iters = 0;
    unsigned long long stamp_prev = 0;
    const unsigned long long stamp_begin = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();

    // MW_AT: word w of block k of a slot; `slot` is a per-stage base pointer, so k and w fold into
    // the instruction's immediate offset
#define MW_SLOT0(idx) (s_mw + (size_t)(idx) * kG * Ly::kWords * 64)
#define MW_SLOT(idx) (MW_SLOT0(idx) + lane)
#define MW_AT(slot, k, w) (slot)[((k) * Ly::kWords + (w)) * 64]

    while (g0 < n_groups) {
        if (penalty > 0) {
            warm = false;
            const uint32_t run = min(penalty, n_groups - g0);
            if (role == 1) {
                sweep_full_run<E>(cb, g0 * kG, (g0 + run) * kG, trash, ell, L, Lrun, M, lane, last_lane, last_r,
                                  h, d_last, csel);
                n_full += run * kG;
            }
            g0 += run;
            if (g0 >= n_groups) break;
        }
        const uint32_t n_left = n_groups - g0;
        uint32_t failed = 0xFFFFFFFFu;  // group whose check failed
        uint32_t sel[4][E];   // CHECK: results of the group checked in this stage, stored after the verdict
        uint32_t flag_seen = kNoFail;  // CHAIN: the flag as read after the previous stage's barrier
        // A warm run starts one stage in: the chain can take group g0 at once, and PREP resumes
        // with group g0+2.
        const uint32_t prep_from = warm ? 2u : 0u;
        const uint32_t t_begin = warm ? 1u : 0u;
#ifdef QMCP_MW_STAMP
#define MW_STAGE_BEGIN()                                                  \
    const unsigned long long stamp0 = __builtin_amdgcn_s_memtime();       \
    if (stamp_prev != 0) stamp_post += stamp0 - stamp_prev;               \
    stamp_iters += 1;
#define MW_STAGE_BARRIER()                                                \
    const unsigned long long stamp1 = __builtin_amdgcn_s_memtime();       \
    __syncthreads();                                                      \
    const unsigned long long stamp2 = __builtin_amdgcn_s_memtime();       \
    stamp_work += stamp1 - stamp0;                                        \
    stamp_wait += stamp2 - stamp1;                                        \
    stamp_prev = stamp2;
#else
#define MW_STAGE_BEGIN()
#define MW_STAGE_BARRIER() __syncthreads();
#endif
        if (role == 0) {
            // PREP waves run their own copy of the stage loop (same barriers, same exits), unrolled
            // kD times: a wave needs four rows for its two blocks (blocks kG*g+pblk .. +3), and loads
            // them kD stages ahead into kD register sets that take turns in a FIXED order in the
            // instruction stream -- so the compiler waits for exactly the oldest set (a counted
            // s_waitcnt) and a row has kD whole stages to land.  Memory latency beside a
            // bandwidth-bound kernel is several microseconds; one stage is about one.
            constexpr uint32_t kD = 3;
            RowRaw<E> R0[4], R1[4], R2[4];
            auto issue_rows = [&](RowRaw<E> (&buf)[4], uint32_t g) {
#pragma unroll
                for (int k = 0; k < 4; ++k) row_issue<E>(cb, (g * kG + pblk + k) * ell, L, lane, buf[k]);
            };
            issue_rows(R0, g0 + t_begin);
            issue_rows(R1, g0 + t_begin + 1);
            issue_rows(R2, g0 + t_begin + 2);
            // one stage with register set `buf` (which holds the rows of group g0+t); false: run over
            auto pstage = [&](RowRaw<E> (&buf)[4], uint32_t t) -> bool {
                MW_STAGE_BEGIN()
                const uint32_t g = g0 + t;
                if (t >= prep_from && t < n_left) {
                    uint32_t* const slot = MW_SLOT(g % Ly::kSlots) + pblk * Ly::kWords * 64;
                    uint32_t Wr[4][E];
#pragma unroll
                    for (int k = 0; k < 4; ++k) row_finish<E>(buf[k], Wr[k]);
                    __builtin_amdgcn_sched_barrier(0);  // rows consumed before the set is reloaded
                    issue_rows(buf, g + kD);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
                        SweepLoads<E> ldk;
                        rows_to_loads<E>(Wr[kk], Wr[kk + 1], Wr[kk + 2], lane, last_lane, last_r, ldk);
                        BlockPrep<E> pr;
                        prep_block<E>(ldk, (g * kG + pblk + kk) * ell, ell, L, M, lane, pr);
#pragma unroll
                        for (int r = 0; r < E; ++r) {
                            MW_AT(slot, kk, Ly::kC + r) = pr.C[r];
                            MW_AT(slot, kk, Ly::kEx + r) = pr.exj[r];
                            MW_AT(slot, kk, Ly::kX0 + r) = pr.x0[r];
                            MW_AT(slot, kk, Ly::kCnt + r) = pr.cnt[r];
                        }
                    }
                } else if (t < n_left) {
                    issue_rows(buf, g + kD);  // a warm run's first stage: the group exists already, the set moves on
                }
                MW_STAGE_BARRIER()
                const uint32_t bad = t >= 2 ? s_flag[0] : kNoFail;
                if (bad != kNoFail) { failed = g0 + t - 2; return false; }
                return true;
            };
            for (uint32_t t = t_begin;; t += kD) {
                if (t >= n_left + 2 || !pstage(R0, t)) break;
                if (t + 1 >= n_left + 2 || !pstage(R1, t + 1)) break;
                if (t + 2 >= n_left + 2 || !pstage(R2, t + 2)) break;
            }
        } else
        for (uint32_t t = t_begin; t < n_left + 2; ++t) {
            MW_STAGE_BEGIN()
            if (role == 1) {
                if (t >= 1 && t <= n_left) {
                    const uint32_t g = g0 + t - 1;
                    uint32_t* const slot = MW_SLOT(g % Ly::kSlots);
                    const uint32_t* const slot0 = MW_SLOT0(g % Ly::kSlots);
                    // state entering the group: d for the CHECK waves (inside a group they read the
                    // previous block's last distance themselves), h and d for a rollback
                    MW_AT(slot, 0, Ly::kDin) = d_last;
#pragma unroll
                    for (int r = 0; r < E; ++r) MW_AT(slot, 0, Ly::kH0 + r) = h[r];
                    // terms of block k+1 are read from LDS before block k's scans start, so their
                    // latency hides under the scans (the compiler will not hoist LDS reads above the
                    // previous block's LDS writes by itself)
                    uint32_t Cn[E], exn[E];
#pragma unroll
                    for (int r = 0; r < E; ++r) {
                        Cn[r] = MW_AT(slot, 0, Ly::kC + r);
                        exn[r] = MW_AT(slot, 0, Ly::kEx + r);
                    }
#pragma unroll
                    for (int k = 0; k < kG; ++k) {
                        uint32_t C[E], exj[E];
#pragma unroll
                        for (int r = 0; r < E; ++r) { C[r] = Cn[r]; exj[r] = exn[r]; }
                        if (k + 1 < kG) {
#pragma unroll
                            for (int r = 0; r < E; ++r) {
                                Cn[r] = MW_AT(slot, k + 1, Ly::kC + r);
                                exn[r] = MW_AT(slot, k + 1, Ly::kEx + r);
                            }
                        }
                        // the two chain scans, interleaved (see sweep_block_fast)
                        int32_t lp[E];
                        int32_t pm = 0x7FFFFFFF;
                        uint32_t sx[E];
                        uint32_t sm = 0xFFFFFFFFu;
#pragma unroll
                        for (int r = 0; r < E; ++r) { pm = min(pm, (int32_t)h[r] - (int32_t)C[r]); lp[r] = pm; }
#pragma unroll
                        for (int r = E - 1; r >= 0; --r) { sx[r] = sm; sm = min(sm, h[r]); }
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
                            const uint32_t off1 = row < 1 ? 0u : 0xFFFFFFFFu;
                            const uint32_t off2 = row < 2 ? 0u : 0xFFFFFFFFu;
                            const uint32_t off3 = row < 3 ? 0u : 0xFFFFFFFFu;
                            sm = min(min(sm, r1 | off1), min(r2 | off2, r3 | off3));
                        }
                        // min(d_last + C, C + min(pp, lp)) = C + min(d_last, pp, lp): d_last joins the prefix.
                        // For the lane's last slot min(pp, lp) is the inclusive scan value itself.
                        const int32_t dl = (int32_t)d_last;
                        const int32_t pp = min(__builtin_amdgcn_update_dpp((int)0x7FFFFFFF, (int)pm, 0x138, 0xF, 0xF, false), dl);
                        const int32_t pin = min(pm, dl);
                        const uint32_t after = QMCP_DPP(0xFFFFFFFFu, sm, 0x130, 0xF);
#pragma unroll
                        for (int r = 0; r < E; ++r) {
                            const int32_t pre = r == E - 1 ? pin : min(pp, lp[r]);
                            const uint32_t viaP = (uint32_t)((int32_t)C[r] + pre);
                            const uint32_t dnr = r == E - 1 ? min(viaP, after) : min(viaP, min(sx[r], after));
                            MW_AT(slot, k, Ly::kDn + r) = dnr;
                            h[r] = dnr + exj[r];
                        }
                        // the block's last distance, read back as a broadcast: cheaper for a lone wave than
                        // selecting the slot and v_readlane, and its latency hides under the next scans
                        d_last = slot0[((k * Ly::kWords + Ly::kDn + last_r) * 64) + last_lane];
                    }
                }
                // the flag raised for group g0+t-3 (read after the previous barrier) is looked at only
                // now, so the chain never waits for it
                if (flag_seen != kNoFail) { failed = g0 + t - 3; break; }
            } else {
                if (t >= 2) {
                    const uint32_t g = g0 + t - 2;
                    uint32_t* const slot = MW_SLOT(g % Ly::kSlots);
                    const uint32_t* const slot0 = MW_SLOT0(g % Ly::kSlots);
                    uint32_t first_bad = kNoFail;
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const uint32_t k = cblk + kk;
                        bool undercut = false;
                        uint32_t dn[E], hn[E];
                        uint32_t vm = 0xFFFFFFFFu;
#pragma unroll
                        for (int r = 0; r < E; ++r) {
                            dn[r] = MW_AT(slot, k, Ly::kDn + r);
                            hn[r] = dn[r] + MW_AT(slot, k, Ly::kEx + r);
                            vm = min(vm, hn[r]);
                        }
                        vm = min(vm, QMCP_DPP_UMIN(vm, 0x111, 0xF));
                        vm = min(vm, QMCP_DPP_UMIN(vm, 0x112, 0xF));
                        vm = min(vm, QMCP_DPP_UMIN(vm, 0x114, 0xF));
                        vm = min(vm, QMCP_DPP_UMIN(vm, 0x118, 0xF));
                        vm = min(vm, QMCP_DPP_UMIN(vm, 0x142, 0xA));
                        vm = min(vm, QMCP_DPP_UMIN(vm, 0x143, 0xC));
                        uint32_t run = QMCP_DPP(0xFFFFFFFFu, vm, 0x138, 0xF);
                        // d entering the lane: the lane below's last distance; lane 0: d entering the block
                        const uint32_t d_blk = k == 0 ? MW_AT(slot, 0, Ly::kDin)
                                                      : slot0[(((k - 1) * Ly::kWords + Ly::kDn + last_r) * 64) + last_lane];
                        uint32_t prev = QMCP_DPP(0u, dn[E - 1], 0x138, 0xF);
                        prev = lane == 0 ? d_blk : prev;
#pragma unroll
                        for (int r = 0; r < E; ++r) {
                            undercut |= run < dn[r];
                            run = min(run, hn[r]);
                            sel[kk][r] = MW_AT(slot, k, Ly::kX0 + r) + (MW_AT(slot, k, Ly::kCnt + r) - (dn[r] - prev));
                            prev = dn[r];
                        }
                        if (__any(undercut)) first_bad = min(first_bad, k);
                    }
                    if (first_bad != kNoFail && lane == 0) atomicMin(&s_flag[0], first_bad);
                }
            }
            MW_STAGE_BARRIER()  // (diagnostic builds: stamps around it, summed in registers)
            if (role == 1) {
                flag_seen = t >= 2 ? s_flag[0] : kNoFail;  // not waited for here
            } else {
                const uint32_t bad = t >= 2 ? s_flag[0] : kNoFail;
                // only now do results reach memory: every block before the first failed one is exact
                // (its check passed and so did those of all blocks before it)
                if (role == 2 && t >= 2) {
                    const uint32_t g = g0 + t - 2;
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        if (cblk + kk < bad) {
                            // one contiguous vector store per lane (its E slots are adjacent positions);
                            // lanes without E valid slots write the spare words behind the table
                            const uint32_t blk_first = (g * kG + cblk + kk) * ell;
                            const uint32_t p0 = blk_first + lane * E;
                            const bool full = lane * E + E <= ell && p0 + E <= Lrun;
                            if constexpr (E == 1) {
                                csel[full ? p0 : trash] = sel[kk][0];
                            } else {
                                typedef typename RowVec<E>::type V;
                                V v;
#pragma unroll
                                for (int r = 0; r < E; ++r) v[r] = sel[kk][r];
                                *reinterpret_cast<V*>(csel + (full ? p0 : trash)) = v;
                                // a lane with only some valid slots exists only if E does not divide the
                                // span or the block is cut by the contig's end (uniform test)
                                if (ell % E != 0 || blk_first + ell > Lrun) {
#pragma unroll
                                    for (int r = 0; r < E; ++r) {
                                        const uint32_t i = lane * E + r;
                                        if (!full && i < ell && blk_first + i < Lrun) csel[blk_first + i] = sel[kk][r];
                                    }
                                }
                            }
                        }
                    }
                }
                if (bad != kNoFail) { failed = g0 + t - 2; break; }
            }
        }
#ifdef QMCP_MW_STAMP
        const unsigned long long stamp_f0 = __builtin_amdgcn_s_memtime();
        stamp_prev = 0;
#endif
        // CHAIN leaves the loop one stage late (or at its end): the last flag it read is still unseen
        if (role == 1 && failed == 0xFFFFFFFFu && flag_seen != kNoFail) failed = g0 + n_left - 1;
        // every wave must agree on whether the pipeline failed: the flag itself says so
        __syncthreads();
        const uint32_t bad_blk = s_flag[0];
        if (bad_blk == kNoFail) { g0 = n_groups; break; }
        __syncthreads();  // everyone has read the flag
        if (threadIdx.x == 0) s_flag[0] = kNoFail;
        if (role == 1) {
            // state on entering the failed block: published at the group's entry, or rebuilt from the
            // (exact) distances of the block before it
            uint32_t* const slot = MW_SLOT(failed % Ly::kSlots);
            if (bad_blk == 0) {
#pragma unroll
                for (int r = 0; r < E; ++r) h[r] = MW_AT(slot, 0, Ly::kH0 + r);
                d_last = MW_AT(slot, 0, Ly::kDin);
            } else {
                uint32_t* const blk = slot + (bad_blk - 1) * Ly::kWords * 64;
#pragma unroll
                for (int r = 0; r < E; ++r) h[r] = MW_AT(blk, 0, Ly::kDn + r) + MW_AT(blk, 0, Ly::kEx + r);
                d_last = MW_SLOT0(failed % Ly::kSlots)[(((bad_blk - 1) * Ly::kWords + Ly::kDn + last_r) * 64) + last_lane];
            }
            sweep_full_run<E>(cb, failed * kG + bad_blk, failed * kG + kG, trash, ell, L, Lrun, M, lane, last_lane,
                              last_r, h, d_last, csel);
            n_full += kG - bad_blk;
        }
        // `failed` is known to every wave: CHAIN derived the same group one stage later.
        // An isolated failure (the usual case on deep data) costs only the rest of the failed group:
        // what PREP made for the two groups after it is still in LDS, so the next run starts warm.
        // A failure of the very first group of a run means the fast form keeps failing here: back off.
        penalty = failed > g0 ? 0u : min(2 * penalty + 1, 63u);
        warm = penalty == 0;
        g0 = failed + 1;
        __syncthreads();
#ifdef QMCP_MW_STAMP
        stamp_fail += __builtin_amdgcn_s_memtime() - stamp_f0;
        stamp_fails += 1;
#endif
    }
#undef MW_AT
#undef MW_SLOT
#undef MW_SLOT0
#undef MW_STAGE_BEGIN
#undef MW_STAGE_BARRIER
#ifdef QMCP_MW_STAMP
    if (lane == 0 && iter_stats) {
        atomicAdd(&iter_stats[4 + 2 * wv], (uint32_t)(stamp_work >> 4));
        atomicAdd(&iter_stats[5 + 2 * wv], (uint32_t)(stamp_wait >> 4));
        if (wv == 3) {
            atomicAdd(&iter_stats[20], (uint32_t)(stamp_fail >> 4));
            atomicAdd(&iter_stats[21], (uint32_t)stamp_fails);
            atomicAdd(&iter_stats[22], (uint32_t)((__builtin_amdgcn_s_memtime() - stamp_begin) >> 4));
            atomicAdd(&iter_stats[23], (uint32_t)(stamp_post >> 4));
            atomicAdd(&iter_stats[24], (uint32_t)stamp_iters);
        }
    }
#endif
    if (role == 1) {
        if (n_groups * kG < n_blocks)
            sweep_full_run<E>(cb, n_groups * kG, n_blocks, trash, ell, L, Lrun, M, lane, last_lane, last_r, h, d_last,
                              csel);
        if (iter_stats && lane == 0) {
            atomicAdd(&iter_stats[0], n_full);
            atomicAdd(&iter_stats[1], n_blocks);
            atomicAdd(&iter_stats[2], 1u);  // stretches swept
        }
    }
}

// ------------------------------------------------------------------ pipelined general form
// The same seven-wave pipeline for data where the fast form rarely holds (mean coverage within a few
// multiples of M): every block is solved in the GENERAL form, which needs no check and no rollback.
// A block is an inclusive scan of the maps (a, b, u, v) above; the (a, b) half depends only on the
// counts and ex -- not on the chain -- so the PREP waves run that half of the scan ahead of time and
// hand the chain wave, for each of the six scan steps, the (a, b) the current lane holds before the
// step (the "right operand" of the composition).  The chain wave is left with the (u, v) half:
//     u' = min(uL + a_s, vL, u),   v' = min(uL + b_s, vL, v)          (uL, vL: DPP reads)
// six instructions per step, plus the suffix-min of the previous block's h that feeds u and v.
// About 90 instructions per block against 50 for the fast form -- and against ~200 for the
// single-wave general block the fast kernel falls back to.
template <int E>
struct MgLayout {
    static constexpr int kG = E <= 3 ? 8 : 4; // blocks per group (LDS: 3 slots x kG x kWords x 256 B)
    static constexpr int kA = 0;              // [6] a before each scan step          PREP -> CHAIN
    static constexpr int kB = 6;              // [6] b before each scan step          PREP -> CHAIN
    static constexpr int kPa = 12;            // [1] a of all lower lanes (0: none)   PREP -> CHAIN
    static constexpr int kPb = 13;            // [1] b of all lower lanes (inf: none) PREP -> CHAIN
    static constexpr int kCnt = 14;           // [E] counts                           PREP -> CHAIN, CHECK
    static constexpr int kEx = 14 + E;        // [E] ex at the jump landing           PREP -> CHAIN
    static constexpr int kX0 = 14 + 2 * E;    // [E] bucket offsets                   PREP -> CHECK
    static constexpr int kDn = 14 + 3 * E;    // [E] distances                        CHAIN -> CHECK
    static constexpr int kWords = 14 + 4 * E;
    static constexpr int kSlots = 3;
    // + per slot one row of 64 words: d entering the group (CHAIN -> CHECK)
    static constexpr size_t kBytes = ((size_t)kSlots * kG * kWords + kSlots) * 64 * sizeof(uint32_t);
};

// wave-scan step s (row_shr 1, 2, 4, 8, row_bcast 15 into rows 1 and 3, row_bcast 31 into rows 2 and 3):
// does this lane take nothing from another lane at it?
__device__ __forceinline__ bool mg_no_source(int s, uint32_t lane) {
    if (s < 4) return (lane & 15u) < (1u << s);
    if (s == 4) return ((lane >> 4) & 1u) == 0;  // rows 0 and 2
    return lane < 32;                            // rows 0 and 1
}

// kAdj (round 4): the near-uniform route's sweep -- the need at every position is moved by nadj[] and capped at what the
// swept (regular) reads can give (block_terms_adj); PREP loads one more row per block, the nadj of its landing positions.
template <int E, bool kAdj>
__global__ __launch_bounds__(448) void k_sweep_uniform_gen(const uint32_t* __restrict__ boff,
                                                           const uint64_t* __restrict__ contig_pos_off,
                                                           uint32_t ell, uint32_t M, uint32_t ltot,
                                                           uint32_t* __restrict__ selend,
                                                           uint32_t* __restrict__ iter_stats,
                                                           const uint32_t* __restrict__ seg,
                                                           uint32_t* __restrict__ selend_run_in /* speculative tables: where a stretch's
                                                               run-in (the positions before the one it owns from) goes; or null */,
                                                           const uint32_t* __restrict__ redo_in /* or null: every stretch */,
                                                           uint32_t n_cand /* entries per column of seg (redo_in != null) */,
                                                           const int32_t* __restrict__ nadj /* kAdj: ltot + 1 entries */,
                                                           const uint32_t* __restrict__ own_marks /* or null; else per
                                                               stretch of THIS table: 0 = leave its output as it is (the
                                                               near-uniform route's later rounds: sweep_segments.inc.hip) */) {
    using Ly = MgLayout<E>;
    // (a later tier of a speculative sweep: only the parts of the genome a disagreement marked)
    if (redo_in != nullptr && (blockIdx.x >= seg[0] || spec_stretch_idle(seg, n_cand, blockIdx.x, redo_in))) return;
    if (own_marks != nullptr && own_marks[blockIdx.x] == 0) return;
    constexpr int kG = Ly::kG;
    extern __shared__ uint32_t s_mw[];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t role = wv == 3 ? 1u : (wv >= 5 ? 2u : 0u);  // 0 PREP, 1 CHAIN, 2 CHECK
    constexpr int kPB = kG / 4, kCB = kG / 2;                  // blocks per PREP wave / per CHECK wave
    const uint32_t pblk = kPB * (wv == 4 ? 3u : wv);           // PREP: first of its blocks of the group
    const uint32_t cblk = kCB * (wv - 5);                      // CHECK: first of its blocks
    const uint32_t c_id = blockIdx.x;
    SweepSeg sg;
    if (!sweep_segment(contig_pos_off, seg, c_id, sg)) return;  // uniform over the workgroup
    const uint32_t base = sg.base, L = sg.L, Lrun = sg.Lrun;
    const uint32_t n_blocks = (Lrun + ell - 1) / ell;
    const uint32_t n_groups = n_blocks / kG;
    const uint32_t* __restrict__ cb = boff + base;
    uint32_t* __restrict__ csel = selend + base;
    // a speculative stretch's run-in (a whole number of blocks) is stored apart: the stretch before it owns
    // those positions, and only the last block of it is looked at again (k_spec_verify)
    uint32_t* __restrict__ csel_run_in = csel;
    uint32_t own_blk = 0;
    if (selend_run_in != nullptr && seg != nullptr) {
        csel_run_in = selend_run_in + base;
        own_blk = (seg[1 + 3 * n_cand + c_id] - base) / ell;
    }
    const uint32_t trash = ltot - base;
    const uint32_t last_lane = (ell - 1) / E, last_r = (ell - 1) % E;
    __builtin_amdgcn_s_setprio(3);

    uint32_t h[E];
    sweep_initial_h<E>(boff, sg, ell, M, lane, h, kAdj ? nadj : nullptr);
    uint32_t d_last = 0;
    const uint32_t* __restrict__ nb = reinterpret_cast<const uint32_t*>(nadj) + base;  // (kAdj only)

#define MG_SLOT0(idx) (s_mw + (size_t)(idx) * kG * Ly::kWords * 64)
#define MG_SLOT(idx) (MG_SLOT0(idx) + lane)
#define MG_AT(slot, k, w) (slot)[((k) * Ly::kWords + (w)) * 64]
#define MG_DIN(idx) (s_mw + ((size_t)Ly::kSlots * kG * Ly::kWords + (idx)) * 64 + lane)[0]

    if (role == 0) {
        // PREP: one block of every group; rows three stages ahead in three register sets (see
        // k_sweep_uniform_mw); its own unrolled copy of the stage loop
        constexpr uint32_t kD = 3;
        constexpr int kRows = kPB + 2;  // rows its blocks need
        RowRaw<E> R0[kRows], R1[kRows], R2[kRows];
        RowRaw<E> N0[kPB], N1[kPB], N2[kPB];  // kAdj: nadj at the blocks' landing positions (the row one block on)
        auto issue_rows = [&](RowRaw<E> (&buf)[kRows], RowRaw<E> (&nbuf)[kPB], uint32_t g) {
#pragma unroll
            for (int k = 0; k < kRows; ++k) row_issue<E>(cb, (g * kG + pblk + k) * ell, L, lane, buf[k]);
            if constexpr (kAdj) {
#pragma unroll
                for (int k = 0; k < kPB; ++k) row_issue<E>(nb, (g * kG + pblk + k + 1) * ell, L, lane, nbuf[k]);
            }
        };
        issue_rows(R0, N0, 0);
        issue_rows(R1, N1, 1);
        issue_rows(R2, N2, 2);
        auto pstage = [&](RowRaw<E> (&buf)[kRows], RowRaw<E> (&nbuf)[kPB], uint32_t t) {
            if (t < n_groups) {
                uint32_t Wr[kRows][E];
                uint32_t Nr[kPB][E];
#pragma unroll
                for (int k = 0; k < kRows; ++k) row_finish<E>(buf[k], Wr[k]);
                if constexpr (kAdj) {
#pragma unroll
                    for (int k = 0; k < kPB; ++k) row_finish<E>(nbuf[k], Nr[k]);
                }
                __builtin_amdgcn_sched_barrier(0);
                issue_rows(buf, nbuf, t + kD);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kk = 0; kk < kPB; ++kk) {
                    uint32_t* const slot = MG_SLOT(t % Ly::kSlots) + (pblk + kk) * Ly::kWords * 64;
                    SweepLoads<E> ld;
                    rows_to_loads<E>(Wr[kk], Wr[kk + 1], Wr[kk + 2], lane, last_lane, last_r, ld);
                    BlockTerms<E> bt;
                    if constexpr (kAdj) {
                        int32_t adj[E];
#pragma unroll
                        for (int r = 0; r < E; ++r) adj[r] = (int32_t)Nr[kk][r];
                        block_terms_adj<E>(ld, adj, (t * kG + pblk + kk) * ell, ell, L, M, lane, bt);
                    } else {
                        block_terms<E>(ld, (t * kG + pblk + kk) * ell, ell, L, M, lane, bt);
                    }
                    // the lane's own (a, b): composition of its E single-position maps
                    uint32_t a = bt.cnt[0], b = bt.cnt[0] + bt.exj[0];
#pragma unroll
                    for (int r = 1; r < E; ++r) {
                        const uint32_t a2 = bt.cnt[r], b2 = bt.cnt[r] + bt.exj[r];
                        const uint32_t na = min(a + a2, b), nb = min(a + b2, b);
                        a = na; b = nb;
                    }
                    // the (a, b) half of the wave scan, recording what each lane holds before every step
#define MG_AB_STEP(s, ctrl, rmask)                                                         \
                    {                                                                      \
                        /* lanes that take nothing at this step (no lane to their left in the row; the rows \
                           the two cross-row steps skip): what they hold is never used there, and +inf in  \
                           its place lets the chain wave fold its shifted read of u into the add (a DPP    \
                           operand that reads 0 where there is no source lane) */          \
                        MG_AT(slot, 0, Ly::kA + (s)) = mg_no_source(s, lane) ? kInf : a;  \
                        MG_AT(slot, 0, Ly::kB + (s)) = mg_no_source(s, lane) ? kInf : b;  \
                        const uint32_t aL = QMCP_DPP(0u, a, ctrl, rmask);                  \
                        const uint32_t bL = QMCP_DPP(kInf, b, ctrl, rmask);                \
                        const uint32_t na = min(aL + a, bL), nb = min(aL + b, bL);         \
                        a = na; b = nb;                                                    \
                    }
                    MG_AB_STEP(0, 0x111, 0xF)
                    MG_AB_STEP(1, 0x112, 0xF)
                    MG_AB_STEP(2, 0x114, 0xF)
                    MG_AB_STEP(3, 0x118, 0xF)
                    MG_AB_STEP(4, 0x142, 0xA)
                    MG_AB_STEP(5, 0x143, 0xC)
#undef MG_AB_STEP
                    MG_AT(slot, 0, Ly::kPa) = QMCP_DPP(0u, a, 0x138, 0xF);     // all lower lanes (lane 0: identity)
                    MG_AT(slot, 0, Ly::kPb) = QMCP_DPP(kInf, b, 0x138, 0xF);
#pragma unroll
                    for (int r = 0; r < E; ++r) {
                        MG_AT(slot, 0, Ly::kCnt + r) = bt.cnt[r];
                        MG_AT(slot, 0, Ly::kEx + r) = bt.exj[r];
                        MG_AT(slot, 0, Ly::kX0 + r) = ld.x0[r];
                    }
                }
            }
            __syncthreads();
        };
        for (uint32_t t = 0;; t += kD) {
            if (t >= n_groups + 2) break;
            pstage(R0, N0, t);
            if (t + 1 >= n_groups + 2) break;
            pstage(R1, N1, t + 1);
            if (t + 2 >= n_groups + 2) break;
            pstage(R2, N2, t + 2);
        }
    } else {
        for (uint32_t t = 0; t < n_groups + 2; ++t) {
            if (role == 1) {
                if (t >= 1 && t <= n_groups) {
                    const uint32_t g = t - 1;
                    uint32_t* const slot = MG_SLOT(g % Ly::kSlots);
                    const uint32_t* const slot0 = MG_SLOT0(g % Ly::kSlots);
                    MG_DIN(g % Ly::kSlots) = d_last;
#pragma unroll
                    for (int k = 0; k < kG; ++k) {
                        uint32_t cnt[E], exj[E];
#pragma unroll
                        for (int r = 0; r < E; ++r) {
                            cnt[r] = MG_AT(slot, k, Ly::kCnt + r);
                            exj[r] = MG_AT(slot, k, Ly::kEx + r);
                        }
                        uint32_t as[6], bs[6];
#pragma unroll
                        for (int q = 0; q < 6; ++q) {
                            as[q] = MG_AT(slot, k, Ly::kA + q);
                            bs[q] = MG_AT(slot, k, Ly::kB + q);
                        }
                        const uint32_t pa = MG_AT(slot, k, Ly::kPa), pb = MG_AT(slot, k, Ly::kPb);
                        // A(i) = min over j >= i of the previous block's h: in-lane suffix + wave suffix
                        uint32_t A[E];
                        uint32_t sm = kInf;
#pragma unroll
                        for (int r = E - 1; r >= 0; --r) { sm = min(sm, h[r]); A[r] = sm; }
                        {
                            uint32_t ws = sm;
                            ws = min(ws, QMCP_DPP_UMIN(ws, 0x101, 0xF));
                            ws = min(ws, QMCP_DPP_UMIN(ws, 0x102, 0xF));
                            ws = min(ws, QMCP_DPP_UMIN(ws, 0x104, 0xF));
                            ws = min(ws, QMCP_DPP_UMIN(ws, 0x108, 0xF));
                            const uint32_t r1 = __builtin_amdgcn_readlane(ws, 16);
                            const uint32_t r2 = __builtin_amdgcn_readlane(ws, 32);
                            const uint32_t r3 = __builtin_amdgcn_readlane(ws, 48);
                            const uint32_t row = lane >> 4;
                            const uint32_t off1 = row < 1 ? 0u : 0xFFFFFFFFu;
                            const uint32_t off2 = row < 2 ? 0u : 0xFFFFFFFFu;
                            const uint32_t off3 = row < 3 ? 0u : 0xFFFFFFFFu;
                            ws = min(min(ws, r1 | off1), min(r2 | off2, r3 | off3));
                            // everything in the lanes above: the shifted read folds into each min (the last
                            // lane has no lane above and keeps its own)
#pragma unroll
                            for (int r = 0; r < E; ++r) A[r] = min(QMCP_DPP_UMIN(ws, 0x130, 0xF), A[r]);
                        }
                        // the lane's own (u, v)
                        uint32_t u = A[0], v = A[0] + exj[0];
#pragma unroll
                        for (int r = 1; r < E; ++r) {
                            const uint32_t nu = min(min(u + cnt[r], v), A[r]);
                            const uint32_t nv = min(min(u + cnt[r] + exj[r], v), A[r] + exj[r]);
                            u = nu; v = nv;
                        }
                        // the (u, v) half of the wave scan
#define MG_UV_STEP(s, ctrl, rmask)                                                     \
                        {                                                              \
                            const uint32_t uL = QMCP_DPP(kInf, u, ctrl, rmask);        \
                            const uint32_t vL = QMCP_DPP(kInf, v, ctrl, rmask);        \
                            const uint32_t nu = min(min(uL + as[s], vL), u);           \
                            const uint32_t nv = min(min(uL + bs[s], vL), v);           \
                            u = nu; v = nv;                                            \
                        }
                        // lanes that take nothing at a step read u as 0 (or some other row's) and find +inf in
                        // as / bs (PREP put it there), so the shifted read of u folds into the add
#define MG_UV_STEP_ROW(s, ctrl, vmask)                                                 \
                        {                                                              \
                            const uint32_t uL = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)u, ctrl, 0xF, 0xF, true); \
                            const uint32_t vL = QMCP_DPP(kInf, v, ctrl, vmask);        \
                            const uint32_t nu = min(min(uL + as[s], vL), u);           \
                            const uint32_t nv = min(min(uL + bs[s], vL), v);           \
                            u = nu; v = nv;                                            \
                        }
                        MG_UV_STEP_ROW(0, 0x111, 0xF)
                        MG_UV_STEP_ROW(1, 0x112, 0xF)
                        MG_UV_STEP_ROW(2, 0x114, 0xF)
                        MG_UV_STEP_ROW(3, 0x118, 0xF)
                        MG_UV_STEP_ROW(4, 0x142, 0xA)
                        MG_UV_STEP_ROW(5, 0x143, 0xC)
#undef MG_UV_STEP_ROW
#undef MG_UV_STEP
                        // state entering this lane: the map of all lower lanes applied to (d_last, +inf); the
                        // shifted reads of u and v fold into the mins (lane 0 has no lower lane: identity)
                        uint32_t dd = min(QMCP_DPP_UMIN(u, 0x138, 0xF), d_last + pa);
                        uint32_t m = min(QMCP_DPP_UMIN(v, 0x138, 0xF), d_last + pb);
#pragma unroll
                        for (int r = 0; r < E; ++r) {
                            dd = min(min(dd + cnt[r], m), A[r]);
                            MG_AT(slot, k, Ly::kDn + r) = dd;
                            h[r] = dd + exj[r];
                            m = min(m, h[r]);
                        }
                        d_last = slot0[((k * Ly::kWords + Ly::kDn + last_r) * 64) + last_lane];
                    }
                }
            } else {
                if (t >= 2) {
                    const uint32_t g = t - 2;
                    uint32_t* const slot = MG_SLOT(g % Ly::kSlots);
                    const uint32_t* const slot0 = MG_SLOT0(g % Ly::kSlots);
#pragma unroll
                    for (int kk = 0; kk < kCB; ++kk) {
                        const uint32_t k = cblk + kk;
                        uint32_t dn[E];
#pragma unroll
                        for (int r = 0; r < E; ++r) dn[r] = MG_AT(slot, k, Ly::kDn + r);
                        const uint32_t d_blk = k == 0 ? MG_DIN(g % Ly::kSlots)
                                                      : slot0[(((k - 1) * Ly::kWords + Ly::kDn + last_r) * 64) + last_lane];
                        uint32_t prev = QMCP_DPP(0u, dn[E - 1], 0x138, 0xF);
                        prev = lane == 0 ? d_blk : prev;
                        uint32_t sel[E];
#pragma unroll
                        for (int r = 0; r < E; ++r) {
                            sel[r] = MG_AT(slot, k, Ly::kX0 + r) + (MG_AT(slot, k, Ly::kCnt + r) - (dn[r] - prev));
                            prev = dn[r];
                        }
                        const uint32_t blk_first = (g * kG + k) * ell;
                        const uint32_t p0 = blk_first + lane * E;
                        const bool full = lane * E + E <= ell && p0 + E <= Lrun;
                        uint32_t* __restrict__ const out = g * kG + k < own_blk ? csel_run_in : csel;
                        if constexpr (E == 1) {
                            out[full ? p0 : trash] = sel[0];
                        } else {
                            typedef typename RowVec<E>::type V;
                            V vv;
#pragma unroll
                            for (int r = 0; r < E; ++r) vv[r] = sel[r];
                            *reinterpret_cast<V*>(out + (full ? p0 : trash)) = vv;
                            if (ell % E != 0 || blk_first + ell > Lrun) {
#pragma unroll
                                for (int r = 0; r < E; ++r) {
                                    const uint32_t i = lane * E + r;
                                    if (!full && i < ell && blk_first + i < Lrun) out[blk_first + i] = sel[r];
                                }
                            }
                        }
                    }
                }
            }
            __syncthreads();
        }
    }
#undef MG_AT
#undef MG_DIN
#undef MG_SLOT
#undef MG_SLOT0
    if (role == 1) {
        if (n_groups * kG < n_blocks) {
            // (the blocks behind the last whole group; those of them that are run-in go to the other output)
            const uint32_t t0 = n_groups * kG;
            const uint32_t mid = min(max(own_blk, t0), n_blocks);
            const int32_t* const nbt = kAdj ? nadj + base : nullptr;
            if (t0 < mid)
                sweep_full_run<E>(cb, t0, mid, trash, ell, L, Lrun, M, lane, last_lane, last_r, h, d_last, csel_run_in, nbt);
            if (mid < n_blocks)
                sweep_full_run<E>(cb, mid, n_blocks, trash, ell, L, Lrun, M, lane, last_lane, last_r, h, d_last, csel, nbt);
        }
        if (iter_stats && lane == 0) {
            atomicAdd(&iter_stats[0], n_blocks);  // every block is in the general form here
            atomicAdd(&iter_stats[1], n_blocks);
            atomicAdd(&iter_stats[2], 1u);  // stretches swept
        }
    }
}
