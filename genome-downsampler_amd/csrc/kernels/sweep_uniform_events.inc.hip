// sweep_uniform_events.inc.hip -- part of qmcp_kernels.hip (one translation unit; included inside namespace qmcp).
// ------------------------------------------------------------------ uniform sweep, event-driven form
// The same selection as k_sweep_uniform* (the canonical greedy of oracle/qmcp_oracle.c, which stands
// in for SimpleMaxFlow::Solve at quasi_mcp_cpu_max_flow_solver.cpp:19-20), stated on the KEPT counts
// instead of the dropped ones.  With S(p) = reads kept among those starting at p, need(p) =
// min(cov(p), M) and all spans equal to ell, the greedy's demand at position p is
//     delta(p) = need(p) - need(p-1) + S(p - ell)          (>= 0; the reads kept at p - ell expire at p)
// served from bucket p first and, when c(p) < delta(p), pushed back to p-1, p-2, ...: with
//     ov(p) = max(0, delta(p) + ov(p+1) - c(p))             (what position p hands on to p-1)
// the kept counts are S(p) = delta(p) + ov(p+1) - ov(p).  Over one block of ell positions the
// pushed-back amounts are a suffix Lindley recursion: a prefix sum T of t = delta - c and a suffix
// maximum of T give every ov at once; what leaves the block's first position goes into the block
// before it (top-down through its free room), whose kept counts feed delta of this block again --
// repeated until nothing more is handed back (rarely once; exhaustively checked against the oracle
// on random instances before this kernel was written).
//
// Why this form: where coverage stays above M (need(p) == need(p-1) == M, "deep"), delta of a block is
// just S of the block before it, and the block changes nothing -- S(p) = S(p - ell) for all its
// positions -- unless some c(p) < S(p - ell).  On deep data that is rare (cfg4: 8.5 % of the blocks,
// nearly all of them early in a contig), so the chain wave only has to TEST a block:
//   k_sweep_pack    (whole chip) packs every block's counts into one word per lane -- E saturating
//                   fields of 30/E bits, the top bit of each a guard, + a flag for blocks the shortcut
//                   applies to (deep, inside the contig, not the first) -- four blocks per lane as
//                   one 16-byte piece;
//   k_sweep_uniform_ev  (one wave per contig or stretch) streams the pieces into an LDS ring by LDS-DMA,
//                   48 KiB ahead of itself, and tests eight blocks at a time: word - packed profile
//                   keeps a guard bit exactly where the count is not below the profile's, so the AND
//                   of eight differences says "nothing changes" in 17 vector instructions.  A block
//                   that fails takes the one-step form (whatever a position cannot serve is taken by
//                   the position before it: one shifted read) or, if that does not settle it, the two
//                   wave scans above.  It writes S only for the blocks that changed, and for every
//                   block the index of the last changed block at or before it;
//   k_sweep_expand  (whole chip) writes selend[p] = boff[p] + S(p) for every position from those.
// Exact for any input (shallow data merely flags every block); the host picks it for deep calls
// whose M is below the field maximum: no kept count reaches a saturated field's value then (at most
// M reads are ever kept from one start position), so a saturated count behaves like the true one.

template <int E> struct EvPack {
    static constexpr uint32_t kW = 30 / E;                  // field pitch: a count and, above it, a guard bit
    static constexpr uint32_t kSat = (1u << (kW - 1)) - 1u; // field maximum: "at least this many"
    // guard bits: set in every block word, clear in the packed profile; word - profile clears a guard
    // exactly where a count is below the profile's (the borrow stops there), so "all guards still set"
    // over any number of AND-ed differences says that none of those blocks changes anything
    static constexpr uint32_t kGuard = E == 1 ? (1u << 29)
                                     : E == 2 ? ((1u << 14) | (1u << 29))
                                     : E == 3 ? ((1u << 9) | (1u << 19) | (1u << 29))
                                              : ((1u << 6) | (1u << 13) | (1u << 20) | (1u << 27));
    static constexpr uint32_t kPlain = 0x80000000u;         // deep block inside the contig (not the first)
    static constexpr uint32_t kAll = kGuard | kPlain;
};
static constexpr uint32_t kEvSlots = 64;    // LDS ring: 64 pieces of 1 KiB
static constexpr int32_t kEvNeg = -(1 << 30);

// A stretch's place in the per-piece / per-block side arrays: pieces hold four blocks, a stretch of
// Lrun positions has at most Lrun / (4 ell) + 1 of them, so base / (4 ell) + index never overlaps.
struct EvGeom { uint32_t base, Lrun, L, n_blocks, piece_base; };
__device__ __forceinline__ uint32_t ev_piece_base(uint32_t base, uint32_t idx, uint32_t ell) { return base / (4u * ell) + idx; }
__device__ __forceinline__ bool ev_geom(const uint64_t* __restrict__ contig_pos_off, const uint32_t* __restrict__ seg,
                                        uint32_t idx, uint32_t ell, EvGeom& g) {
    SweepSeg sg;
    const bool ok = sweep_segment(contig_pos_off, seg, idx, sg);
    g.base = sg.base; g.Lrun = ok ? sg.Lrun : 0u; g.L = ok ? sg.L : 0u;
    g.n_blocks = (g.Lrun + ell - 1) / ell;
    g.piece_base = ev_piece_base(g.base, idx, ell);
    return ok;
}
// the stretch that owns global piece w (false: none)
__device__ __forceinline__ bool ev_find(const uint64_t* __restrict__ contig_pos_off, const uint32_t* __restrict__ seg,
                                        uint32_t n_wg, uint32_t ell, uint32_t w, EvGeom& g, uint32_t& idx) {
    const uint32_t count = seg ? min(seg[0], n_wg) : n_wg;
    if (count == 0) return false;
    auto pb = [&](uint32_t s) {
        const uint32_t b = seg ? seg[1 + 3 * s] : (uint32_t)contig_pos_off[s];
        return ev_piece_base(b, s, ell);
    };
    uint32_t lo = 0, hi = count;
    if (pb(0) > w) return false;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (pb(mid) <= w) lo = mid; else hi = mid;
    }
    idx = lo;
    if (!ev_geom(contig_pos_off, seg, lo, ell, g)) return false;
    return w - g.piece_base < (g.n_blocks + 3) / 4;
}

template <int E>
__global__ __launch_bounds__(256) void k_sweep_pack(const uint32_t* __restrict__ boff,
                                                    const uint64_t* __restrict__ contig_pos_off, uint32_t n_wg,
                                                    uint32_t ell, uint32_t M, uint32_t ltot,
                                                    const uint32_t* __restrict__ seg, uint32_t n_pieces_max,
                                                    uint32_t* __restrict__ pk,
                                                    const int32_t* __restrict__ nadj /* near-uniform route: need(p) += nadj[p]; else null */,
                                                    const uint32_t* __restrict__ from /* per stretch: first block the chain will sweep, or null */) {
    using P = EvPack<E>;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n_pieces_max) return;
    EvGeom g;
    uint32_t idx;
    if (!ev_find(contig_pos_off, seg, n_wg, ell, w, g, idx)) return;
    const uint32_t q = w - g.piece_base;
    if (from != nullptr && (uint64_t)4u * q + 3u < (uint64_t)from[idx]) return;  // (what changed lies at or beyond the chain's first block)
    uint32_t out[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t k = 4 * q + j;
        const uint32_t p0 = g.base + k * ell + lane * E;   // global position of the lane's first slot
        uint32_t X[E + 1], Pv[E + 1];
#pragma unroll
        for (int r = 0; r <= E; ++r) {
            const uint32_t p = p0 + r;
            X[r] = boff[min(p, ltot)];
            // (not `p >= ell ? min(p - ell, ltot) : 0`: hipcc 7.2 folds that select away and reads
            // boff[min(p - ell wrapped, ltot)] = boff[ltot] for p < ell -- clamp first, subtract after)
            const uint32_t pm = min(p, ltot + ell);
            Pv[r] = boff[pm >= ell ? pm - ell : 0u];
        }
        uint32_t word = 0;
        bool deep = true;
#pragma unroll
        for (int r = 0; r < E; ++r) {
            const uint32_t i = lane * E + r;
            const bool valid = i < ell && k * ell + i < g.L;
            const uint32_t c = valid ? X[r + 1] - X[r] : 0u;
            word |= min(c, P::kSat) << (r * P::kW);
            // need(p) == need(p - 1) == M
            if (i < ell) deep = deep && (X[r + 1] - Pv[r + 1] >= M) && (X[r] - Pv[r] >= M);
            if (nadj != nullptr && i < ell) {
                const uint32_t p = min(p0 + r, ltot);
                deep = deep && nadj[p] == 0 && nadj[p > 0 ? p - 1 : 0u] == 0;
            }
        }
        const bool inside = (uint64_t)(k + 1) * ell <= g.L;
        const bool plain = k != 0 && k < g.n_blocks && inside && __all(deep);
        out[j] = word | P::kGuard | (plain ? P::kPlain : 0u);
    }
    uint4 v;
    v.x = out[0]; v.y = out[1]; v.z = out[2]; v.w = out[3];
    reinterpret_cast<uint4*>(pk)[(size_t)w * 64 + lane] = v;
}

// LDS-DMA of one 1-KiB piece (16 bytes per lane) to the wave-uniform LDS byte address lds_dst
__device__ __forceinline__ void ev_glds16(const void* gsrc, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

__device__ __forceinline__ int32_t ev_incl_suffix_max(int32_t v, uint32_t lane) {
#define QMCP_EV_DPP(x, ctrl) __builtin_amdgcn_update_dpp((int)kEvNeg, (int)(x), (ctrl), 0xF, 0xF, false)
    v = max(v, QMCP_EV_DPP(v, 0x101));
    v = max(v, QMCP_EV_DPP(v, 0x102));
    v = max(v, QMCP_EV_DPP(v, 0x104));
    v = max(v, QMCP_EV_DPP(v, 0x108));
#undef QMCP_EV_DPP
    const int32_t r1 = __builtin_amdgcn_readlane(v, 16);
    const int32_t r2 = __builtin_amdgcn_readlane(v, 32);
    const int32_t r3 = __builtin_amdgcn_readlane(v, 48);
    const uint32_t row = lane >> 4;
    v = max(max(v, row < 1 ? r1 : kEvNeg), max(row < 2 ? r2 : kEvNeg, row < 3 ? r3 : kEvNeg));
    return v;
}

// One block in the general (kept-count) form: demand d[] against counts c[]; returns what is handed
// back past the block's first position and leaves the kept counts in S[].
template <int E>
__device__ __forceinline__ uint32_t ev_block_step(const int32_t (&d)[E], const uint32_t (&c)[E], uint32_t lane,
                                                  uint32_t (&S)[E]) {
    int32_t T[E], Ml[E];
    int32_t ls = 0;
#pragma unroll
    for (int r = 0; r < E; ++r) { ls += d[r] - (int32_t)c[r]; T[r] = ls; }
    const int32_t incl = (int32_t)wave_incl_scan_add((uint32_t)ls);
    const int32_t before = __builtin_amdgcn_update_dpp(0, incl, 0x138, 0xF, 0xF, false);  // lane 0: 0
#pragma unroll
    for (int r = 0; r < E; ++r) T[r] += before;
    int32_t mx = kEvNeg;
#pragma unroll
    for (int r = E - 1; r >= 0; --r) { mx = max(mx, T[r]); Ml[r] = mx; }
    const int32_t suf = ev_incl_suffix_max(mx, lane);
    const int32_t after = __builtin_amdgcn_update_dpp((int)kEvNeg, suf, 0x130, 0xF, 0xF, false);  // lane 63: none
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const int32_t mx_here = max(Ml[r], after);
        const int32_t mx_next = r + 1 < E ? max(Ml[r + 1 < E ? r + 1 : r], after) : after;
        const int32_t t_prev = r > 0 ? T[r > 0 ? r - 1 : 0] : before;
        const int32_t ov = max(0, mx_here - t_prev);
        const int32_t ovn = max(0, mx_next - T[r]);
        S[r] = (uint32_t)(d[r] + ovn - ov);
    }
    const int32_t top = __builtin_amdgcn_readlane(max(Ml[0], after), 0);
    return (uint32_t)max(0, top);
}

template <int E>
__global__ __launch_bounds__(128) void k_sweep_uniform_ev(const uint32_t* __restrict__ boff,
                                                         const uint64_t* __restrict__ contig_pos_off,
                                                         uint32_t n_contigs, uint32_t ell, uint32_t M, uint32_t ltot,
                                                         const uint32_t* __restrict__ pk,
                                                         uint32_t* __restrict__ sev,      // ltot + 8: S of changed blocks
                                                         uint32_t* __restrict__ lastns,   // per block: last changed block <= it
                                                         uint32_t* __restrict__ iter_stats,
                                                         const uint32_t* __restrict__ seg,
                                                         const int32_t* __restrict__ nadj /* see k_sweep_pack */,
                                                         // near-uniform route: the chain's state entering every 64th block
                                                         // is kept (512 words each), and a later sweep of the same stretch
                                                         // may start from one of them: restart[stretch] = first block to
                                                         // sweep (a multiple of 64; beyond the stretch: nothing to do)
                                                         uint32_t* __restrict__ ckpt, const uint32_t* __restrict__ restart) {
    using P = EvPack<E>;
    extern __shared__ uint4 s_evring[];
    // two waves: wave 1 only moves pieces into the LDS ring (an LDS-DMA request costs its issuer ~96
    // cycles, lab/dma_lab.hip -- more than the chain spends on the four blocks in it), wave 0 is the chain
    __shared__ uint32_t s_ctl_words[2];  // [0] pieces landed, [1] pieces read
    // (an LDS-qualified pointer, taken by value below: through a generic one these become flat_ loads)
    typedef __attribute__((address_space(3))) uint32_t LdsWord;
    volatile LdsWord* const s_ctl = (volatile LdsWord*)s_ctl_words;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    EvGeom gm;
    if (!ev_geom(contig_pos_off, seg, blockIdx.x, ell, gm)) return;
    const uint32_t base = gm.base, L = gm.L, Lrun = gm.Lrun, n_blocks = gm.n_blocks;
    const uint32_t n_pieces = (n_blocks + 3) / 4;
    const uint32_t k0 = restart != nullptr ? (uint32_t)__builtin_amdgcn_readfirstlane((int)restart[blockIdx.x]) & ~63u : 0u;
    if (k0 >= n_blocks && restart != nullptr) return;  // (uniform: both waves)
    const uint32_t q0 = k0 / 4;  // first piece to sweep
    uint32_t* const my_ckpt = ckpt != nullptr ? ckpt + ((size_t)(gm.piece_base / 16) + 2u * blockIdx.x) * 512u : nullptr;
    const uint4* __restrict__ src = reinterpret_cast<const uint4*>(pk) + (size_t)gm.piece_base * 64 + lane;
    uint32_t* __restrict__ my_last = lastns + (size_t)gm.piece_base * 4;
    const uint32_t ring0 = (uint32_t)(uintptr_t)s_evring;
    if (threadIdx.x == 0) { s_ctl[0] = q0; s_ctl[1] = q0; }
    __syncthreads();
    if (wv == 1) {
        // LOADER: keeps the ring full.  A slot is reused once the chain has said it read the piece in it;
        // "landed" is published 32 requests behind the issue point (counted wait).
        __builtin_amdgcn_s_setprio(1);
        uint32_t read_seen = q0;
        for (uint32_t idx = q0; idx < n_pieces; ++idx) {
            while (idx - read_seen >= kEvSlots) {
                read_seen = s_ctl[1];
                if (idx - read_seen >= kEvSlots) __builtin_amdgcn_s_sleep(2);
            }
            ev_glds16(src + (size_t)idx * 64, ring0 + (idx % kEvSlots) * 1024u);
            if ((idx & 3) == 3 && idx >= q0 + 32) {
                asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
                if (lane == 0) s_ctl[0] = idx + 1 - 32;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no request may outlive the workgroup's LDS
        if (lane == 0) s_ctl[0] = n_pieces;
        return;
    }
    __builtin_amdgcn_s_setprio(3);

    // is the stretch's first position a contig's first position?  (a stretch that starts behind a
    // cut point inherits the reads kept across the cut)
    bool contig_start = seg == nullptr;
    if (seg != nullptr) {
        bool hit = false;
        for (uint32_t cc = lane; cc < n_contigs; cc += 64) hit |= (uint32_t)contig_pos_off[cc] == base;
        contig_start = __any(hit);
    }

    uint32_t g[E];       // S of the block before the current one: this block's expiries
    uint32_t gp = 0;     // the same, packed like a piece word (guards clear)
    uint32_t cprevw = 0; // the previous block's packed counts (room for what is handed back)
#pragma unroll
    for (int r = 0; r < E; ++r) {
        g[r] = 0;
        if (!contig_start) {
            // behind a cut point every read covering it is kept: S(p) = c(p) on the ell positions before
            const uint32_t i = lane * E + r;
            const uint32_t qpos = base + i;  // position base - ell + i, shifted by ell
            if (i < ell && qpos >= ell) g[r] = boff[qpos - ell + 1] - boff[qpos - ell];
        }
    }
    uint32_t last_ns = 0, n_changed = 0;
    uint32_t lastv = 0;  // lane j: last changed block at or before block (current group of 64) + j
    if (k0 != 0) {
        // the state an earlier sweep of this stretch had when it entered block k0
        const uint32_t* ck = my_ckpt + (size_t)(k0 / 64) * 512u + lane * 8u;
        gp = 0;
#pragma unroll
        for (int r = 0; r < E; ++r) { g[r] = ck[r]; gp |= g[r] << (r * P::kW); }
        cprevw = ck[4];
        last_ns = (uint32_t)__builtin_amdgcn_readfirstlane((int)ck[5]);
        lastv = last_ns;
    }

#ifdef QMCP_EV_STAMP
    unsigned long long st_gen = 0, st_slow = 0, st_wait = 0, st_fail = 0;
    uint32_t st_slow_pieces = 0, st_full = 0, st_fail_rounds = 0;
    const unsigned long long st_begin = __builtin_amdgcn_s_memtime();
#endif
    // pieces [0, landed) are in the ring; the chain looks at the ring's progress word only when it has
    // caught up with what it last saw there
    uint32_t landed = q0;
    auto wait_for = [&](uint32_t upto) {  // pieces [0, min(upto, n_pieces)) have landed
        const uint32_t want = min(upto, n_pieces);
#ifdef QMCP_EV_STAMP
        const unsigned long long w0 = __builtin_amdgcn_s_memtime();
#endif
        while (landed < want) {
            landed = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_ctl[0]);
            if (landed < want) __builtin_amdgcn_s_sleep(1);
        }
#ifdef QMCP_EV_STAMP
        st_wait += __builtin_amdgcn_s_memtime() - w0;
#endif
    };
    wait_for(q0 + 4);
    uint4 cur[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) cur[i] = s_evring[((q0 + i) % kEvSlots) * 64 + lane];

    // the block's kept counts S[] become the profile; they and the block's index are recorded
    auto commit = [&](uint32_t k, const uint32_t (&S)[E], uint32_t cword) {
        gp = 0;
#pragma unroll
        for (int r = 0; r < E; ++r) {
            g[r] = S[r];
            gp |= S[r] << (r * P::kW);   // S <= M < the field maximum (the launcher checks M)
        }
        {
            // one contiguous store per lane (its E slots are adjacent positions); lanes without E valid
            // slots write the spare words behind the table, and their valid slots one by one
            const uint32_t blk_first = k * ell;
            const uint32_t p0 = blk_first + lane * E;
            const bool full = lane * E + E <= ell && p0 + E <= Lrun;
            if constexpr (E == 1) {
                sev[full ? base + p0 : ltot] = S[0];
            } else {
                typedef typename RowVec<E>::type V;
                V v;
#pragma unroll
                for (int r = 0; r < E; ++r) v[r] = S[r];
                *reinterpret_cast<V*>(sev + (full ? base + p0 : ltot)) = v;
                if (ell % E != 0 || blk_first + ell > Lrun) {
#pragma unroll
                    for (int r = 0; r < E; ++r) {
                        const uint32_t i = lane * E + r;
                        if (!full && i < ell && blk_first + i < Lrun) sev[base + blk_first + i] = S[r];
                    }
                }
            }
        }
        cprevw = cword;
        last_ns = k;
        lastv = lane >= (k & 63) ? k : lastv;
        ++n_changed;
    };

    // The same for a deep block that lies whole inside the stretch and whose lanes' slots are whole (ell a multiple of E:
    // a lane has all E slots or none): no partial lanes, no per-slot stores.  The lane's byte offset inside a block is a
    // constant; lanes without slots store to the spare words behind the table.
    const bool lane_has_slots = lane * E + E <= ell;
    const uint32_t n_lean = ell % E == 0 ? Lrun / ell : 0u;  // blocks [0, n_lean) may take it
    auto commit_plain = [&](uint32_t k, const uint32_t (&S)[E], uint32_t cword) {
        gp = 0;
#pragma unroll
        for (int r = 0; r < E; ++r) {
            g[r] = S[r];
            gp |= S[r] << (r * P::kW);   // S <= M < the field maximum (the launcher checks M)
        }
        uint32_t* const at = sev + (lane_has_slots ? (size_t)base + (size_t)k * ell : (size_t)ltot);  // (uniform part + the lane's constant)
        if constexpr (E == 1) {
            at[lane_has_slots ? lane : 0u] = S[0];
        } else {
            typedef typename RowVec<E>::type V;
            V v;
#pragma unroll
            for (int r = 0; r < E; ++r) v[r] = S[r];
            *reinterpret_cast<V*>(at + (lane_has_slots ? lane * E : 0u)) = v;
        }
        cprevw = cword;
        last_ns = k;
        lastv = lane >= (k & 63) ? k : lastv;
        ++n_changed;
    };

    // one block that is not known to be unchanged
    auto general_block = [&](uint32_t k, uint32_t word) {
        uint32_t c[E];
        uint32_t S[E];
        // (the flag is the same in every lane: read on the scalar unit, so that the branches on it are scalar branches
        //  and not exec-masked regions)
        const bool plain = ((uint32_t)__builtin_amdgcn_readfirstlane((int)word) & P::kPlain) != 0;
        if (plain) {
            // Deep block: the demand is the profile itself.  Nearly always whatever a position cannot
            // serve is taken by the position just before it: one shifted read instead of two scans.
            //   ex = what the slot cannot serve = max(g - c, 0) (one saturating subtract), S = min(g, c) + ex of the
            //   slot after it; valid iff no receiving slot overflows in turn -- S <= c everywhere: where nothing comes in
            //   S = min(g, c) <= c, where something does S <= c means g + in <= c -- and nothing is handed back past the
            //   block's first position (lane 0's first ex).  One maximum over the differences and one ballot decide.
            uint32_t ex[E], mn[E];
#pragma unroll
            for (int r = 0; r < E; ++r) {
                c[r] = (word >> (r * P::kW)) & P::kSat;
                ex[r] = __builtin_elementwise_sub_sat(g[r], c[r]);
                mn[r] = g[r] - ex[r];
            }
            const uint32_t up = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)ex[0], 0x130, 0xF, 0xF, false);  // next lane's first slot
            int32_t viol = lane == 0 ? (int32_t)ex[0] : 0;
#pragma unroll
            for (int r = 0; r < E; ++r) {
                const uint32_t in = r + 1 < E ? ex[r + 1 < E ? r + 1 : r] : up;
                S[r] = mn[r] + in;
                viol = max(viol, (int32_t)S[r] - (int32_t)c[r]);
            }
            if (__builtin_amdgcn_ballot_w64(viol > 0) == 0) {
                if (k < n_lean) commit_plain(k, S, word);  // (uniform)
                else commit(k, S, word);
                return;
            }
        }
#ifdef QMCP_EV_STAMP
        ++st_full;
#endif
        int32_t dn[E];
        bool kill[E];
        if (!plain) {
#pragma unroll
            for (int r = 0; r < E; ++r) {
                const uint32_t i = lane * E + r;
                const uint32_t pos = k * ell + i;
                const bool valid = i < ell && pos < L;
                const uint32_t p = base + min(pos, L);
                const uint32_t a1 = boff[min(p + 1, ltot)], a0 = boff[p];
                const uint32_t b1 = boff[p + 1 >= ell ? p + 1 - ell : 0u];  // (<= ltot: p <= ltot, ell >= 1)
                const uint32_t b0 = boff[p >= ell ? p - ell : 0u];
                int32_t need_p = (int32_t)min(a1 - b1, M);
                int32_t need_m = (int32_t)min(a0 - b0, M);
                if (nadj != nullptr) {
                    // (capped at what the swept reads can give: where the other reads are needed to reach the need --
                    //  a contig's last positions -- the demand could not be met, and what cannot be met is handed back
                    //  through the whole block before, beyond the positions that could have served it; the near-uniform
                    //  route's verification works on the uncapped need and selects those other reads)
                    need_p = min(need_p + nadj[min(p, ltot)], (int32_t)(a1 - b1));
                    need_m = min(need_m + nadj[p > 0 ? p - 1 : 0u], (int32_t)(a0 - b0));
                }
                if (pos == 0 && contig_start) need_m = 0;
                c[r] = valid ? a1 - a0 : 0u;
                dn[r] = valid ? need_p - need_m : 0;
                kill[r] = !valid;
            }
        } else {
#pragma unroll
            for (int r = 0; r < E; ++r) { dn[r] = 0; kill[r] = false; }
        }
        uint32_t pushed = 0;
        for (uint32_t round = 0; round < 4096; ++round) {  // (bounded: every round hands back at least one more read)
            int32_t d[E];
#pragma unroll
            for (int r = 0; r < E; ++r) d[r] = kill[r] ? 0 : max((int32_t)g[r] + dn[r], 0);  // (never negative for a true need;
                                                                                              //  the near-uniform route's trial sweeps may ask)
            const uint32_t e = ev_block_step<E>(d, c, lane, S);
            if (e == pushed) break;
            // hand e - pushed more back into the block before: top-down through its free room
            const uint32_t rem = e - pushed;
            pushed = e;
            uint32_t room[E], pre[E];
            uint32_t ls = 0;
#pragma unroll
            for (int r = 0; r < E; ++r) {
                const uint32_t cp = (cprevw >> (r * P::kW)) & P::kSat;
                room[r] = cp > g[r] ? cp - g[r] : 0u;
                ls += room[r];
                pre[r] = ls;
            }
            const uint32_t incl = wave_incl_scan_add(ls);
            const uint32_t before = QMCP_DPP(0u, incl, 0x138, 0xF);
            const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
#pragma unroll
            for (int r = 0; r < E; ++r) {
                const uint32_t above = total - (before + pre[r]);  // free room in the slots above this one
                const uint32_t left = rem > above ? rem - above : 0u;
                g[r] += min(room[r], left);
            }
            if (k > 0) {
                // the block before changed after all: record it
#pragma unroll
                for (int r = 0; r < E; ++r) {
                    const uint32_t i = lane * E + r;
                    const uint32_t pos = (k - 1) * ell + i;
                    if (i < ell && pos < Lrun) sev[base + pos] = g[r];
                }
                if (((k - 1) >> 6) == (k >> 6)) {
                    lastv = lane >= ((k - 1) & 63) ? k - 1 : lastv;
                } else if (lane == 0) {
                    my_last[k - 1] = k - 1;
                }
            }
            if (total < rem) break;  // cannot happen for a feasible demand; do not spin
        }
        uint32_t cword = word;
        if (!plain) {
            cword = P::kGuard;
#pragma unroll
            for (int r = 0; r < E; ++r) cword |= min(c[r], P::kSat) << (r * P::kW);
        }
        commit(k, S, cword);
    };

    // the four blocks of one piece, one after the other (the profile may change under them)
    auto slow_piece = [&](uint32_t q, const uint4& w) {
#ifdef QMCP_EV_STAMP
        const unsigned long long s0 = __builtin_amdgcn_s_memtime();
        ++st_slow_pieces;
#endif
        uint32_t wj = w.x, w1 = w.y, w2 = w.z, w3 = w.w;  // (rotated, not indexed: an indexed vector goes to scratch)
#pragma unroll 1
        for (uint32_t j = 0; j < 4; ++j, wj = w1, w1 = w2, w2 = w3) {
            const uint32_t k = 4 * q + j;
            if (k >= n_blocks) break;
            if (__builtin_amdgcn_ballot_w64(((wj - gp) & P::kAll) != P::kAll) != 0) {
#ifdef QMCP_EV_STAMP
                const unsigned long long g0 = __builtin_amdgcn_s_memtime();
#endif
                general_block(k, wj);
#ifdef QMCP_EV_STAMP
                st_gen += __builtin_amdgcn_s_memtime() - g0;
#endif
            } else {
                cprevw = wj;
            }
        }
#ifdef QMCP_EV_STAMP
        st_slow += __builtin_amdgcn_s_memtime() - s0;
#endif
    };

    // Four pieces (sixteen blocks) per round.  An UNCHANGED round is the common case late in a contig and must be cheap:
    // up to four of them run back to back on two register sets that swap roles (the next round's pieces are read from
    // the ring while this one is tested; as first written the loop moved sixteen registers, tested (q & 15) three times
    // and compared `landed` on the vector unit every round -- 76 instructions against 26 of test proper).  A round with
    // a block that changes the profile leaves that run and is taken piece by piece in ONE copy of the slow path, which
    // reads its pieces from the ring again (they stay until the chain says it is 64 blocks further).
    auto piece_and = [&](const uint4& w) { return ((w.x - gp) & (w.y - gp)) & ((w.z - gp) & (w.w - gp)); };
    // the bookkeeping at the end of the round that starts at piece q: every 64 blocks (and at the stretch's end) the
    // last-changed index of each of them, and the state entering the next 64
    auto round_done = [&](uint32_t q) {
        if ((q & 15) != 12 && q + 4 < n_pieces) return;  // (uniform)
        const uint32_t kb = (q >> 4) * 64 + lane;
        if (kb < n_blocks) my_last[kb] = lastv;
        lastv = last_ns;
        if (my_ckpt != nullptr && (q & 15) == 12) {
            uint32_t* ck = my_ckpt + (size_t)((q >> 4) + 1u) * 512u + lane * 8u;
#pragma unroll
            for (int r = 0; r < E; ++r) ck[r] = g[r];
            ck[4] = cprevw;
            ck[5] = last_ns;
        }
        if (lane == 0) s_ctl[1] = q + 4 < n_pieces ? q + 4 : q;  // every piece before the next group has been read
    };
    uint32_t q = q0;
    bool failed = false;
    // one unchanged round on the register set `now`, the next round's pieces into `nxt`
#define QMCP_EV_ROUND(now, nxt)                                                                                      \
    if (!failed && q < n_pieces) {                                                                                   \
        if (landed < min(q + 8, n_pieces)) wait_for(q + 8);                                                          \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) nxt[i] = s_evring[(((q + 4) % kEvSlots) + i) * 64 + lane];     \
        const uint32_t all = (piece_and(now[0]) & piece_and(now[1])) & (piece_and(now[2]) & piece_and(now[3]));      \
        if (__builtin_amdgcn_ballot_w64((all & P::kAll) != P::kAll) != 0) {                                          \
            failed = true;                                                                                           \
        } else {                                                                                                     \
            cprevw = now[3].w;                                                                                       \
            round_done(q);                                                                                           \
            q += 4;                                                                                                  \
        }                                                                                                            \
    }
    uint4 alt[4];
    while (q < n_pieces) {
        QMCP_EV_ROUND(cur, alt)
        QMCP_EV_ROUND(alt, cur)
        QMCP_EV_ROUND(cur, alt)
        QMCP_EV_ROUND(alt, cur)
        if (failed) {
            // some block of round q changes the profile: piece by piece, each tested against the profile as it is then
            failed = false;
#ifdef QMCP_EV_STAMP
            const unsigned long long f0 = __builtin_amdgcn_s_memtime();
            ++st_fail_rounds;
#endif
            uint4 w[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) w[i] = s_evring[((q % kEvSlots) + i) * 64 + lane];  // (q is a multiple of 4)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (q + i < n_pieces) {
                    if (__builtin_amdgcn_ballot_w64((piece_and(w[i]) & P::kAll) != P::kAll) != 0) slow_piece(q + i, w[i]);
                    else cprevw = w[i].w;
                }
            }
            round_done(q);
            q += 4;
            if (q < n_pieces) {
                if (landed < min(q + 4, n_pieces)) wait_for(q + 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) cur[i] = s_evring[((q % kEvSlots) + i) * 64 + lane];
            }
#ifdef QMCP_EV_STAMP
            st_fail += __builtin_amdgcn_s_memtime() - f0;
#endif
        }
    }
#undef QMCP_EV_ROUND
    if (iter_stats && lane == 0) {
        atomicAdd(&iter_stats[0], n_changed);
        atomicAdd(&iter_stats[1], n_blocks);
        atomicAdd(&iter_stats[2], 1u);  // stretches swept
#ifdef QMCP_EV_STAMP
        atomicAdd(&iter_stats[4], (uint32_t)((__builtin_amdgcn_s_memtime() - st_begin) >> 4));
        atomicAdd(&iter_stats[5], (uint32_t)(st_gen >> 4));
        atomicAdd(&iter_stats[6], (uint32_t)(st_wait >> 4));
        atomicAdd(&iter_stats[7], (uint32_t)(st_slow >> 4));
        atomicAdd(&iter_stats[8], st_slow_pieces);
        atomicAdd(&iter_stats[9], (uint32_t)(st_fail >> 4));
        atomicAdd(&iter_stats[10], st_fail_rounds);
        atomicAdd(&iter_stats[11], st_full);
#endif
    }
}

// selend[p] = boff[p] + S(p) for every position: S of the last changed block at or before p's block
template <int E>
__global__ __launch_bounds__(256) void k_sweep_expand(const uint32_t* __restrict__ boff,
                                                      const uint64_t* __restrict__ contig_pos_off, uint32_t n_wg,
                                                      uint32_t ell, uint32_t ltot, const uint32_t* __restrict__ seg,
                                                      uint32_t n_pieces_max, const uint32_t* __restrict__ sev,
                                                      const uint32_t* __restrict__ lastns,
                                                      uint32_t* __restrict__ selend,
                                                      const uint32_t* __restrict__ from /* per stretch: first block the chain swept, or null */) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t gb = blockIdx.x * 4 + (threadIdx.x >> 6);  // global block slot
    const uint32_t w = gb >> 2;
    if (w >= n_pieces_max) return;
    EvGeom g;
    uint32_t idx;
    if (!ev_find(contig_pos_off, seg, n_wg, ell, w, g, idx)) return;
    const uint32_t k = gb - 4 * g.piece_base;
    if (k >= g.n_blocks) return;
    if (from != nullptr && (uint64_t)k + 1u < (uint64_t)from[idx]) return;  // (the chain's first block may hand back into the one before)
    const uint32_t kk = lastns[gb];
    const uint32_t back = (k - kk) * ell;
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const uint32_t i = lane * E + r;
        const uint32_t pos = k * ell + i;
        if (i < ell && pos < g.Lrun) {
            const uint32_t p = g.base + pos;
            selend[p] = boff[p] + sev[p - back];
        }
    }
}
