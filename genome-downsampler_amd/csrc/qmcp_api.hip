// qmcp_api.hip -- context, device arena and stage orchestration behind include/qmcp_hip.h.
//
// Stage order of one solve (all on the context's stream):
//   prepare   validate, span min/max, global start keys, reads-per-start counts
//   scan      exclusive scan of the counts -> bucket offsets boff (and eoff on the mixed path)
//   sort      LSD radix bucketing of read indices by (start[, span desc]), stable in index
//   sweep     selection: block-parallel shortest-path sweep (uniform span) or the
//             event-driven priority sweep (mixed spans); one wave per contig
//   mark      keep bitmask from per-bucket selected prefixes
// One host round trip (12 bytes) after `prepare` picks the path and the radix width.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "qmcp_hip.h"
#include "qmcp_kernels.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess)                                                              \
            return fail(_e == hipErrorOutOfMemory ? QMCP_ENOMEM : QMCP_EHIP, "%s: %s (%s:%d)", \
                        #expr, hipGetErrorString(_e), __FILE__, __LINE__);                 \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

enum Ev { EV_BEGIN = 0, EV_PREP, EV_SCAN, EV_SORT, EV_SWEEP, EV_MARK, EV_COUNT };

struct Problem {
    uint64_t n = 0;
    uint32_t n_contigs = 0;
    uint64_t ltot = 0;
    std::vector<uint64_t> poff;
};

// what the two halves of a solve's enqueue share (see enqueue_head)
struct SolveRun {
    const uint32_t* d_starts = nullptr;
    const uint32_t* d_ends = nullptr;
    const uint64_t* roff = nullptr;     // host; valid until enqueue_tail has returned
    const uint32_t* lengths = nullptr;  // host; likewise
    uint32_t n_contigs = 0, M = 0;
    uint64_t n64 = 0;
    uint64_t* d_mask = nullptr;
    Problem pr;
    qmcp_hip_stats local;
    bool trivial = false, head_done = false, may_rank = false, two_level = false, have_gstart = true;
    bool ranked_counted = false, wait_empty = false;
    bool pm = false;                    // range-ranked route in its pass-major form (kernels/pass_major.inc.hip)
    uint32_t range_shift = 0;
    uint32_t nu_filter = 0;             // near-uniform route: the span the head's producer treated as regular (0: every read)
};

}  // namespace

struct qmcp_hip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[EV_COUNT] = {};
    hipEvent_t ev_in = nullptr;
    hipStream_t stream2 = nullptr;  // side stream: small read-backs beside the work queued on `stream`
    hipEvent_t ev_fork = nullptr;   // main stream -> side stream: statistics and heaviest load are final
    // arena (grow-only, reused across solves like a reference solver instance's members)
    DevBuf roff, poff, stats, cstart, boff, ecnt, eoff, selend, spine, hist, spine2, hist2, specsnap, specflags;
    DevBuf keys[2], vals[2];
    DevBuf in_starts, in_ends, in_aux0, in_aux1, mask, cov, amp, next_head;
    DevBuf f_starts, f_ends, f_map, f_words, f_mask;  // filter -> solve pipeline
    DevBuf ranges;     // range-ranked path: 257 range starts + heaviest load
    DevBuf rankamb;    // range-ranked path: per-range lists of quota-crossing groups settled after the walk
    // pass-major form (kernels/pass_major.inc.hip): one descriptor word per wave-slot; k_pm_descr's working words + the
    // ranges' counts of quota-crossing groups
    DevBuf pm_desc, pm_work;
    // near-uniform route (kernels/near_uniform.inc.hip): the dominant span of the last call that took it -- the next
    // call's head filters on it at once -- and the route's buffers
    uint32_t nu_ell = 0;
    // a call of this shape did not settle within its budget of rounds (or met a run the replay does not model): the next
    // one goes straight to the mixed-span route instead of burning the budget again
    uint64_t nu_failed_n = 0, nu_failed_ltot = 0;
    uint32_t nu_failed_ell = 0, nu_failed_M = 0;
    DevBuf nu_exc, nu_nadj, nu_ce, nu_state, nu_sus, nu_ckpt;
    uint32_t* h_nu = nullptr;       // pinned landing zone of the route's state words (8)
    uint64_t* h_tables = nullptr;  // pinned staging for the contig tables (2 x (n_contigs + 1))
    size_t h_tables_cap = 0;
    unsigned long long* h_scalars = nullptr;  // pinned landing zone of the solve's result scalars (4 words)
    // qmcp_hip_solve_host64: pinned staging, two slots per narrowing thread, and a pinned mask landing zone
    uint32_t* h_stage = nullptr;
    size_t h_stage_words = 0;
    uint64_t* h_mask = nullptr;
    size_t h_mask_words = 0;
    std::vector<hipEvent_t> stage_ev;
    std::vector<hipStream_t> stage_streams;  // copy streams: one DMA engine moves ~29 GB/s, PCIe twice that
    std::vector<hipEvent_t> stage_done;
    // a solve that has been enqueued but not yet completed (qmcp_hip_solve_device_begin / _end)
    bool pending = false;
    qmcp_hip_stats pend_stats;
    uint32_t pend_whole_contig_chains = 0;  // mixed spans without cut points: one chain per non-empty contig
    // positions that start no read, as counted by the last range-ranked solve (picks the sweep kernel of the next)
    bool spiky_known = false, pend_spiky = false;
    uint64_t spiky_n = 0, spiky_ltot = 0;
    uint32_t spiky_empty = 0;
    // the mixed-span route's speculative boundaries disagreed nearly everywhere on the last call of this shape (data
    // that forgets its state slowly: one dominant read length, deep): the next call of the shape does not speculate
    uint64_t spec_hopeless_n = 0, spec_hopeless_ltot = 0;
    uint32_t spec_hopeless_M = 0;
    uint32_t pend_M = 0;
    DevBuf scalars;  // popcount + sweep iteration counters
    DevBuf segs;     // cut-point windows and the sweep's stretch table
    DevBuf rings;    // mixed spans beyond 16 383: the plain event sweep's rings, in global memory
    DevBuf kidx;          // qmcp_hip_kept_indices_host: the expanded index list
    uint64_t mask_reads = 0;  // reads the context's own mask buffer (c->mask) currently describes
    DevBuf evpk, evlast;  // event-driven uniform sweep: packed block words, last-changed-block index per block
    uint32_t last_iters = 0, last_blocks = 0;
    // the two halves of a solve's enqueue (enqueue_head / enqueue_tail) and what they share
    SolveRun run;
    uint32_t* h_head = nullptr;       // pinned landing zone of the read-back that picks the route (8 words)
    hipEvent_t ev_head = nullptr;     // the solve's head (prepare, partition, bucket offsets) has been queued up to here
    hipEvent_t ev_done = nullptr;     // everything of the solve has been queued up to here
    bool mixed_seen = false;          // a call took the mixed-span route: its arrays are sized up front from then on
    bool sized = false;               // the arena block of the current solve is behind us (growth now is growth mid-solve)
    uint32_t grew_mid_solve = 0;      // buffers that had to grow after the solve's first launch (stats.arena_grown_mid_solve)
    // optional per-kernel timing (qmcp_hip_set_profiling): one event pair per launch group
    int profiling = 0;  // 0 off, 1 every kernel, 2 the selection sweep only
    size_t tables_count = 0;          // contig tables currently on the device (upload_tables)
    void* tables_dev_roff = nullptr;
    void* tables_dev_poff = nullptr;
    struct Span { const char* name; hipEvent_t a, b; };
    std::vector<Span> spans;          // spans of the solve in flight
    std::vector<hipEvent_t> ev_pool;  // recycled events
    struct Acc { std::string name; uint64_t launches; double ms; };
    std::vector<Acc> acc;             // accumulated since the last reset
};

namespace {

hipEvent_t pool_event(qmcp_hip_ctx* c) {
    if (!c->ev_pool.empty()) {
        hipEvent_t e = c->ev_pool.back();
        c->ev_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

// RAII bracket around one kernel (or one kernel + its helper launches) when profiling is on
struct KernelSpan {
    qmcp_hip_ctx* c;
    hipEvent_t a = nullptr, b = nullptr;
    const char* name;
    hipStream_t st;
    KernelSpan(qmcp_hip_ctx* ctx, const char* nm, hipStream_t stream = nullptr)
        : c(ctx), name(nm), st(stream ? stream : ctx->stream) {
        if (!c->profiling) return;
        if (c->profiling == 2 && std::strncmp(nm, "k_sweep", 7) != 0) return;
        a = pool_event(c);
        b = pool_event(c);
        if (a) (void)hipEventRecord(a, st);
    }
    ~KernelSpan() {
        if (!c->profiling || !a || !b) return;
        (void)hipEventRecord(b, st);
        c->spans.push_back({name, a, b});
    }
};

void collect_spans(qmcp_hip_ctx* c) {
    for (auto& sp : c->spans) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) {
            bool found = false;
            for (auto& a : c->acc)
                if (a.name == sp.name) { a.launches++; a.ms += ms; found = true; break; }
            if (!found) c->acc.push_back({sp.name, 1, ms});
        }
        c->ev_pool.push_back(sp.a);
        c->ev_pool.push_back(sp.b);
    }
    c->spans.clear();
}

int ensure(qmcp_hip_ctx* c, DevBuf& b, size_t bytes) {
    if (bytes == 0) bytes = 16;
    if (b.cap >= bytes) return QMCP_OK;
    if (b.p) {
        if (c->sized) c->grew_mid_solve++;  // (after the solve's arena block: a stall on work already queued)
        // growing a buffer frees it: nothing queued on this context may still be using the old one
        // (hipFree would wait for the whole device anyway -- this names the wait and keeps it to the
        // one case where a later call is larger than every earlier one)
        if (c->stream) HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->stream2) HIP_TRY(hipStreamSynchronize(c->stream2));
        HIP_TRY(hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    HIP_TRY(hipMalloc(&b.p, bytes));
    b.cap = bytes;
    return QMCP_OK;
}

#define TRY(expr)                      \
    do {                               \
        int _rc = (expr);              \
        if (_rc != QMCP_OK) return _rc; \
    } while (0)

uint32_t bit_width(uint32_t v) { return v == 0 ? 0u : 32u - (uint32_t)__builtin_clz(v); }

int check_problem(const uint64_t* roff, const uint32_t* lengths, uint32_t n_contigs, uint64_t n,
                  Problem& pr) {
    if (!roff || !lengths || n_contigs == 0) return fail(QMCP_EINVAL, "contig tables missing or n_contigs == 0");
    if (roff[0] != 0 || roff[n_contigs] != n)
        return fail(QMCP_EINVAL, "contig_read_offsets must start at 0 and end at n_reads");
    pr.poff.assign((size_t)n_contigs + 1, 0);
    for (uint32_t c = 0; c < n_contigs; ++c) {
        if (roff[c + 1] < roff[c]) return fail(QMCP_EINVAL, "contig_read_offsets not monotone at %u", c);
        pr.poff[c + 1] = pr.poff[c] + lengths[c];
    }
    pr.n = n;
    pr.n_contigs = n_contigs;
    pr.ltot = pr.poff[n_contigs];
    if (n > (1ull << 30)) return fail(QMCP_ERANGE, "n_reads %llu exceeds 2^30 per call", (unsigned long long)n);
    if (pr.ltot > (1ull << 31) - 2)
        return fail(QMCP_ERANGE, "total contig length %llu exceeds 2^31-2", (unsigned long long)pr.ltot);
    return QMCP_OK;
}

int upload_tables(qmcp_hip_ctx* c, const uint64_t* roff, const Problem& pr) {
    const size_t count = (size_t)pr.n_contigs + 1;
    const size_t bytes = count * sizeof(uint64_t);
    TRY(ensure(c, c->roff, bytes));
    TRY(ensure(c, c->poff, bytes));
    // staged through pinned memory owned by the context: the copies are truly asynchronous and
    // nothing has to wait for them on the host (the previous solve has fully completed)
    if (c->h_tables_cap < 2 * count) {
        if (c->h_tables) HIP_TRY(hipHostFree(c->h_tables));
        c->h_tables = nullptr;
        HIP_TRY(hipHostMalloc((void**)&c->h_tables, 2 * bytes, hipHostMallocDefault));
        c->h_tables_cap = 2 * count;
    }
    // the device copies stay valid across solves: skip the upload when nothing changed (a caller
    // that solves the same genome repeatedly saves two small copies per call)
    if (c->tables_count == count && c->tables_dev_roff == c->roff.p && c->tables_dev_poff == c->poff.p &&
        std::memcmp(c->h_tables, roff, bytes) == 0 &&
        std::memcmp(c->h_tables + count, pr.poff.data(), bytes) == 0)
        return QMCP_OK;
    std::memcpy(c->h_tables, roff, bytes);
    std::memcpy(c->h_tables + count, pr.poff.data(), bytes);
    c->tables_count = count;
    c->tables_dev_roff = c->roff.p;
    c->tables_dev_poff = c->poff.p;
    HIP_TRY(hipMemcpyAsync(c->roff.p, c->h_tables, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->poff.p, c->h_tables + count, bytes, hipMemcpyHostToDevice, c->stream));
    return QMCP_OK;
}

// prepare + host round trip.  Leaves gstart (global start position per read) in vals[1] when
// want_keys; counts reads per start position into cstart (global atomics) only when
// want_counts -- the solve derives its bucket offsets from the sorted keys instead.
int run_prepare(qmcp_hip_ctx* c, const uint32_t* d_starts, const uint32_t* d_ends,
                const Problem& pr, const uint64_t* d_keep_mask, bool want_keys, bool want_counts,
                bool want_part_hist, uint32_t part_shift, uint32_t* d_global_digit_hist,
                uint32_t host_stats[3]) {
    const uint32_t n = (uint32_t)pr.n;
    TRY(ensure(c, c->stats, 4 * sizeof(uint32_t)));
    if (want_counts) TRY(ensure(c, c->cstart, ((size_t)pr.ltot + 1) * sizeof(uint32_t)));
    if (want_keys) TRY(ensure(c, c->vals[1], (size_t)n * sizeof(uint32_t)));
    const uint32_t init[4] = {0xFFFFFFFFu, 0u, 0u, 0u};
    HIP_TRY(hipMemcpyAsync(c->stats.p, init, sizeof(init), hipMemcpyHostToDevice, c->stream));
    if (want_counts)
        HIP_TRY(hipMemsetAsync(c->cstart.p, 0, ((size_t)pr.ltot + 1) * sizeof(uint32_t), c->stream));
    {
        KernelSpan sp(c, "k_prepare");
        qmcp::launch_prepare(c->stream, d_starts, d_ends, n, (const uint64_t*)c->roff.p,
                             (const uint64_t*)c->poff.p, pr.n_contigs, d_keep_mask,
                             want_keys ? (uint32_t*)c->vals[1].p : nullptr,
                             want_counts ? (uint32_t*)c->cstart.p : nullptr, (uint32_t*)c->stats.p,
                             part_shift, want_part_hist ? (uint32_t*)c->hist2.p : nullptr, nullptr,
                             d_global_digit_hist, nullptr);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(host_stats, c->stats.p, 3 * sizeof(uint32_t), hipMemcpyDeviceToHost,
                           c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (host_stats[2] != 0)
        return fail(QMCP_EREAD, "a read has start > end or end >= its contig length");
    return QMCP_OK;
}

int scan_counts(qmcp_hip_ctx* c, DevBuf& counts, DevBuf& out, uint32_t ltot) {
    TRY(ensure(c, out, ((size_t)ltot + 1) * sizeof(uint32_t)));
    TRY(ensure(c, c->spine, (size_t)qmcp::scan_spine_entries(ltot) * sizeof(uint32_t) + 16));
    {
        KernelSpan sp(c, "scan_positions(3 kernels)");
        qmcp::launch_exclusive_scan(c->stream, (const uint32_t*)counts.p, ltot, (uint32_t*)out.p,
                                    (uint32_t*)c->spine.p, true);
    }
    HIP_TRY(hipGetLastError());
    return QMCP_OK;
}

// the ranked path is taken when no position range holds more than 1/kRankBalance of the reads:
// a range's ranking is one wave's serial walk (~0.65 ns per read) against ~0.03 ns per read for
// the radix sort it replaces
constexpr uint64_t kRankBalance = 24;
// mean coverage / M below which the sweep runs every block in the general form (lab/sweep_lab.hip)
constexpr double kGenDepth = 11.0;  // lab, cycles per block fast / general: 674 / 542 at 9 x M, 595 / 545 at 10.5, 500 / 543 at 12
// ... and when the call is large enough for a per-range workgroup to have work (QMCP_HIP_RANK_MIN
// overrides, for experiments)
static uint32_t rank_min_reads() {
    const char* e = std::getenv("QMCP_HIP_RANK_MIN");
    return e ? (uint32_t)std::strtoul(e, nullptr, 10) : (1u << 17);
}

// shortest span the event-driven sweep is used for: its scratch is 256 bytes per block, i.e. grows as
// the span shrinks; at 32 positions it is 8 bytes per position, what the bucket offsets themselves take
static uint32_t ev_min_span() { return 32u; }

float elapsed(hipEvent_t a, hipEvent_t b) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, a, b) != hipSuccess) return 0.f;
    return ms;
}

// The uniform-span sweep: seven waves per contig where the span allows it (fast form with checked
// fallback on deep data, every block in the general form on shallow data -- both exact, the
// choice is about speed only), else the single-wave kernel.  QMCP_HIP_SWEEP=fast|gen overrides.
// Cut-point segmentation of the uniform sweeps (QMCP_HIP_CUTS=0|1 overrides): looked for where mean
// coverage is a small multiple of M -- deep data has no cut points, and the look costs two launches.
uint32_t sweep_cut_windows(uint32_t ltot, uint32_t span, uint32_t n_contigs, bool shallow,
                           uint32_t max_windows = qmcp::kSweepWindowsOneSpan) {
    bool on = shallow;
    if (const char* e = std::getenv("QMCP_HIP_CUTS")) on = e[0] == '1';
    return on ? qmcp::sweep_segment_windows(ltot, span, n_contigs, max_windows) : 0u;
}

// Speculative stretch boundaries: below this mean coverage (in units of M), with a run-in (in blocks)
// that grows with the depth.  lab/spec_burn_study.py, cfg5's shape at 1/32 scale, boundaries that
// disagreed at a run-in of 128 / 256 / 512 / 1024 blocks: depth 2.0: 2 of 364 / 0 / 0 / 0; 2.5: 67 of 364 /
// 1 of 240 / 0 / 0; 3.0: 157 / 28 / 0 of 118 / 0; 4.0: 273 / 86 / 6 of 118 / 0 of 56 -- about twice
// the run-in per half unit of depth.  Two tiers: the first with the run-in of this table, and -- only
// if some boundary disagreed -- a second with three times that (or, where the genome is too short for it,
// none: the exact table); the exact sweep runs only if the second tier disagrees somewhere too.  Every
// tier's launches are queued at once and gated by device words, so nothing waits for the host.
// Round 3 (lab/spec_depth_gap.py, one contig of 20 M positions at 100 x coverage, profiles/r03_spec_depth_gap.log):
// between 4.1 and 11 x M -- where round 2 swept whole contigs as one chain each -- the sweep forgets its start too,
// within about a thousand blocks: boundaries that disagreed at a run-in of 256 / 512 / 1024 / 2048 blocks: depth 4.2:
// 55 of 127 / 2 of 63 / 0 of 31 / 0; 5.9: 85 / 11 / 0 / 0; 8.3: 100 / 20 / 0 / 0; 10: 108 / 23 / 1 of 31 / 0 of 15 --
// sweep 30.9 -> 1.3 ms.  So every depth the general-form sweep takes (below kGenDepth) is speculated on; at cfg4's
// depth (18.75, and at 37.5) every boundary still disagrees at 2 048 blocks (lab/spec_deep_probe.py): the event-driven
// chain stays whole there.
constexpr double kSpecDepth = kGenDepth, kSpecMinDepth = 1.3;
uint32_t spec_burn_blocks(double depth) {
    return depth < 2.1 ? 320u : depth < 2.6 ? 640u : depth < 3.1 ? 1152u : depth < 4.1 ? 2304u : 1536u;
}
bool spec_wanted(double depth) {
    bool on = depth < kSpecDepth && depth > kSpecMinDepth;  // (shallower: nearly every window has a real cut point)
    if (const char* e = std::getenv("QMCP_HIP_SPEC")) on = e[0] == '1';  // (0 / 1: never / at any depth)
    return on;
}
uint32_t spec_first_run_in(double depth) {
    if (const char* e = std::getenv("QMCP_HIP_SPEC_BURN")) return (uint32_t)std::strtoul(e, nullptr, 10);
    return spec_burn_blocks(depth);
}

// device words of a speculative sweep, behind the solve's other scalars
struct SpecWords {
    uint32_t* mismatches1;  // tier 1: boundaries that disagreed
    uint32_t* n_spec1;      //         speculative boundaries
    uint32_t* mismatches2;  // tier 2
    uint32_t* n_spec2;
};
SpecWords spec_words(qmcp_hip_ctx* c) {
    uint32_t* w = (uint32_t*)((char*)c->scalars.p + 32);
    return SpecWords{w, w + 1, w + 2, w + 3};
}

// The tiers of a speculative sweep.  `unit`: positions per block of run-in (the span; the largest span of a
// mix), `round_to`: the run-in is made a multiple of this many positions.  sweep(table, second output or null,
// marks to obey or null) launches the sweep kernel; check(table, mismatch counter, marks to obey or
// null, marks to set) the comparison and the merge behind it.  A disagreement marks the exact stretch it
// lies in; tier 2 (three times the run-in) sweeps only marked parts, the exact sweep only what tier 2 marked.
template <class Sweep, class Check>
int speculative_sweep(qmcp_hip_ctx* c, hipStream_t st, uint32_t n_contigs, uint32_t ltot, uint32_t windows,
                      uint32_t unit, uint32_t round_to, uint32_t burn_blocks, uint32_t run_ins_apart,
                      const uint32_t* seg_exact, const char* sweep_name, Sweep sweep, Check check) {
    const SpecWords w = spec_words(c);
    const uint64_t* poff = (const uint64_t*)c->poff.p;
    const uint32_t n_cand = n_contigs + windows;
    TRY(ensure(c, c->specflags, 2 * (size_t)n_cand * sizeof(uint32_t)));
    uint32_t* redo1 = (uint32_t*)c->specflags.p;
    uint32_t* redo2 = redo1 + n_cand;
    HIP_TRY(hipMemsetAsync(redo1, 0, 2 * (size_t)n_cand * sizeof(uint32_t), st));
    auto positions = [&](uint64_t blocks) { return (uint32_t)((blocks * unit + round_to - 1) / round_to * round_to); };
    const uint32_t burn1 = positions(burn_blocks);
    uint32_t burn2 = positions(3ull * burn_blocks);
    if ((uint64_t)ltot < 2ull * run_ins_apart * burn2) burn2 = 0;  // too short a genome: tier 2 is the exact table
    const uint32_t *seg1, *seg2;
    {
        KernelSpan sp(c, "k_find_cuts", st);
        seg1 = qmcp::launch_sweep_segments_speculative(st, poff, n_contigs, ltot, windows, burn1, (uint32_t*)c->segs.p,
                                                       w.n_spec1, run_ins_apart, 1);
        seg2 = qmcp::launch_sweep_segments_speculative(st, poff, n_contigs, ltot, windows, burn2, (uint32_t*)c->segs.p,
                                                       w.n_spec2, run_ins_apart, 2);
    }
    // the second output: one span -- every stretch's run-in; a mix of spans -- the odd stretches' whole output
    uint32_t* second_out = (uint32_t*)c->cstart.p;
    {
        KernelSpan sp(c, sweep_name, st);
        if (!sweep(seg1, second_out, nullptr)) return fail(QMCP_ERANGE, "speculative sweep: span not supported");
    }
    {
        KernelSpan sp(c, "k_spec_verify + k_spec_merge", st);
        check(seg1, w.mismatches1, nullptr, redo1);
    }
    {
        KernelSpan sp(c, "second tier, where the first disagreed", st);
        (void)sweep(seg2, second_out, redo1);
        check(seg2, w.mismatches2, redo1, redo2);
    }
    KernelSpan sp(c, "exact sweep, where the second tier disagreed", st);
    (void)sweep(seg_exact, nullptr, redo2);
    HIP_TRY(hipGetLastError());
    return QMCP_OK;
}

int launch_uniform_sweep(qmcp_hip_ctx* c, hipStream_t st, uint32_t n, uint32_t ltot, uint32_t n_contigs,
                         uint32_t span, uint32_t M, uint32_t* d_iters, uint32_t empty_positions,
                         bool* expand_left_out = nullptr /* in: the caller can read the event sweep's own output;
                                                            out: the event sweep ran whole contigs and selend[] was not written */) {
    const bool may_leave_expand = expand_left_out != nullptr && *expand_left_out;
    if (expand_left_out) *expand_left_out = false;
    // mean coverage in units of M: the fast form needs the binding jumps to come from the previous
    // block, which holds while coverage is many times M
    const double depth = (double)n * (double)span / ((double)ltot * (double)(M ? M : 1));
    bool gen = depth < kGenDepth;
    if (const char* e = std::getenv("QMCP_HIP_SWEEP")) {
        if (std::strcmp(e, "gen") == 0) gen = true;
        if (std::strcmp(e, "fast") == 0) gen = false;
    }
    const uint32_t* boff = (const uint32_t*)c->boff.p;
    const uint64_t* poff = (const uint64_t*)c->poff.p;
    uint32_t* selend = (uint32_t*)c->selend.p;
    // shallow or gapped data: split the contigs at cut points so that more than n_contigs chains run
    const uint32_t* seg = nullptr;
    uint32_t n_seg_max = 0;
    const uint32_t windows = sweep_cut_windows(ltot, span, n_contigs, gen);
    // Data a few times deeper than M: hardly any cut points, but the sweep forgets its start within tens
    // of blocks (kernels/sweep_segments.inc.hip), so windows without a cut get a speculative boundary with a
    // run-in (every few windows, so that stretches stay several run-ins long); the stretches' outputs are compared where they
    // meet, and if any pair disagrees the exact sweep runs after all (its launch is there either way and
    // returns at once when all agreed).
    const uint32_t burn_blocks = spec_first_run_in(depth);
    const bool speculate = spec_wanted(depth) && gen && windows != 0 && qmcp::sweep_uniform_mw_supported(span) &&
                           burn_blocks >= 2 && (uint64_t)ltot >= 8ull * burn_blocks * span;
    if (windows != 0) {
        KernelSpan sp(c, "k_find_cuts", st);
        seg = qmcp::launch_sweep_segments(st, boff, nullptr, poff, n_contigs, ltot, span, M, windows, (uint32_t*)c->segs.p);
        n_seg_max = n_contigs + windows;
    }
    // deep data: the event-driven form (a block is only TESTED unless its counts fall below the kept
    // profile); spans below ev_min_span() would need more scratch than the arena holds for it
    // ... and only where few blocks have a start position that holds no read: such a block nearly always
    // changes the kept profile, and a changed block costs the event-driven chain ~6 x the block-scan
    // pipeline's chain step (amplicon panels, whose reads start in a few windows: cfg3 took 0.16 ms against
    // 0.05).  With a fraction z of empty positions about 1 - (1 - z)^span of the blocks have one: more than
    // half of them from z = ln 2 / span on.  (Unknown on the small-call route: block scan, as in round 1.)
    // (no read can start in the last span - 1 positions of a contig: those are not holes in the data)
    const double structural = (double)n_contigs * (double)(span - 1);
    const double holes = (double)empty_positions > structural ? (double)empty_positions - structural : 0.0;
    const bool spiky = empty_positions == 0xFFFFFFFFu || holes * (double)span > 0.693 * (double)ltot;
    bool ev = !gen && !spiky && span >= ev_min_span();
    if (const char* e = std::getenv("QMCP_HIP_SWEEP")) {
        if (std::strcmp(e, "ev") == 0) ev = span >= ev_min_span();
        if (std::strcmp(e, "fast") == 0 || std::strcmp(e, "gen") == 0) ev = false;
    }
    if (ev && qmcp::sweep_uniform_ev_supported(span, M)) {
        // scratch of the event-driven form: 256 bytes per block, so it depends on the span, which is only
        // known here -- grown on the first deep call of a size (ensure() waits for the streams then), kept after
        {
            const uint32_t wg_max = n_contigs + 768;
            TRY(ensure(c, c->evpk, qmcp::sweep_ev_pack_bytes(ltot, span, wg_max)));
            TRY(ensure(c, c->evlast, qmcp::sweep_ev_last_bytes(ltot, span, wg_max)));
        }
        uint32_t* pk = (uint32_t*)c->evpk.p;
        uint32_t* sev = (uint32_t*)c->cstart.p;
        uint32_t* lastns = (uint32_t*)c->evlast.p;
        {
            KernelSpan sp(c, "k_sweep_pack", st);
            qmcp::launch_sweep_ev_pack(st, boff, poff, n_contigs, span, M, ltot, seg, n_seg_max, pk);
        }
        {
            KernelSpan sp(c, "k_sweep_uniform_ev", st);
            qmcp::launch_sweep_ev_chain(st, boff, poff, n_contigs, span, M, ltot, seg, n_seg_max, pk, sev, lastns, d_iters);
        }
        if (may_leave_expand && seg == nullptr) {
            *expand_left_out = true;  // (the ranking reads sev / lastns itself)
            return QMCP_OK;
        }
        KernelSpan sp(c, "k_sweep_expand", st);
        qmcp::launch_sweep_ev_expand(st, boff, poff, n_contigs, span, M, ltot, seg, n_seg_max, sev, lastns, selend);
        return QMCP_OK;
    }
    if (speculate && seg != nullptr) {
        return speculative_sweep(
            c, st, n_contigs, ltot, windows, span, span, burn_blocks, 4, seg, "k_sweep_uniform_gen",
            [&](const uint32_t* table, uint32_t* run_in_out, const uint32_t* redo_in) {
                return qmcp::launch_sweep_uniform_gen(st, boff, poff, n_contigs, span, M, ltot, selend, d_iters, table, n_seg_max,
                                                      run_in_out, redo_in);
            },
            [&](const uint32_t* table, uint32_t* mismatches, const uint32_t* redo_in, uint32_t* redo_out) {
                qmcp::launch_spec_verify(st, table, n_seg_max, span, selend, (const uint32_t*)c->cstart.p, mismatches,
                                         redo_in, redo_out);
            });
    }
    if (qmcp::sweep_uniform_mw_supported(span)) {
        KernelSpan sp(c, gen ? "k_sweep_uniform_gen" : "k_sweep_uniform_mw", st);
        const bool ok = gen ? qmcp::launch_sweep_uniform_gen(st, boff, poff, n_contigs, span, M, ltot, selend, d_iters, seg, n_seg_max)
                            : qmcp::launch_sweep_uniform_mw(st, boff, poff, n_contigs, span, M, ltot, selend, d_iters, seg, n_seg_max);
        if (ok) return QMCP_OK;
    }
    KernelSpan sp(c, "k_sweep_uniform", st);
    if (!qmcp::launch_sweep_uniform(st, boff, poff, n_contigs, span, M, ltot, selend, d_iters, seg, n_seg_max))
        return fail(QMCP_ERANGE, "uniform span %u not supported", span);
    return QMCP_OK;
}

// Near-uniform route: sizes.  Exceptions beyond a tenth of the reads are not worth the route (every one the sweep
// wants costs a sweep of its own); the list holds an eighth of every wave's reads.
constexpr uint32_t kNuSuspects = 1u << 16;
// Rounds the route may take before it gives way to the mixed-span walk: that walk is one serial chain per contig at
// ~0.07 us per position (79.8 ms for cfg4's 10^6-position contigs), a round is ~0.1 ms + what it sweeps again (at most
// a contig: 0.5 ms per 10^6 positions); the route may spend up to about half of what the walk would take.
constexpr uint32_t kNuMinRounds = 8, kNuMaxRoundsCap = 160;
uint32_t nu_round_budget(const uint32_t* lengths, uint32_t n_contigs) {
    if (const char* e = std::getenv("QMCP_HIP_NEAR_ROUNDS")) return (uint32_t)std::strtoul(e, nullptr, 10);
    uint32_t longest = 0;
    for (uint32_t k = 0; k < n_contigs; ++k) longest = lengths[k] > longest ? lengths[k] : longest;
    const double walk_ms = 0.07e-3 * (double)longest;
    const double round_ms = 0.1 + 0.5e-6 * (double)longest;
    const double r = 0.5 * walk_ms / round_ms;
    return r < kNuMinRounds ? kNuMinRounds : r > kNuMaxRoundsCap ? kNuMaxRoundsCap : (uint32_t)r;
}
// mean coverage / M below which the route is not tried: the shallower the data, the more exceptions are wanted and the
// longer the runs of used-up buckets (cfg4's reads with 1 % clipped, lab/near_uniform_depths.py, near-uniform /
// mixed-span ms: 12.5 x M 2.9 / 105; 6.3 x M 6.0 / 401; 4.7 x M 5.5 / 503; 3.75 x M 6.4 / 659; with 40 % of the reads:
// 5 x M 12.4 / 579; 3 x M 15.7 / 710; 2.1 x M: gives up after four sweeps, 648 / 627 -- runs of used-up buckets with
// neither an anchor nor a cut point --; 1.5 x M 41.9 / 600, from cut points).  Below 1.3 x M nearly every window of the
// mixed-span sweep has a real cut point and that sweep is quick.
constexpr double kNuMinDepth = 1.3;
uint32_t nu_cap_for(uint32_t n) {  // 128 slots per wave and pass (or tile): an eighth of the reads, on either producer
    const uint32_t a = qmcp::pm_exc_slots(n), b = qmcp::prepare_exc_slots(n);
    return a > b ? a : b;
}
int ensure_near_uniform(qmcp_hip_ctx* c, uint32_t n, uint32_t ltot, uint32_t n_contigs) {
    TRY(ensure(c, c->nu_exc, qmcp::nu_exc_bytes(nu_cap_for(n))));
    TRY(ensure(c, c->nu_nadj, ((size_t)ltot + 2) * sizeof(int32_t)));
    TRY(ensure(c, c->nu_ce, ((size_t)ltot + 4) * sizeof(uint32_t)));
    TRY(ensure(c, c->nu_state, 64 + (size_t)n_contigs * 20 + 16));
    TRY(ensure(c, c->nu_sus, (size_t)kNuSuspects * 8));
    if (c->nu_ell != 0) {
        // (the route's sweep scratch depends on the span: known from the last call that took the route, so a second call
        //  of the shape grows nothing after its first launch)
        TRY(ensure(c, c->evpk, qmcp::sweep_ev_pack_bytes(ltot, c->nu_ell, n_contigs + 768)));
        TRY(ensure(c, c->evlast, qmcp::sweep_ev_last_bytes(ltot, c->nu_ell, n_contigs + 768)));
        TRY(ensure(c, c->nu_ckpt, qmcp::sweep_ev_ckpt_bytes(ltot, c->nu_ell, n_contigs + 768)));
    }
    TRY(ensure(c, c->spine, (size_t)(qmcp::scan_spine_entries(ltot + 2) + 1) * sizeof(uint32_t) + 16));
    if (!c->h_nu) HIP_TRY(hipHostMalloc((void**)&c->h_nu, 8 * sizeof(uint32_t), hipHostMallocDefault));
    return QMCP_OK;
}

int queue_rm_head(qmcp_hip_ctx* c, hipStream_t s1, uint32_t filter, bool clear_mask);
int queue_pm_head(qmcp_hip_ctx* c, hipStream_t st, uint32_t filter);

// The pass-major form of the range-ranked route (kernels/pass_major.inc.hip) pads every (range, pass) slice to whole
// groups of 64 slots: it pays where slices are long -- a pass's 8 192 reads over the ranges its contig spans --, and
// where they would be short (narrow ranges: small genomes) the padding is most of a group and the range-major form is
// kept.  Hard limits: one partition level, slots addressable with 32-bit byte offsets.
bool pm_route_ok(const uint64_t* roff, const Problem& pr, uint32_t shift) {
    const char* e = std::getenv("QMCP_HIP_PM");
    if (e && e[0] == '0') return false;  // (A/B: the range-major form)
    const uint32_t n = (uint32_t)pr.n, ltot = (uint32_t)pr.ltot;
    if ((uint64_t)qmcp::pm_slots(n, ltot, shift) >= (1ull << 31)) return false;
    if (e && e[0] == '1') return true;   // (tests: the form on small inputs)
    // expected wave-slots against the records' own 1 / 64: a contig's pass deals its reads to the ranges the contig spans
    double slots = 0.0;
    for (uint32_t k = 0; k < pr.n_contigs; ++k) {
        const uint64_t reads = roff[k + 1] - roff[k];
        if (reads == 0 || pr.poff[k + 1] == pr.poff[k]) continue;
        const double ranges = (double)(((pr.poff[k + 1] - 1) >> shift) - (pr.poff[k] >> shift) + 1);
        const double passes = (double)reads / (double)qmcp::pm_pass() < 1.0 ? 1.0 : (double)reads / (double)qmcp::pm_pass();
        const double slice = (double)reads / (passes * ranges);
        slots += passes * ranges * std::ceil(slice / 64.0);
    }
    return slots <= 1.3 * ((double)n / 64.0);
}
// ---------------------------------------------------------------------------------------------------
// One solve = enqueue_head (everything that depends only on the reads' start positions: prepare, the
// range partition and the bucket offsets; nothing in it waits for the device on large calls) +
// enqueue_tail (waits for the 16-byte read-back that picks the route, then queues the sweep and the
// keep mask) + solve_complete (collects).  SolveRun is what the two enqueue halves share.
int enqueue_head(qmcp_hip_ctx* c, const uint32_t* d_starts, const uint32_t* d_ends,
                 const uint64_t* roff, const uint32_t* lengths, uint32_t n_contigs, uint64_t n64,
                 uint32_t M, uint64_t* d_mask) {
    if (c->pending) return fail(QMCP_EINVAL, "a solve is already pending on this context (call qmcp_hip_solve_end)");
    c->pend_spiky = false;
    if (!c->h_scalars) HIP_TRY(hipHostMalloc((void**)&c->h_scalars, 16 * sizeof(unsigned long long), hipHostMallocDefault));
    SolveRun& run = c->run;
    run = SolveRun();
    run.d_starts = d_starts; run.d_ends = d_ends; run.roff = roff; run.lengths = lengths;
    run.n_contigs = n_contigs; run.n64 = n64; run.M = M; run.d_mask = d_mask;
    Problem& pr = run.pr;
    TRY(check_problem(roff, lengths, n_contigs, n64, pr));
    const uint32_t n = (uint32_t)pr.n;
    const uint32_t ltot = (uint32_t)pr.ltot;
    const size_t mask_words = (size_t)((n64 + 63) / 64);
    qmcp_hip_stats& local = run.local;
    std::memset(&local, 0, sizeof(local));
    local.n_reads = n64;
    local.n_contigs = n_contigs;
    local.total_length = pr.ltot;
    if (n == 0 || ltot == 0) {
        if (mask_words) HIP_TRY(hipMemsetAsync(d_mask, 0, mask_words * sizeof(uint64_t), c->stream));
        if (n != 0) return fail(QMCP_EREAD, "reads given for zero-length contigs");
        run.trivial = true;
        HIP_TRY(hipEventRecord(c->ev[EV_BEGIN], c->stream));
        return QMCP_OK;
    }
    // size the whole arena before anything is enqueued (growing a buffer frees it, and
    // hipFree would stall on the work in flight)
    c->sized = false;
    {
        const uint32_t tiles_seg = qmcp::seg_tile_bound(n);  // second partition level: tiles aligned to super-ranges
        const uint32_t spine_a = qmcp::scan_spine_entries(256u * tiles_seg);
        const uint32_t spine_b = qmcp::scan_spine_entries(ltot + 1) + 1;
        TRY(ensure(c, c->spine, (size_t)(spine_a > spine_b ? spine_a : spine_b) * sizeof(uint32_t) + 16));
        TRY(ensure(c, c->hist, (size_t)256 * tiles_seg * sizeof(uint32_t)));
        // (the pass-major form's two 16-bit record streams live in keys[0] and keys[1]: padded slices, ~6 B per read)
        const bool may_pm = n >= rank_min_reads() && qmcp::range_path_supported(ltot) && !qmcp::range_path_two_level(ltot) &&
                            pm_route_ok(roff, pr, qmcp::range_shift_for(ltot));
        const size_t pm_bytes = may_pm ? qmcp::pm_slots(n, ltot, qmcp::range_shift_for(ltot)) * sizeof(uint16_t) : 0;
        TRY(ensure(c, c->keys[0], std::max((size_t)n * sizeof(uint64_t), pm_bytes)));
        TRY(ensure(c, c->keys[1], std::max((size_t)n * sizeof(uint64_t), pm_bytes)));
        if (may_pm) {
            const size_t groups = pm_bytes / (64 * sizeof(uint16_t));
            TRY(ensure(c, c->pm_desc, groups * sizeof(uint32_t)));
            TRY(ensure(c, c->pm_work, 1024 * sizeof(uint32_t)));
        }
        TRY(ensure(c, c->vals[0], (size_t)n * sizeof(uint32_t)));
        TRY(ensure(c, c->vals[1], (size_t)n * sizeof(uint32_t)));
        TRY(ensure(c, c->spine2, (size_t)(spine_a > spine_b ? spine_a : spine_b) * sizeof(uint32_t) + 16));
        TRY(ensure(c, c->hist2, ((size_t)256 * qmcp::part_pass_pitch(n) + 4) * sizeof(uint32_t)));  // (+ the scan's total)
        TRY(ensure(c, c->cstart, ((size_t)ltot + 8) * sizeof(uint32_t)));  // also the event sweep's changed-block S
        TRY(ensure(c, c->boff, ((size_t)ltot + 1) * sizeof(uint32_t)));
        TRY(ensure(c, c->selend, ((size_t)ltot + 8) * sizeof(uint32_t)));  // + spare words for idle lanes
        TRY(ensure(c, c->scalars, 64));
        TRY(ensure(c, c->segs, qmcp::sweep_segment_words(n_contigs < 256 ? n_contigs : 0, qmcp::kMaxSweepWindows) * sizeof(uint32_t)));
        TRY(ensure(c, c->specflags, 2 * 4096 * sizeof(uint32_t)));           // speculative sweeps: marks per exact stretch, two tiers
        TRY(ensure(c, c->specsnap, qmcp::spec_snap_bytes(4096)));            // ... and the mixed-span walk's states at boundaries
        TRY(ensure(c, c->ranges, (65537 + 7 + 771 + 5) * sizeof(uint32_t)));  // range starts, heaviest load, level-2 tables
        if (n >= rank_min_reads() && qmcp::range_path_supported(ltot))
            TRY(ensure(c, c->rankamb, qmcp::rank_scratch_bytes(qmcp::range_shift_for(ltot), ltot, n)));
        TRY(ensure(c, c->stats, 8 * sizeof(uint32_t)));
        // (the near-uniform route's buffers: a context that has met mixed spans may look at the route on any call)
        if (c->nu_ell != 0 || c->mixed_seen) TRY(ensure_near_uniform(c, n, ltot, n_contigs));
        // The mixed-span route's own arrays.  Which route a call takes is known only after its first kernel,
        // so a context that has taken the mixed route once sizes them for every later call up front: growing
        // them after the partition has been queued would stall on it (ensure() waits for the streams).
        if (c->mixed_seen || !(n >= rank_min_reads() && qmcp::range_path_supported(ltot))) {
            TRY(ensure(c, c->ecnt, ((size_t)ltot + 1) * sizeof(uint32_t)));
            TRY(ensure(c, c->eoff, ((size_t)ltot + 1) * sizeof(uint32_t)));
            TRY(ensure(c, c->next_head, ((size_t)n + 2) * sizeof(uint32_t)));
            TRY(ensure(c, c->spine, (size_t)(qmcp::scan_spine_entries(n + 1) + 1) * sizeof(uint32_t) + 16));
        }
    }
    c->grew_mid_solve = 0;
    HIP_TRY(hipEventRecord(c->ev[EV_BEGIN], c->stream));
    TRY(upload_tables(c, roff, pr));
    c->sized = true;

    uint32_t* const hs = c->h_head;  // pinned: [0..2] span min / max / error flag, [3] heaviest range, [4] empty positions
    // the range partition's per-tile histogram is produced by the same pass when the range-ranked
    // path can be taken (uniformity is only known afterwards; the table is cheap)
    run.range_shift = qmcp::range_shift_for(ltot);
    run.may_rank = n >= rank_min_reads() && qmcp::range_path_supported(ltot);
    const uint32_t range_shift = run.range_shift;
    uint32_t* d_range_start = (uint32_t*)c->ranges.p;
    uint32_t* d_max_load = d_range_start + 65540;
    uint32_t* d_seg_tables = d_range_start + 65544;  // super_start, tile_base, pass_base (257 each)
    run.two_level = qmcp::range_path_two_level(ltot);
    const bool two_level = run.two_level;
    hs[3] = 0;
    hs[4] = 0xFFFFFFFFu;  // unknown unless the range-ranked route counted them
    run.have_gstart = true;
    if (!run.may_rank) {
        HIP_TRY(hipMemsetAsync(d_mask, 0, mask_words * sizeof(uint64_t), c->stream));
        uint32_t hs3[3];
        TRY(run_prepare(c, d_starts, d_ends, pr, nullptr, true, false, false, range_shift, nullptr, hs3));
        hs[0] = hs3[0]; hs[1] = hs3[1]; hs[2] = hs3[2];
        HIP_TRY(hipEventRecord(c->ev[EV_PREP], c->stream));
    } else {
        // Large call that can take the range-ranked route if its spans turn out uniform.  The host
        // needs the span statistics before it can pick the sweep, but the device need not idle for
        // that round trip: the partition and the bucket offsets depend only on the start positions,
        // so they are queued behind k_prepare at once and the statistics (and the heaviest range's
        // load) are fetched on the side stream meanwhile.  k_prepare does not write the global
        // start positions on this route -- the partition rebuilds them from the starts.
        run.have_gstart = false;
        hipStream_t s1 = c->stream;
        static const uint32_t init[8] = {0xFFFFFFFFu, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
        HIP_TRY(hipMemcpyAsync(c->stats.p, init, sizeof(init), hipMemcpyHostToDevice, s1));
        run.pm = !two_level && pm_route_ok(roff, pr, range_shift);
        run.nu_filter = c->nu_ell;
        hs[5] = hs[6] = 0;
        if (run.pm) {
            // One pass over the reads: validate, statistics, mask clear, and every pass of 8 192 reads sorted by range
            // (4 B per read out, two [range][pass] tables); a scan of the padded count table gives the padded flat
            // coordinates, one more small kernel the wave-slot descriptors the per-range kernels follow.  No range-major
            // copy, no second read of the starts.
            TRY(queue_pm_head(c, s1, run.nu_filter));
        } else {
            // the range-major form: k_prepare, scan, partition (one or two levels), bucket offsets
            TRY(queue_rm_head(c, s1, run.nu_filter, true));
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(c->ev_head, s1));
        HIP_TRY(hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
        HIP_TRY(hipMemcpyAsync(hs, c->stats.p, 3 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream2));
        HIP_TRY(hipMemcpyAsync(hs + 3, d_max_load, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream2));
        if (run.nu_filter) HIP_TRY(hipMemcpyAsync(hs + 5, (uint32_t*)c->stats.p + 4, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream2));
        // How spiky the starts are only decides WHICH exact sweep kernel runs, so the count of the previous
        // call of this shape is good enough (and saves waiting for k_range_offsets); a first call waits.
        if (c->spiky_known && c->spiky_n == n64 && c->spiky_ltot == pr.ltot) {
            hs[4] = c->spiky_empty;
        } else {
            HIP_TRY(hipMemcpyAsync(hs + 4, (uint32_t*)c->stats.p + 3, sizeof(uint32_t), hipMemcpyDeviceToHost, s1));
            run.wait_empty = true;
        }
        run.ranked_counted = true;
    }
    run.head_done = true;
    return QMCP_OK;
}

// The range-major head's stages on `st`: k_prepare (span statistics, partition table; regular reads: span == filter, or
// every read when filter == 0), scan, the partition (one level, or two for genomes beyond 8.39 M positions), bucket
// offsets.  Used by enqueue_head and, for a call whose head ran with the wrong idea of the spans, again by the tail.
int queue_rm_head(qmcp_hip_ctx* c, hipStream_t s1, uint32_t filter, bool clear_mask) {
    SolveRun& run = c->run;
    const uint32_t n = (uint32_t)run.pr.n, ltot = (uint32_t)run.pr.ltot, n_contigs = run.n_contigs;
    const uint32_t range_shift = run.range_shift;
    const bool two_level = run.two_level;
    uint32_t* d_range_start = (uint32_t*)c->ranges.p;
    uint32_t* d_max_load = d_range_start + 65540;
    uint32_t* d_seg_tables = d_range_start + 65544;  // super_start, tile_base, pass_base (257 each)
    // (a re-run of the head -- after a span change, or with the filter switched on -- must not add to what the first
    //  run counted: empty positions, exceptions, list flag, overflow entries)
    static const uint32_t zeros[4] = {0u, 0u, 0u, 0u};
    HIP_TRY(hipMemcpyAsync((uint32_t*)c->stats.p + 3, zeros, sizeof(zeros), hipMemcpyHostToDevice, s1));
    uint32_t* exc = filter ? (uint32_t*)c->nu_exc.p : nullptr;
    const uint32_t cap = nu_cap_for(n);
    uint32_t* exc_cnt = filter ? qmcp::nu_exc_counts(exc, cap) : nullptr;
    if (filter) HIP_TRY(hipMemsetAsync(exc_cnt, 0, ((size_t)cap / 128 + 4) * sizeof(uint32_t), s1));
    {
        KernelSpan sp(c, "k_prepare");
        qmcp::launch_prepare(s1, run.d_starts, run.d_ends, n, (const uint64_t*)c->roff.p,
                             (const uint64_t*)c->poff.p, n_contigs, nullptr, nullptr, nullptr,
                             (uint32_t*)c->stats.p, two_level ? range_shift + 8 : range_shift,
                             (uint32_t*)c->hist2.p, nullptr, nullptr,
                             clear_mask ? (unsigned long long*)run.d_mask : nullptr,  // also clears the keep mask
                             filter, exc, cap, exc_cnt);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->ev[EV_PREP], s1));
    {
        KernelSpan sp(c, "scan_radix_hist(3 kernels)");
        qmcp::launch_exclusive_scan(s1, (const uint32_t*)c->hist2.p, 256u * qmcp::part_pass_pitch(n),
                                    (uint32_t*)c->hist2.p, (uint32_t*)c->spine2.p, true);
    }
    if (!two_level) {
        KernelSpan sp(c, "k_range_partition");
        qmcp::launch_range_partition(s1, nullptr, run.d_starts, (const uint64_t*)c->roff.p,
                                     (const uint64_t*)c->poff.p, n_contigs, n, range_shift,
                                     (const uint32_t*)c->hist2.p, (uint16_t*)c->keys[0].p,
                                     (uint32_t*)c->vals[0].p, d_range_start, d_max_load, run.d_ends, filter);
    } else {
        // more than 256 ranges (genomes beyond 8.39 M positions): first into <= 256 super-ranges as
        // {global start, index} records, then every super-range into its final ranges
        {
            KernelSpan sp(c, "k_range_partition(level 1)");
            qmcp::launch_partition_level1(s1, run.d_starts, (const uint64_t*)c->roff.p,
                                          (const uint64_t*)c->poff.p, n_contigs, n, range_shift + 8,
                                          (const uint32_t*)c->hist2.p, c->keys[1].p, d_seg_tables, d_max_load,
                                          run.d_ends, filter);
        }
        KernelSpan sp(c, "partition level 2 (tables, hist, scan, scatter)");
        qmcp::launch_partition_level2(s1, c->keys[1].p, n, range_shift, d_seg_tables, (uint32_t*)c->hist.p,
                                      (uint32_t*)c->spine.p, (uint16_t*)c->keys[0].p,
                                      (uint32_t*)c->vals[0].p, d_range_start, d_max_load);
    }
    HIP_TRY(hipEventRecord(c->ev_fork, s1));  // statistics and heaviest load are final here
    {
        // per-range LDS histogram scanned in place: bucket offsets without a genome-wide scan; it also
        // counts the positions that start no read (stats word 3: the host picks the sweep kernel by it)
        KernelSpan sp(c, "k_range_offsets");
        qmcp::launch_range_offsets(s1, (const uint16_t*)c->keys[0].p, d_range_start, range_shift, ltot,
                                   (uint32_t*)c->boff.p, (uint32_t*)c->stats.p + 3);
    }
    HIP_TRY(hipGetLastError());
    run.nu_filter = filter;
    return QMCP_OK;
}

// The pass-major head's stages once more on `st` -- producer (regular reads: span == filter, or every read when
// filter == 0), scan, range table, bucket offsets -- for a call whose head ran with the wrong idea of the spans.
int queue_pm_head(qmcp_hip_ctx* c, hipStream_t st, uint32_t filter) {
    SolveRun& run = c->run;
    const uint32_t n = (uint32_t)run.pr.n, ltot = (uint32_t)run.pr.ltot, n_contigs = run.n_contigs;
    uint32_t* d_stats = (uint32_t*)c->stats.p;
    static const uint32_t zeros[4] = {0u, 0u, 0u, 0u};
    HIP_TRY(hipMemcpyAsync(d_stats + 3, zeros, sizeof(zeros), hipMemcpyHostToDevice, st));  // empty positions, exceptions, list flag, overflow entries
    uint32_t* d_range_start = (uint32_t*)c->ranges.p;
    uint32_t* d_max_load = d_range_start + 65540;
    {
        KernelSpan sp(c, "k_pm_prepare_sort", st);
        qmcp::launch_pm_prepare_sort(st, run.d_starts, run.d_ends, n, (const uint64_t*)c->roff.p, (const uint64_t*)c->poff.p,
                                     n_contigs, run.range_shift, ltot, (uint16_t*)c->keys[0].p, (uint16_t*)c->keys[1].p,
                                     (uint32_t*)c->hist2.p, (uint32_t*)c->hist.p,
                                     (uint32_t*)c->pm_work.p, d_stats, (unsigned long long*)run.d_mask, filter,
                                     filter ? (uint32_t*)c->nu_exc.p : nullptr, nu_cap_for(n),
                                     filter ? qmcp::nu_exc_counts((uint32_t*)c->nu_exc.p, nu_cap_for(n)) : nullptr);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->ev[EV_PREP], st));
    {
        KernelSpan sp(c, "scan_radix_hist(3 kernels)", st);
        qmcp::launch_exclusive_scan(st, (const uint32_t*)c->hist2.p, 256u * qmcp::pm_pitch(n), (uint32_t*)c->hist2.p,
                                    (uint32_t*)c->spine2.p, true);
    }
    {
        KernelSpan sp(c, "k_pm_descr + k_pm_range_table", st);
        qmcp::launch_pm_descr(st, (const uint32_t*)c->hist2.p, (const uint32_t*)c->hist.p, n, ltot, run.range_shift,
                              (uint32_t*)c->pm_desc.p, (uint32_t*)c->pm_work.p, d_range_start, d_max_load);
    }
    HIP_TRY(hipEventRecord(c->ev_fork, st));  // statistics and heaviest load are final here
    {
        KernelSpan sp(c, "k_pm_offsets", st);
        qmcp::launch_pm_offsets(st, (const uint16_t*)c->keys[0].p, (const uint32_t*)c->pm_desc.p, (const uint32_t*)c->hist2.p, n,
                                d_range_start, run.range_shift, ltot, (uint32_t*)c->boff.p, d_stats + 3);
    }
    HIP_TRY(hipGetLastError());
    run.nu_filter = filter;
    return QMCP_OK;
}

// The ranking of the pass-major form on `st`: the ordered walk, then the settling of the quota-crossing groups it listed.
void queue_pm_rank(qmcp_hip_ctx* c, hipStream_t st, const uint32_t* ev_sev, const uint32_t* ev_lastns, uint32_t ell) {
    SolveRun& run = c->run;
    const uint32_t n = (uint32_t)run.pr.n, ltot = (uint32_t)run.pr.ltot;
    const bool by_records = qmcp::rank_scratch_by_records(run.range_shift, ltot, n);
    const uint16_t* keys16 = (const uint16_t*)c->keys[0].p;
    const uint16_t* idx16 = (const uint16_t*)c->keys[1].p;
    const uint32_t* desc = (const uint32_t*)c->pm_desc.p;
    const uint32_t* Tp = (const uint32_t*)c->hist2.p;
    const uint32_t* range_start = (const uint32_t*)c->ranges.p;
    uint32_t* amb_count = (uint32_t*)c->pm_work.p + 512;
    unsigned long long* kept_total = (unsigned long long*)c->scalars.p;
    {
        KernelSpan sp(c, "k_pm_walk", st);
        qmcp::launch_pm_walk(st, keys16, idx16, desc, Tp, n, range_start, run.range_shift, ltot, (const uint32_t*)c->boff.p,
                             (const uint32_t*)c->selend.p, (unsigned long long*)run.d_mask, kept_total, c->rankamb.p, by_records,
                             amb_count, ev_sev, ev_lastns, (const uint64_t*)c->poff.p, run.n_contigs, ell);
    }
    KernelSpan sp(c, "k_pm_settle", st);
    qmcp::launch_pm_settle(st, keys16, idx16, desc, Tp, n, range_start, run.range_shift, ltot, c->rankamb.p, by_records,
                           amb_count, (unsigned long long*)run.d_mask, kept_total);
}

// The near-uniform route's half of the tail (kernels/near_uniform.inc.hip).  Called when the call's spans differ.
// done = true: the keep mask is written (sweep, ranking and the selected exceptions); false: the caller takes the
// mixed-span route (nothing has touched the mask).  The head's producer may already have filtered on c->nu_ell
// (run.nu_filter); otherwise the reads are counted first and, if the longest span is the dominant one, the head's
// stages are queued again with the filter on.
int near_uniform_tail(qmcp_hip_ctx* c, uint32_t min_span, uint32_t max_span, uint32_t max_load, uint32_t* d_iters, bool& done) {
    done = false;
    SolveRun& run = c->run;
    const Problem& pr = run.pr;
    qmcp_hip_stats& local = run.local;
    const uint32_t n = (uint32_t)pr.n, ltot = (uint32_t)pr.ltot, n_contigs = run.n_contigs, M = run.M;
    local.near_uniform_giveup = QMCP_NU_GIVEUP_NOT_TRIED;
    if (const char* e = std::getenv("QMCP_HIP_NEAR"))
        if (e[0] == '0') return QMCP_OK;
    const uint32_t ell = max_span;
    const double depth = (double)n * (double)ell / ((double)ltot * (double)(M ? M : 1));
    const bool dbg = std::getenv("QMCP_HIP_NEAR_DEBUG") != nullptr;
    if (dbg) fprintf(stderr, "[near] pm %d may_rank %d ell %u ev %d depth %.2f min_span %u filter %u\n", (int)run.pm,
                     (int)run.may_rank, ell, (int)qmcp::sweep_uniform_ev_supported(ell, M), depth, min_span, run.nu_filter);
    double min_depth = kNuMinDepth;
    if (const char* e = std::getenv("QMCP_HIP_NEAR_MIN_DEPTH")) min_depth = std::strtod(e, nullptr);  // (lab)
    if (!run.may_rank || ell < ev_min_span() || !qmcp::sweep_uniform_ev_supported(ell, M) ||
        depth < min_depth || min_span == 0)
        return QMCP_OK;
    {
        // The route's sweep is one chain per contig (the event-driven form).  On data deeper than 11 x M that is what
        // the one-span route runs too; shallower, the one-span and mixed-span routes split contigs into stretches, and a
        // whole chain per round only pays while contigs are short (cfg4's 10^6 positions at 1.5 x M: 7 ms a sweep).
        uint32_t longest = 0;
        for (uint32_t k = 0; k < n_contigs; ++k) longest = run.lengths[k] > longest ? run.lengths[k] : longest;
        if (depth < kGenDepth && longest > 2000000u) return QMCP_OK;
    }
    if (c->nu_failed_n == run.n64 && c->nu_failed_ltot == pr.ltot && c->nu_failed_ell == ell && c->nu_failed_M == M) {
        local.near_uniform_giveup = QMCP_NU_GIVEUP_REMEMBERED;
        c->nu_ell = 0;
        return QMCP_OK;
    }
    hipStream_t st = c->stream;
    const uint32_t cap = nu_cap_for(n);
    uint32_t n_exc = 0;
    uint32_t* d_stats = (uint32_t*)c->stats.p;
    if (run.nu_filter == ell) {
        n_exc = c->h_head[5];  // (read back beside the statistics)
        if (c->h_head[6] != 0) { c->nu_ell = 0; local.near_uniform_giveup = QMCP_NU_GIVEUP_TOO_MANY; return QMCP_OK; }  // a pass held more exceptions than it can stage
    } else {
        // how many reads have the longest span?  (one pass over the spans; the host waits for the count)
        TRY(ensure_near_uniform(c, n, ltot, n_contigs));
        HIP_TRY(hipMemsetAsync(d_stats + 7, 0, sizeof(uint32_t), st));
        qmcp::launch_nu_count_span(st, run.d_starts, run.d_ends, n, ell, d_stats + 7);
        HIP_TRY(hipMemcpyAsync(c->h_nu, d_stats + 7, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        n_exc = n - c->h_nu[0];
        if (dbg) fprintf(stderr, "[near] reads of span %u: %u of %u, list holds %u\n", ell, c->h_nu[0], n, cap);
        if (n_exc > n / 10u) {
            // (fewer than nine tenths of the reads have the LONGEST span: either many exceptions, or -- nearly all reads
            //  shorter than a few -- the dominant span is not the longest: reads lengthened by a deletion)
            c->nu_ell = 0;
            local.near_uniform_giveup = n_exc > n - n / 10u ? QMCP_NU_GIVEUP_LONGER_READS : QMCP_NU_GIVEUP_TOO_MANY;
            return QMCP_OK;
        }
        // the head again, regular reads only (exceptions listed): producer, scan, range table, bucket offsets
        c->nu_ell = ell;
        if (run.pm) TRY(queue_pm_head(c, st, ell));
        else TRY(queue_rm_head(c, st, ell, true));
        uint32_t* d_max_load = (uint32_t*)c->ranges.p + 65540;
        HIP_TRY(hipMemcpyAsync(c->h_nu, d_max_load, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(c->h_nu + 1, d_stats + 4, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        max_load = c->h_nu[0];
        if (c->h_nu[1] != n_exc || c->h_nu[2] != 0) { c->nu_ell = 0; local.near_uniform_giveup = QMCP_NU_GIVEUP_TOO_MANY; return QMCP_OK; }  // (a pass held more than it can stage)
    }
    local.near_uniform_exceptions = n_exc;
    if (n_exc == 0 || n_exc > n / 10u || (uint64_t)max_load * kRankBalance > (uint64_t)n) {
        if (n_exc > n / 10u) c->nu_ell = 0;
        local.near_uniform_giveup = n_exc > n / 10u ? QMCP_NU_GIVEUP_TOO_MANY : n_exc == 0 ? QMCP_NU_GIVEUP_NOT_TRIED : QMCP_NU_GIVEUP_HEAVY_RANGE;
        return QMCP_OK;
    }
    // scratch of the event-driven sweep (launch_uniform_sweep)
    TRY(ensure(c, c->evpk, qmcp::sweep_ev_pack_bytes(ltot, ell, n_contigs + 768)));
    TRY(ensure(c, c->evlast, qmcp::sweep_ev_last_bytes(ltot, ell, n_contigs + 768)));
    const uint32_t* boff = (const uint32_t*)c->boff.p;
    const uint64_t* poff = (const uint64_t*)c->poff.p;
    uint32_t* selend = (uint32_t*)c->selend.p;
    uint32_t* exc = (uint32_t*)c->nu_exc.p;
    int32_t* nadj = (int32_t*)c->nu_nadj.p;
    uint32_t* state = (uint32_t*)c->nu_state.p;
    unsigned long long* viol_key = (unsigned long long*)((char*)c->nu_state.p + 64);
    uint32_t* viol_idx = (uint32_t*)(viol_key + n_contigs);
    uint32_t* sweep_from[2] = {viol_idx + n_contigs, viol_idx + 2 * (size_t)n_contigs};  // this round's, the next round's
    TRY(ensure(c, c->nu_ckpt, qmcp::sweep_ev_ckpt_bytes(ltot, ell, n_contigs + 768)));
    HIP_TRY(hipMemsetAsync(sweep_from[0], 0, (size_t)n_contigs * sizeof(uint32_t), st));
    HIP_TRY(hipEventRecord(c->ev[EV_SCAN], st));
    HIP_TRY(hipEventRecord(c->ev[EV_SORT], st));
    {
        KernelSpan sp(c, "near-uniform setup (exception coverage, need, pre-selection)");
        qmcp::launch_nu_setup(st, exc, cap, n_exc, d_stats + 6, boff, ltot, ell, M, (uint32_t*)c->nu_ce.p, (uint32_t*)c->spine.p,
                              nadj, state);
    }
    // Rounds are queued two at a time and the host looks at the state words after each pair: a round whose contigs are
    // all settled is eight launches that return at once (the chain sweeps nothing, the verification skips every
    // exception: ~0.1 ms), about what one more host round trip costs; measured at cfg4 with 1 % clipped reads (7 rounds),
    // batches of 1 / 2 / 2 + 4 + 4: 4.62 / 4.5 / 4.60 ms.
    uint32_t rounds = 0;
    bool settled = false;
    const uint32_t budget = nu_round_budget(run.lengths, n_contigs);
    while (rounds < budget && !settled) {
        const uint32_t batch = 2u;
        for (uint32_t r = 0; r < batch; ++r) {
            ++rounds;
            {
                KernelSpan sp(c, "k_sweep_pack", st);
                qmcp::launch_sweep_ev_pack(st, boff, poff, n_contigs, ell, M, ltot, nullptr, 0, (uint32_t*)c->evpk.p, nadj, sweep_from[0]);
            }
            {
                KernelSpan sp(c, "k_sweep_uniform_ev", st);
                qmcp::launch_sweep_ev_chain(st, boff, poff, n_contigs, ell, M, ltot, nullptr, 0, (const uint32_t*)c->evpk.p,
                                            (uint32_t*)c->cstart.p, (uint32_t*)c->evlast.p, d_iters, nadj, (uint32_t*)c->nu_ckpt.p,
                                            sweep_from[0]);
            }
            {
                KernelSpan sp(c, "k_sweep_expand", st);
                qmcp::launch_sweep_ev_expand(st, boff, poff, n_contigs, ell, M, ltot, nullptr, 0, (const uint32_t*)c->cstart.p,
                                             (const uint32_t*)c->evlast.p, selend, sweep_from[0]);
            }
            {
                KernelSpan sp(c, "near-uniform round (verify, replay, select, apply)");
                qmcp::launch_nu_round(st, exc, cap, n_exc, d_stats + 6, rounds == 1, boff, selend, nadj, (const uint32_t*)c->nu_ce.p, poff, n_contigs, ell, M,
                                      (uint2*)c->nu_sus.p, kNuSuspects, state, viol_key, viol_idx, sweep_from[0], sweep_from[1]);
                std::swap(sweep_from[0], sweep_from[1]);
            }
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(c->h_nu, state, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (dbg) {
            uint32_t more[8];
            (void)hipMemcpy(more, state + 8, sizeof(more), hipMemcpyDeviceToHost);
            fprintf(stderr, "[near] open question: s %u u1 %u S(u1-1) %u C(u1-1) %u b %u c0 %u\n", more[0], more[1], more[2], more[3], more[4], more[5]);
            fprintf(stderr, "[near] after %u rounds: last round selected %u, flags %u (read %u), selected in all %u, suspects %u, rounds that selected %u; next sweeps from block",
                    rounds, c->h_nu[1], c->h_nu[2], c->h_nu[5], c->h_nu[3], c->h_nu[4], c->h_nu[6]);
            std::vector<uint32_t> from(n_contigs);
            (void)hipMemcpy(from.data(), sweep_from[0], (size_t)n_contigs * sizeof(uint32_t), hipMemcpyDeviceToHost);
            for (uint32_t k = 0; k < n_contigs && k < 16; ++k) fprintf(stderr, " %d", (int)from[k]);
            fprintf(stderr, "\n");
        }
        if (c->h_nu[2] != 0) break;          // a run the replay does not model, or too many suspects
        settled = c->h_nu[1] == 0;           // the last round wanted no exception: the sweep's counts are the greedy's
    }
    if (settled) rounds = c->h_nu[6] + 1;    // (the rounds that did something, and the one that found nothing left)
    local.near_uniform_rounds = rounds;
    local.near_uniform_selected = c->h_nu[3];
    if (!settled) {
        // (the head must not filter on this span again, and the next call of this shape must not burn the budget again)
        local.near_uniform_giveup = c->h_nu[2] != 0 ? QMCP_NU_GIVEUP_UNMODELLED : QMCP_NU_GIVEUP_BUDGET;
        c->nu_ell = 0;
        c->nu_failed_n = run.n64; c->nu_failed_ltot = pr.ltot; c->nu_failed_ell = ell; c->nu_failed_M = M;
        return QMCP_OK;
    }
    local.near_uniform_giveup = QMCP_NU_GIVEUP_NONE;
    HIP_TRY(hipEventRecord(c->ev[EV_SWEEP], st));
    if (run.pm) {
        queue_pm_rank(c, st, nullptr, nullptr, 0);
    } else {
        KernelSpan sp(c, "k_rank_mark");
        qmcp::launch_rank_mark(st, (const uint16_t*)c->keys[0].p, (const uint32_t*)c->vals[0].p, (const uint32_t*)c->ranges.p,
                               run.range_shift, ltot, boff, selend, (unsigned long long*)run.d_mask,
                               (unsigned long long*)c->scalars.p, c->rankamb.p,
                               qmcp::rank_scratch_by_records(run.range_shift, ltot, n));
    }
    {
        KernelSpan sp(c, "k_nu_mark_selected");
        qmcp::launch_nu_mark_selected(st, exc, cap, n_exc, d_stats + 6, (unsigned long long*)run.d_mask,
                                      (unsigned long long*)c->scalars.p);
    }
    HIP_TRY(hipGetLastError());
    done = true;
    return QMCP_OK;
}

int enqueue_tail(qmcp_hip_ctx* c) {
    SolveRun& run = c->run;
    const Problem& pr = run.pr;
    qmcp_hip_stats& local = run.local;
    const uint32_t n = (uint32_t)pr.n, ltot = (uint32_t)pr.ltot, n_contigs = run.n_contigs, M = run.M;
    const uint64_t n64 = run.n64;
    const uint32_t *d_starts = run.d_starts, *d_ends = run.d_ends;
    uint64_t* const d_mask = run.d_mask;
    const uint64_t* roff = run.roff;
    const uint32_t* lengths = run.lengths;
    if (run.trivial) {
        for (int i = EV_PREP; i <= EV_MARK; ++i) HIP_TRY(hipEventRecord(c->ev[i], c->stream));
        for (int i = 0; i < 8; ++i) c->h_scalars[i] = 0;
        c->pend_stats = local;
        c->pend_whole_contig_chains = 0;
        c->pending = true;
        return QMCP_OK;
    }
    const uint32_t range_shift = run.range_shift;
    const bool may_rank = run.may_rank;
    uint32_t* d_range_start = (uint32_t*)c->ranges.p;
    const uint32_t* hs = c->h_head;
    bool have_gstart = run.have_gstart;
    const bool ranked_counted = run.ranked_counted;
    if (may_rank) {
        HIP_TRY(hipStreamSynchronize(c->stream2));
        if (run.wait_empty) HIP_TRY(hipStreamSynchronize(c->stream));
        if (hs[2] != 0) {
            (void)hipStreamSynchronize(c->stream);  // what was queued stays in bounds; let it drain
            return fail(QMCP_EREAD, "a read has start > end or end >= its contig length");
        }
    }
    const uint32_t max_load = hs[3];
    const uint32_t empty_positions = hs[4];            // (the last solve of this shape's, or this one's: set by the head)
    const uint32_t min_span = hs[0], max_span = hs[1];
    local.min_span = min_span;
    local.max_span = max_span;
    const bool uniform = (min_span == max_span) && max_span <= qmcp::kMaxUniformSpan;
    if (!uniform && max_span > qmcp::kMaxGeneralSpan)
        return fail(QMCP_ERANGE, "mixed-span reads with span %u > %u are not supported by this build",
                    max_span, qmcp::kMaxGeneralSpan);
    local.path = uniform ? QMCP_PATH_UNIFORM : QMCP_PATH_GENERAL;
    if (uniform)
        for (uint32_t k = 0; k < n_contigs; ++k)
            if (roff[k + 1] - roff[k] >= (1ull << 28))
                return fail(QMCP_ERANGE, "contig %u holds %llu reads; the block sweep handles < 2^28 per contig",
                            k, (unsigned long long)(roff[k + 1] - roff[k]));

    // bucketing keys.  gstart (global start position per read) sits in vals[1].
    const uint32_t pos_bits = bit_width(ltot - 1) == 0 ? 1u : bit_width(ltot - 1);
    uint32_t span_bits = 0;
    bool wide = false;
    const uint32_t* d_gstart = (const uint32_t*)c->vals[1].p;
    const uint32_t* d_key32 = d_gstart;  // uniform span: the key is the start position itself
    auto need_gstart = [&]() {
        if (have_gstart) return;
        KernelSpan sp(c, "k_gstart");
        qmcp::launch_gstart(c->stream, d_starts, n, (const uint64_t*)c->roff.p, (const uint64_t*)c->poff.p,
                            n_contigs, (uint32_t*)c->vals[1].p);
        have_gstart = true;
    };
    bool sweep_done = false, ranked = false;
    uint32_t* d_iters = (uint32_t*)((char*)c->scalars.p + 16);
    bool near_done = false;
    uint32_t max_load_now = max_load;
    if (uniform && run.nu_filter != 0 && run.nu_filter != max_span) {
        // the head listed every read as an exception to the last call's span: its stages again, unfiltered
        c->nu_ell = 0;
        if (run.pm) TRY(queue_pm_head(c, c->stream, 0));
        else TRY(queue_rm_head(c, c->stream, 0, true));
        HIP_TRY(hipMemcpyAsync(c->h_nu, (uint32_t*)c->ranges.p + 65540, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        max_load_now = c->h_nu[0];
    }
    if (!uniform) c->mixed_seen = true;
    if (!uniform && max_span <= qmcp::kMaxUniformSpan) {
        HIP_TRY(hipMemsetAsync(c->scalars.p, 0, 64, c->stream));
        TRY(near_uniform_tail(c, min_span, max_span, max_load, d_iters, near_done));
        if (near_done) {
            local.path = QMCP_PATH_NEAR_UNIFORM;
            sweep_done = ranked = true;
        }
    }
    if (!uniform && !near_done) {
        need_gstart();
        span_bits = bit_width(max_span - min_span);
        wide = pos_bits + span_bits > 32;
        c->mixed_seen = true;
        TRY(ensure(c, c->ecnt, ((size_t)ltot + 1) * sizeof(uint32_t)));
        TRY(ensure(c, c->eoff, ((size_t)ltot + 1) * sizeof(uint32_t)));
        HIP_TRY(hipMemsetAsync(c->ecnt.p, 0, ((size_t)ltot + 1) * sizeof(uint32_t), c->stream));
        void* key_dst = wide ? c->keys[0].p : c->vals[0].p;
        {
            KernelSpan sp(c, "k_general_keys");
            qmcp::launch_general_keys(c->stream, wide, d_gstart, d_starts, d_ends, n, span_bits,
                                      max_span, nullptr, key_dst, (uint32_t*)c->ecnt.p, ltot + 1);
        }
        HIP_TRY(hipGetLastError());
        TRY(scan_counts(c, c->ecnt, c->eoff, ltot));
        d_key32 = (const uint32_t*)c->vals[0].p;
    }
    if (!near_done) HIP_TRY(hipEventRecord(c->ev[EV_SCAN], c->stream));
    // Uniform span, large call: neither the sweep nor the keep mask needs a full sort.  One stable
    // partition of {start, index} records by position range, per-range LDS counts, the sweep, and
    // a per-range ordered ranking against S(p) -- see "range-ranked uniform path" in the kernels.
    // The heaviest range's load is read back on the second stream while the partition runs; if
    // one range holds too much (its ranking is one wave's serial walk), the keep mask comes from
    // the radix sort instead (the counts and the sweep done here stay valid).
    bool mixed_whole_contigs = false;
    if (!near_done) HIP_TRY(hipMemsetAsync(c->scalars.p, 0, 64, c->stream));
    if (uniform && may_rank) {
        hipStream_t s1 = c->stream;  // (partition and bucket offsets are already queued)
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(c->ev[EV_SORT], s1));
        ranked = (uint64_t)max_load_now * kRankBalance <= (uint64_t)n;
        if (std::getenv("QMCP_HIP_NO_RANK") != nullptr) ranked = false;  // test hook: force the sort
        // (the pass-major ranking can take its quotas from the event-driven sweep's own output: no expand, no selend[])
        bool expand_left_out = ranked && run.pm && std::getenv("QMCP_HIP_EXPAND") == nullptr;
        TRY(launch_uniform_sweep(c, s1, n, ltot, n_contigs, max_span, M, d_iters, empty_positions, &expand_left_out));
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(c->ev[EV_SWEEP], s1));
        sweep_done = true;
        if (ranked && run.pm) {
            queue_pm_rank(c, s1, expand_left_out ? (const uint32_t*)c->cstart.p : nullptr, (const uint32_t*)c->evlast.p, max_span);
            HIP_TRY(hipGetLastError());
        } else if (ranked) {
            KernelSpan sp(c, "k_rank_mark");
            qmcp::launch_rank_mark(s1, (const uint16_t*)c->keys[0].p, (const uint32_t*)c->vals[0].p,
                                   d_range_start, range_shift, ltot,
                                   (const uint32_t*)c->boff.p, (const uint32_t*)c->selend.p,
                                   (unsigned long long*)d_mask, (unsigned long long*)c->scalars.p,
                                   c->rankamb.p, qmcp::rank_scratch_by_records(range_shift, ltot, n));
            HIP_TRY(hipGetLastError());
        }
    }

    // radix bucketing: stable LSD, 8-bit digits
    const uint32_t key_bits = pos_bits + span_bits;
    const uint32_t passes = (key_bits + 7) / 8;
    local.sort_passes = ranked ? 1u : passes;  // ranked path: one range partition, no sort
    const uint32_t n_tiles = qmcp::sort_tiles(n);
    int kin = 0, vin = 0;  // buffers holding the sorted output at the end
    if (!ranked && uniform) need_gstart();  // the sort-based routes bucket the bare keys
    if (ranked) {
        // keep mask already written by k_rank_mark
    } else if (!wide) {
        // records {key, read index}: keys[0] <-> keys[1]; the first pass reads bare keys
        const void* recs_in = nullptr;
        for (uint32_t p = 0; p < passes; ++p) {
            const bool first = p == 0;
            const int kout = first ? 0 : (kin ^ 1);
            {
                KernelSpan sp(c, "k_radix_hist_rec");
                qmcp::launch_radix_hist_rec(c->stream, first, d_key32, recs_in, n, 8 * p,
                                            (uint32_t*)c->hist.p);
            }
            {
                KernelSpan sp(c, "scan_radix_hist(3 kernels)");
                qmcp::launch_exclusive_scan(c->stream, (const uint32_t*)c->hist.p, 256u * n_tiles,
                                            (uint32_t*)c->hist.p, (uint32_t*)c->spine.p, false);
            }
            {
                KernelSpan sp(c, "k_radix_scatter_rec");
                qmcp::launch_radix_scatter_rec(c->stream, first, d_key32, recs_in, n, 8 * p,
                                               (const uint32_t*)c->hist.p, c->keys[kout].p);
            }
            HIP_TRY(hipGetLastError());
            kin = kout;
            recs_in = c->keys[kin].p;
        }
    } else {
        // 64-bit composite keys (huge genome x wide span range): split key / payload arrays
        const uint32_t* vals_in = nullptr;
        for (uint32_t p = 0; p < passes; ++p) {
            const int kout = kin ^ 1, vout = (vals_in == nullptr) ? 0 : (vin ^ 1);
            {
                KernelSpan sp(c, "k_radix_hist");
                qmcp::launch_radix_hist(c->stream, true, c->keys[kin].p, n, 8 * p, (uint32_t*)c->hist.p);
            }
            {
                KernelSpan sp(c, "scan_radix_hist(3 kernels)");
                qmcp::launch_exclusive_scan(c->stream, (const uint32_t*)c->hist.p, 256u * n_tiles,
                                            (uint32_t*)c->hist.p, (uint32_t*)c->spine.p, false);
            }
            {
                KernelSpan sp(c, "k_radix_scatter");
                qmcp::launch_radix_scatter(c->stream, true, c->keys[kin].p, vals_in, n, 8 * p,
                                           (const uint32_t*)c->hist.p, c->keys[kout].p,
                                           (uint32_t*)c->vals[vout].p);
            }
            HIP_TRY(hipGetLastError());
            kin = kout;
            vin = vout;
            vals_in = (const uint32_t*)c->vals[vin].p;
        }
    }
    // bucket offsets straight from the sorted keys (no atomics)
    if (!sweep_done) {
    HIP_TRY(hipMemsetAsync(c->boff.p, 0xFF, ((size_t)ltot + 1) * sizeof(uint32_t), c->stream));
    {
        KernelSpan sp(c, "k_bucket_heads");
        qmcp::launch_bucket_heads(c->stream, wide, c->keys[kin].p, (const uint32_t*)c->vals[vin].p, n,
                                  span_bits, ltot, (uint32_t*)c->boff.p);
    }
    {
        KernelSpan sp(c, "reverse_min_scan(3 kernels)");
        qmcp::launch_reverse_min_scan(c->stream, (uint32_t*)c->boff.p, ltot + 1, (uint32_t*)c->spine.p);
    }
    }
    HIP_TRY(hipGetLastError());
    if (!sweep_done) HIP_TRY(hipEventRecord(c->ev[EV_SORT], c->stream));

    // selection sweep
    if (sweep_done) {
        // done above, from the early counts
    } else if (uniform) {
        TRY(launch_uniform_sweep(c, c->stream, n, ltot, n_contigs, max_span, M, d_iters, empty_positions));
    } else {
        uint32_t ring = 64;
        while (ring <= max_span) ring <<= 1;
        // shallow or gapped data: stretches between cut points, one wave each (depth judged with the
        // longest span: an upper bound)
        const uint32_t* seg = nullptr;
        uint32_t n_seg_max = 0;
        const double depth = (double)n * (double)max_span / ((double)ltot * (double)(M ? M : 1));
        // (a mixed-span walk is one light workgroup per stretch and slow per position: five times the windows
        //  the one-span sweeps get, whose seven-wave workgroups fill the chip at three per compute unit)
        const uint32_t windows = sweep_cut_windows(ltot, max_span, n_contigs, depth < kGenDepth, qmcp::kMaxSweepWindows);
        if (windows != 0) {
            KernelSpan sp(c, "k_find_cuts");
            seg = qmcp::launch_sweep_segments(c->stream, (const uint32_t*)c->boff.p, (const uint32_t*)c->eoff.p,
                                              (const uint64_t*)c->poff.p, n_contigs, ltot, max_span, M, windows,
                                              (uint32_t*)c->segs.p);
            n_seg_max = n_contigs + windows;
            // stats.sweep_stretches: the table's count (the uniform kernels count themselves)
            HIP_TRY(hipMemcpyAsync(d_iters + 2, seg, sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
        }
        mixed_whole_contigs = seg == nullptr;
        if (max_span <= qmcp::kMaxCachedSpan) {
            ring = 64;
            while (ring < max_span + 64) ring <<= 1;  // 64 buckets enter per chunk
            // run lengths of equal (start, end) groups: heads + reverse min-scan -> next_head[]
            TRY(ensure(c, c->next_head, ((size_t)n + 2) * sizeof(uint32_t)));
            TRY(ensure(c, c->spine, (size_t)(qmcp::scan_spine_entries(n + 1) + 1) * sizeof(uint32_t) + 16));
            {
                KernelSpan sp(c, "k_group_heads");
                qmcp::launch_group_heads(c->stream, wide, c->keys[kin].p, n, (uint32_t*)c->next_head.p);
            }
            {
                KernelSpan sp(c, "reverse_min_scan(3 kernels)");
                qmcp::launch_reverse_min_scan(c->stream, (uint32_t*)c->next_head.p, n + 1,
                                              (uint32_t*)c->spine.p);
            }
            // spans up to 448: the window of live buckets fits the wave's registers (8 per lane)
            const bool in_regs = max_span + 64 <= 512 && std::getenv("QMCP_HIP_GENERAL_LDS") == nullptr;
            // speculative stretch boundaries, as for one span (launch_uniform_sweep): the state is the
            // selected reads still alive, i.e. the kept counts of the last max_span start positions, which
            // k_spec_verify compares (selend = bucket start + kept count); the run-in is counted in
            // windows of max_span positions
            // (the first tier starts lower than for one span: a walk is slow per position, so short stretches
            //  matter more, and the second tier is there)
            const uint32_t burn_blocks = std::getenv("QMCP_HIP_SPEC_BURN") ? spec_first_run_in(depth) : spec_first_run_in(depth) * 3u / 5u;
            bool hopeless = c->spec_hopeless_n == n64 && c->spec_hopeless_ltot == pr.ltot && c->spec_hopeless_M == M;
            if (!hopeless && spec_wanted(depth) && in_regs && seg != nullptr && n >= (1u << 20)) {
                // One dominant read length (what is left for this route once the shorter reads have their own: a few
                // LONGER ones) forgets its state as slowly as one-length data, and the walk's boundaries then disagree
                // nearly everywhere (lab/mixed_spec_check.py: 430 against 185 ms at 7.5 x M); a broad mix of lengths
                // forgets fast and gains (lab/mixed_spec_broad.py: 117 against 271 ms at 5 x M).  A sample of the spans
                // tells the two apart before anything is queued.
                uint32_t* d_share = (uint32_t*)c->stats.p + 6;
                qmcp::launch_span_mode_share(c->stream, d_starts, d_ends, n, d_share);
                uint32_t share[2] = {0, 0};
                HIP_TRY(hipMemcpyAsync(share, d_share, sizeof(share), hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(hipStreamSynchronize(c->stream));
                hopeless = share[0] != 0 && (uint64_t)share[1] * 10u >= (uint64_t)share[0] * 9u;
            }
            if (std::getenv("QMCP_HIP_SPEC") != nullptr || std::getenv("QMCP_HIP_SPEC_BURN") != nullptr) hopeless = false;
            const bool speculate = !hopeless && spec_wanted(depth) && in_regs && seg != nullptr && burn_blocks >= 2 &&
                                   (uint64_t)ltot >= 4ull * burn_blocks * max_span;
            if (speculate) {
                TRY(ensure(c, c->specsnap, qmcp::spec_snap_bytes(n_seg_max)));
                const void* sorted = c->keys[kin].p;
                // (a walk is one light workgroup: many short stretches beat few long ones -- two run-ins apart)
                TRY(speculative_sweep(
                    c, c->stream, n_contigs, ltot, windows, max_span, 64, burn_blocks, 2, seg, "k_sweep_general_reg",
                    [&](const uint32_t* table, uint32_t* out_odd, const uint32_t* redo_in) {
                        return qmcp::launch_sweep_general_reg(c->stream, wide, (const uint32_t*)c->boff.p, (const uint32_t*)c->eoff.p,
                                                              sorted, (const uint32_t*)c->next_head.p, (const uint64_t*)c->poff.p,
                                                              n_contigs, span_bits, max_span, M, (uint32_t*)c->selend.p, table,
                                                              n_seg_max, out_odd, redo_in, (uint32_t*)c->specsnap.p);
                    },
                    [&](const uint32_t* table, uint32_t* mismatches, const uint32_t* redo_in, uint32_t* redo_out) {
                        qmcp::launch_spec_verify_merge_mixed(c->stream, table, n_seg_max, max_span, (uint32_t*)c->selend.p,
                                                             (const uint32_t*)c->cstart.p, (const uint32_t*)c->specsnap.p,
                                                             mismatches, redo_in, redo_out);
                    }));
                // stats.sweep_stretches: the first tier's table
                HIP_TRY(hipMemcpyAsync(d_iters + 2, (uint32_t*)c->segs.p + windows + (1 + 5 * (size_t)n_seg_max), sizeof(uint32_t),
                                       hipMemcpyDeviceToDevice, c->stream));
            }
            if (!speculate) {  // (else: swept above)
                KernelSpan sp(c, in_regs ? "k_sweep_general_reg" : "k_sweep_general_cached");
                if (!in_regs ||
                    !qmcp::launch_sweep_general_reg(c->stream, wide, (const uint32_t*)c->boff.p,
                                                    (const uint32_t*)c->eoff.p, c->keys[kin].p,
                                                    (const uint32_t*)c->next_head.p, (const uint64_t*)c->poff.p,
                                                    n_contigs, span_bits, max_span, M, (uint32_t*)c->selend.p, seg,
                                                    n_seg_max))
                    qmcp::launch_sweep_general_cached(c->stream, wide, (const uint32_t*)c->boff.p,
                                                      (const uint32_t*)c->eoff.p, c->keys[kin].p,
                                                      (const uint32_t*)c->next_head.p, (const uint64_t*)c->poff.p,
                                                      n_contigs, span_bits, max_span, M,
                                                      (uint32_t*)c->selend.p, ring, seg, n_seg_max);
            }
        } else {
            uint32_t* g_rings = nullptr;
            if (max_span > qmcp::kMaxLdsRingSpan) {
                // long reads: the two rings of a workgroup no longer fit LDS
                const size_t n_wg = seg ? n_seg_max : n_contigs;
                TRY(ensure(c, c->rings, n_wg * 2 * (size_t)ring * sizeof(uint32_t)));
                g_rings = (uint32_t*)c->rings.p;
            }
            KernelSpan sp(c, "k_sweep_general");
            qmcp::launch_sweep_general(c->stream, wide, (const uint32_t*)c->boff.p,
                                       (const uint32_t*)c->eoff.p, c->keys[kin].p,
                                       (const uint64_t*)c->poff.p, n_contigs, span_bits, max_span, M,
                                       (uint32_t*)c->selend.p, ring, seg, n_seg_max, g_rings);
        }
    }
    HIP_TRY(hipGetLastError());
    if (!sweep_done) HIP_TRY(hipEventRecord(c->ev[EV_SWEEP], c->stream));

    // keep mask
    if (!ranked) {
        KernelSpan sp(c, "k_mark");
        qmcp::launch_mark(c->stream, wide, c->keys[kin].p, (const uint32_t*)c->vals[vin].p, ltot,
                          (const uint32_t*)c->boff.p, (const uint32_t*)c->selend.p, d_mask,
                          (unsigned long long*)c->scalars.p);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->ev[EV_MARK], c->stream));
    HIP_TRY(hipMemcpyAsync(c->h_scalars, c->scalars.p, 7 * sizeof(unsigned long long), hipMemcpyDeviceToHost,
                           c->stream));
    c->pend_spiky = ranked_counted;
    if (ranked_counted) {
        HIP_TRY(hipMemcpyAsync(c->h_scalars + 7, (uint32_t*)c->stats.p + 3, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        c->spiky_n = n64;
        c->spiky_ltot = pr.ltot;
    }
    c->pend_stats = local;
    c->pend_M = M;
    c->pend_whole_contig_chains = 0;
    if (mixed_whole_contigs)  // one wave per non-empty contig
        for (uint32_t k = 0; k < n_contigs; ++k) c->pend_whole_contig_chains += lengths[k] != 0 ? 1u : 0u;
    c->pending = true;
    return QMCP_OK;
}

int collect_one(qmcp_hip_ctx* c, qmcp_hip_stats* st) {
    if (!c->pending) return fail(QMCP_EINVAL, "no solve is pending on this context");
    c->pending = false;
    HIP_TRY(hipStreamSynchronize(c->stream));
    collect_spans(c);
#ifdef QMCP_EV_STAMP
    if (c->scalars.p) {
        uint32_t raw[16];
        HIP_TRY(hipMemcpy(raw, c->scalars.p, sizeof(raw), hipMemcpyDeviceToHost));
        const uint32_t* it = raw + 4;
        fprintf(stderr, "[ev stamp] changed %u of %u blocks, stretches %u | x16 cycles: total %u general %u wait-for-ring %u slow-pieces %u (%u pieces)\n",
                it[0], it[1], it[2], it[4], it[5], it[6], it[7], it[8]);
    }
#endif
    qmcp_hip_stats local = c->pend_stats;
    const unsigned long long* host_scalars = c->h_scalars;
    if (c->pend_spiky) {
        c->spiky_empty = (uint32_t)(c->h_scalars[7] & 0xFFFFFFFFull);
        c->spiky_known = true;
    }
    local.n_kept = host_scalars[0];
    c->last_iters = (uint32_t)(host_scalars[2] & 0xFFFFFFFFu);
    c->last_blocks = (uint32_t)(host_scalars[2] >> 32);
    local.sweep_blocks_changed = c->last_iters;
    local.sweep_blocks = c->last_blocks;
    local.sweep_stretches = (uint32_t)(host_scalars[3] & 0xFFFFFFFFu) + c->pend_whole_contig_chains;
    local.spec_mismatches = (uint32_t)(host_scalars[4] & 0xFFFFFFFFu);
    local.spec_boundaries = (uint32_t)(host_scalars[4] >> 32);
    local.spec_retry_mismatches = local.spec_mismatches ? (uint32_t)(host_scalars[5] & 0xFFFFFFFFu) : 0u;
    if (local.path == QMCP_PATH_GENERAL && local.spec_boundaries >= 4 && 2u * local.spec_mismatches > local.spec_boundaries &&
        std::getenv("QMCP_HIP_SPEC_BURN") == nullptr) {
        // (three sweeps -- both tiers and the exact one -- where one would have done: 430 against 185 ms on cfg4's reads
        //  with 1 % clipped at 7.5 x M, lab/mixed_spec_check.py)
        c->spec_hopeless_n = local.n_reads; c->spec_hopeless_ltot = local.total_length; c->spec_hopeless_M = c->pend_M;
    }
    local.ms_prepare = elapsed(c->ev[EV_BEGIN], c->ev[EV_PREP]);
    local.ms_scan = elapsed(c->ev[EV_PREP], c->ev[EV_SCAN]);
    local.ms_sort = elapsed(c->ev[EV_SCAN], c->ev[EV_SORT]);
    local.ms_sweep = elapsed(c->ev[EV_SORT], c->ev[EV_SWEEP]);
    local.ms_mark = elapsed(c->ev[EV_SWEEP], c->ev[EV_MARK]);
    local.ms_total = elapsed(c->ev[EV_BEGIN], c->ev[EV_MARK]);
    local.arena_grown_mid_solve = c->grew_mid_solve;
    if (st) *st = local;
    return QMCP_OK;
}

int solve_complete(qmcp_hip_ctx* c, qmcp_hip_stats* st) { return collect_one(c, st); }

int create_ctx(int device, qmcp_hip_ctx** out_ctx) {
    if (!out_ctx) return fail(QMCP_EINVAL, "out_ctx is null");
    *out_ctx = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(QMCP_ENODEVICE, "no HIP device available (quasi-mcp-hip has no CPU fallback)");
    }
    if (device < 0 || device >= n) return fail(QMCP_ENODEVICE, "device %d out of range [0,%d)", device, n);
    HIP_TRY(hipSetDevice(device));
    qmcp_hip_ctx* c = new (std::nothrow) qmcp_hip_ctx();
    if (!c) return fail(QMCP_ENOMEM, "host allocation failed");
    c->device = device;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    for (int i = 0; e == hipSuccess && i < EV_COUNT; ++i) e = hipEventCreate(&c->ev[i]);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming);
    if (e == hipSuccess) {
        int lo = 0, hi = 0;  // numerically lower == higher priority
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        e = hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, hi);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_head, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipHostMalloc((void**)&c->h_head, 8 * sizeof(uint32_t), hipHostMallocDefault);
    if (e != hipSuccess) {
        qmcp_hip_destroy(c);
        return fail(QMCP_EHIP, "context setup: %s", hipGetErrorString(e));
    }
    *out_ctx = c;
    return QMCP_OK;
}


// Everything of a solve up to and including its last launch; nothing here waits for the device
// except the small read-back that picks the route (span statistics, heaviest range), and that
// wait leaves the device free to work on whatever else is queued.  solve_complete collects.
int solve_enqueue(qmcp_hip_ctx* c, const uint32_t* d_starts, const uint32_t* d_ends,
                  const uint64_t* roff, const uint32_t* lengths, uint32_t n_contigs, uint64_t n64,
                  uint32_t M, uint64_t* d_mask) {
    if (c->pending) return fail(QMCP_EINVAL, "a solve is already pending on this context (call qmcp_hip_solve_end)");
    TRY(enqueue_head(c, d_starts, d_ends, roff, lengths, n_contigs, n64, M, d_mask));
    return enqueue_tail(c);
}

int solve_on_device(qmcp_hip_ctx* c, const uint32_t* d_starts, const uint32_t* d_ends,
                    const uint64_t* roff, const uint32_t* lengths, uint32_t n_contigs, uint64_t n64,
                    uint32_t M, uint64_t* d_mask, qmcp_hip_stats* st) {
    TRY(solve_enqueue(c, d_starts, d_ends, roff, lengths, n_contigs, n64, M, d_mask));
    return solve_complete(c, st);
}

int use_device(qmcp_hip_ctx* c, bool may_be_pending = false) {
    if (!c) return fail(QMCP_EINVAL, "null context");
    if (c->pending && !may_be_pending)
        return fail(QMCP_EINVAL, "a solve is pending on this context (call qmcp_hip_solve_end first)");
    HIP_TRY(hipSetDevice(c->device));
    return QMCP_OK;
}

// Order the solver stream after the caller's producer stream (NULL = the default stream:
// the solver stream is non-blocking, so even that needs an explicit edge).
int order_after(qmcp_hip_ctx* c, void* user_stream) {
    if ((hipStream_t)user_stream != c->stream) {
        HIP_TRY(hipEventRecord(c->ev_in, (hipStream_t)user_stream));
        HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_in, 0));
    }
    return QMCP_OK;
}

int coverage_common(qmcp_hip_ctx* c, const uint32_t* starts, const uint32_t* ends, uint64_t n64,
                    const uint64_t* roff, const uint32_t* lengths, uint32_t n_contigs,
                    const uint64_t* keep_mask, uint32_t* cov_out) {
    TRY(use_device(c));
    if (c->pending) return fail(QMCP_EINVAL, "a solve is pending on this context (call qmcp_hip_solve_end)");
    if (n64 && (!starts || !ends)) return fail(QMCP_EINVAL, "null buffer");
    Problem pr;
    TRY(check_problem(roff, lengths, n_contigs, n64, pr));
    const uint32_t n = (uint32_t)pr.n, ltot = (uint32_t)pr.ltot;
    if (ltot == 0) return n ? fail(QMCP_EREAD, "reads given for zero-length contigs") : QMCP_OK;
    if (n == 0) {
        // (cov_out == null: the caller wants the coverage left in the context's device buffer)
        if (cov_out) std::memset(cov_out, 0, (size_t)ltot * sizeof(uint32_t));
        TRY(ensure(c, c->cov, (size_t)ltot * sizeof(uint32_t)));
        HIP_TRY(hipMemsetAsync(c->cov.p, 0, (size_t)ltot * sizeof(uint32_t), c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return QMCP_OK;
    }
    TRY(ensure(c, c->in_starts, (size_t)n * 4));
    TRY(ensure(c, c->in_ends, (size_t)n * 4));
    HIP_TRY(hipMemcpyAsync(c->in_starts.p, starts, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->in_ends.p, ends, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    const uint64_t* d_keep = nullptr;
    if (keep_mask) {
        const size_t words = (size_t)((n64 + 63) / 64);
        TRY(ensure(c, c->mask, words * 8));
        c->mask_reads = n64;
        HIP_TRY(hipMemcpyAsync(c->mask.p, keep_mask, words * 8, hipMemcpyHostToDevice, c->stream));
        d_keep = (const uint64_t*)c->mask.p;
    }
    TRY(upload_tables(c, roff, pr));
    uint32_t hs[3];
    TRY(run_prepare(c, (const uint32_t*)c->in_starts.p, (const uint32_t*)c->in_ends.p, pr, d_keep,
                    true, true, false, 0, nullptr, hs));
    TRY(scan_counts(c, c->cstart, c->boff, ltot));
    TRY(ensure(c, c->ecnt, ((size_t)ltot + 1) * sizeof(uint32_t)));
    HIP_TRY(hipMemsetAsync(c->ecnt.p, 0, ((size_t)ltot + 1) * sizeof(uint32_t), c->stream));
    qmcp::launch_general_keys(c->stream, false, (const uint32_t*)c->vals[1].p,
                              (const uint32_t*)c->in_starts.p, (const uint32_t*)c->in_ends.p, n, 0,
                              hs[1], d_keep, nullptr, (uint32_t*)c->ecnt.p, ltot + 1);
    HIP_TRY(hipGetLastError());
    TRY(scan_counts(c, c->ecnt, c->eoff, ltot));
    TRY(ensure(c, c->cov, (size_t)ltot * sizeof(uint32_t)));
    qmcp::launch_coverage(c->stream, (const uint32_t*)c->boff.p, (const uint32_t*)c->eoff.p, ltot,
                          (uint32_t*)c->cov.p);
    HIP_TRY(hipGetLastError());
    if (cov_out)
        HIP_TRY(hipMemcpyAsync(cov_out, c->cov.p, (size_t)ltot * sizeof(uint32_t), hipMemcpyDeviceToHost,
                               c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return QMCP_OK;
}

// Host uint32 columns -> c->in_starts / c->in_ends on the solver stream.  Large calls whose first reads
// all have one span: host threads check that every read has it while the starts are copied; if so the
// ends never cross the link -- the device rebuilds them (bit for bit: ends[i] == starts[i] + span mod
// 2^32 is what was checked, so invalid reads stay invalid).  *columns_sent: 1 or 2.
int upload_columns(qmcp_hip_ctx* c, const uint32_t* starts, const uint32_t* ends, uint64_t n_reads,
                   uint32_t* columns_sent) {
    const size_t nb = (size_t)n_reads * sizeof(uint32_t);
    *columns_sent = 2;
    if (nb == 0) return QMCP_OK;
    bool ends_on_device = false;
    const uint32_t span0 = ends[0] - starts[0];
    bool speculate = n_reads >= (1u << 20) && std::getenv("QMCP_HIP_HOST_BOTH_COLUMNS") == nullptr;
    for (size_t i = 0; speculate && i < 4096; ++i) speculate = ends[i] - starts[i] == span0;
    if (speculate) {
        unsigned T = 8;
        if (const char* e = std::getenv("QMCP_HIP_HOST_THREADS")) T = (unsigned)std::strtoul(e, nullptr, 10);
        const unsigned hw = std::thread::hardware_concurrency();
        if (T < 1) T = 1;
        if (hw != 0 && T > hw) T = hw;
        std::atomic<uint32_t> differs{0};
        const size_t n = (size_t)n_reads;
        auto check = [&](unsigned t) {
            // interleaved 64 Ki-read pieces, so that all threads walk the columns front to back together
            constexpr size_t kPiece = 1u << 16;
            uint32_t d = 0;
            for (size_t lo = (size_t)t * kPiece; lo < n && differs.load(std::memory_order_relaxed) == 0; lo += (size_t)T * kPiece) {
                const size_t hi = lo + kPiece < n ? lo + kPiece : n;
                for (size_t i = lo; i < hi; ++i) d |= (ends[i] - starts[i]) ^ span0;
                if (d) differs.fetch_or(d, std::memory_order_relaxed);
            }
        };
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < T; ++t) pool.emplace_back(check, t);
        const bool copied = hipMemcpyAsync(c->in_starts.p, starts, nb, hipMemcpyHostToDevice, c->stream) == hipSuccess;
        for (auto& th : pool) th.join();
        if (!copied) return fail(QMCP_EHIP, "H2D copy failed: %s", hipGetErrorString(hipGetLastError()));
        ends_on_device = differs.load() == 0;
        if (ends_on_device) {
            *columns_sent = 1;
            qmcp::launch_fill_ends(c->stream, (const uint32_t*)c->in_starts.p, (uint32_t)n_reads, span0, (uint32_t*)c->in_ends.p);
        }
    } else if (hipMemcpyAsync(c->in_starts.p, starts, nb, hipMemcpyHostToDevice, c->stream) != hipSuccess) {
        return fail(QMCP_EHIP, "H2D copy failed: %s", hipGetErrorString(hipGetLastError()));
    }
    if (!ends_on_device && hipMemcpyAsync(c->in_ends.p, ends, nb, hipMemcpyHostToDevice, c->stream) != hipSuccess)
        return fail(QMCP_EHIP, "H2D copy failed: %s", hipGetErrorString(hipGetLastError()));
    return QMCP_OK;
}

}  // namespace

extern "C" {

int qmcp_hip_abi_version(void) { return QMCP_HIP_ABI_VERSION; }

const char* qmcp_hip_last_error(void) { return g_err; }

int qmcp_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int qmcp_hip_create(int device, qmcp_hip_ctx** out_ctx) { return create_ctx(device, out_ctx); }

void qmcp_hip_destroy(qmcp_hip_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->stream2) (void)hipStreamSynchronize(c->stream2);
    DevBuf* bufs[] = {&c->roff, &c->poff, &c->stats, &c->cstart, &c->boff, &c->ecnt, &c->eoff,
                      &c->selend, &c->spine, &c->hist, &c->spine2, &c->hist2, &c->keys[0], &c->keys[1], &c->vals[0],
                      &c->vals[1], &c->in_starts, &c->in_ends, &c->in_aux0, &c->in_aux1, &c->mask,
                      &c->cov, &c->amp, &c->scalars, &c->next_head, &c->ranges, &c->rankamb, &c->pm_desc, &c->pm_work, &c->segs, &c->specsnap, &c->specflags, &c->rings, &c->evpk, &c->evlast, &c->kidx, &c->f_starts, &c->f_ends, &c->f_map, &c->f_words, &c->f_mask, &c->nu_exc, &c->nu_nadj, &c->nu_ce, &c->nu_state, &c->nu_sus, &c->nu_ckpt};
    for (DevBuf* b : bufs)
        if (b->p) (void)hipFree(b->p);
    for (int i = 0; i < EV_COUNT; ++i)
        if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    if (c->ev_in) (void)hipEventDestroy(c->ev_in);
    for (auto& sp : c->spans) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
    if (c->h_tables) (void)hipHostFree(c->h_tables);
    if (c->h_scalars) (void)hipHostFree(c->h_scalars);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->h_mask) (void)hipHostFree(c->h_mask);
    for (hipEvent_t e : c->stage_ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->stage_done) (void)hipEventDestroy(e);
    for (hipStream_t st : c->stage_streams) (void)hipStreamDestroy(st);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_head) (void)hipEventDestroy(c->ev_head);
    if (c->ev_done) (void)hipEventDestroy(c->ev_done);
    if (c->h_head) (void)hipHostFree(c->h_head);
    if (c->h_nu) (void)hipHostFree(c->h_nu);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int qmcp_hip_set_profiling(qmcp_hip_ctx* c, int enabled) {
    if (!c) return fail(QMCP_EINVAL, "null context");
    c->profiling = enabled == 2 ? 2 : (enabled != 0 ? 1 : 0);
    c->acc.clear();
    return QMCP_OK;
}

int qmcp_hip_kernel_times(qmcp_hip_ctx* c, char* buf, size_t cap) {
    if (!c || !buf || cap == 0) return fail(QMCP_EINVAL, "null argument");
    size_t used = 0;
    buf[0] = 0;
    for (const auto& a : c->acc) {
        int w = snprintf(buf + used, cap - used, "%s\t%llu\t%.6f\n", a.name.c_str(),
                         (unsigned long long)a.launches, a.ms);
        if (w < 0 || (size_t)w >= cap - used) return fail(QMCP_EINVAL, "buffer too small");
        used += (size_t)w;
    }
    return (int)c->acc.size();
}

int qmcp_hip_solve_device(qmcp_hip_ctx* c, const uint32_t* d_starts, const uint32_t* d_ends,
                          uint64_t n_reads, const uint64_t* contig_read_offsets,
                          const uint32_t* contig_lengths, uint32_t n_contigs, uint32_t max_coverage,
                          uint64_t* d_keep_mask_out, void* hip_stream, qmcp_hip_stats* stats) {
    TRY(use_device(c));
    if (n_reads && (!d_starts || !d_ends || !d_keep_mask_out)) return fail(QMCP_EINVAL, "null buffer");
    TRY(order_after(c, hip_stream));
    return solve_on_device(c, d_starts, d_ends, contig_read_offsets, contig_lengths, n_contigs,
                           n_reads, max_coverage, d_keep_mask_out, stats);
}

int qmcp_hip_solve_device_begin(qmcp_hip_ctx* c, const uint32_t* d_starts, const uint32_t* d_ends,
                                uint64_t n_reads, const uint64_t* contig_read_offsets,
                                const uint32_t* contig_lengths, uint32_t n_contigs, uint32_t max_coverage,
                                uint64_t* d_keep_mask_out, void* hip_stream) {
    TRY(use_device(c));
    if (n_reads && (!d_starts || !d_ends || !d_keep_mask_out)) return fail(QMCP_EINVAL, "null buffer");
    TRY(order_after(c, hip_stream));
    return solve_enqueue(c, d_starts, d_ends, contig_read_offsets, contig_lengths, n_contigs, n_reads,
                         max_coverage, d_keep_mask_out);
}

int qmcp_hip_solve_end(qmcp_hip_ctx* c, qmcp_hip_stats* stats) {
    TRY(use_device(c, true));
    return solve_complete(c, stats);
}

int qmcp_hip_solve_host(qmcp_hip_ctx* c, const uint32_t* starts, const uint32_t* ends,
                        uint64_t n_reads, const uint64_t* contig_read_offsets,
                        const uint32_t* contig_lengths, uint32_t n_contigs, uint32_t max_coverage,
                        uint64_t* keep_mask_out, qmcp_hip_stats* stats) {
    TRY(use_device(c));
    if (n_reads && (!starts || !ends || !keep_mask_out)) return fail(QMCP_EINVAL, "null buffer");
    if (n_reads > (1ull << 30)) return fail(QMCP_ERANGE, "n_reads exceeds 2^30 per call");
    const size_t nb = (size_t)n_reads * sizeof(uint32_t);
    const size_t words = (size_t)((n_reads + 63) / 64);
    TRY(ensure(c, c->in_starts, nb));
    TRY(ensure(c, c->in_ends, nb));
    TRY(ensure(c, c->mask, words * sizeof(uint64_t)));
    hipEvent_t t0, t1, t2, t3;
    HIP_TRY(hipEventCreate(&t0)); HIP_TRY(hipEventCreate(&t1));
    HIP_TRY(hipEventCreate(&t2)); HIP_TRY(hipEventCreate(&t3));
    int rc = QMCP_OK;
    uint32_t sent_columns = 2;
    do {
        if (hipEventRecord(t0, c->stream) != hipSuccess) { rc = fail(QMCP_EHIP, "event record"); break; }
        if (nb) {
            rc = upload_columns(c, starts, ends, n_reads, &sent_columns);
            if (rc != QMCP_OK) break;
        }
        (void)hipEventRecord(t1, c->stream);
        c->mask_reads = 0;
        rc = solve_on_device(c, (const uint32_t*)c->in_starts.p, (const uint32_t*)c->in_ends.p,
                             contig_read_offsets, contig_lengths, n_contigs, n_reads, max_coverage,
                             (uint64_t*)c->mask.p, stats);
        if (rc != QMCP_OK) break;
        c->mask_reads = n_reads;
        (void)hipEventRecord(t2, c->stream);
        if (words) {
            if (hipMemcpyAsync(keep_mask_out, c->mask.p, words * sizeof(uint64_t),
                               hipMemcpyDeviceToHost, c->stream) != hipSuccess) {
                rc = fail(QMCP_EHIP, "D2H copy failed: %s", hipGetErrorString(hipGetLastError()));
                break;
            }
        }
        (void)hipEventRecord(t3, c->stream);
        if (hipStreamSynchronize(c->stream) != hipSuccess) { rc = fail(QMCP_EHIP, "stream sync failed"); break; }
        if (stats) {
            stats->ms_h2d = elapsed(t0, t1);
            stats->ms_d2h = elapsed(t2, t3);
            stats->columns_sent = sent_columns;
        }
    } while (0);
    (void)hipEventDestroy(t0); (void)hipEventDestroy(t1);
    (void)hipEventDestroy(t2); (void)hipEventDestroy(t3);
    return rc;
}

int qmcp_hip_solve_host64(qmcp_hip_ctx* c, const uint64_t* start_inds, const uint64_t* end_inds,
                          uint64_t n_reads, const uint64_t* contig_read_offsets,
                          const uint32_t* contig_lengths, uint32_t n_contigs, uint32_t max_coverage,
                          uint64_t* keep_mask_out, qmcp_hip_stats* stats, qmcp_hip_host_breakdown* breakdown) {
    using clock = std::chrono::steady_clock;
    auto ms_since = [](clock::time_point t) { return std::chrono::duration<float, std::milli>(clock::now() - t).count(); };
    const clock::time_point t_begin = clock::now();
    TRY(use_device(c));
    if (c->pending) return fail(QMCP_EINVAL, "a solve is pending on this context (call qmcp_hip_solve_end)");
    if (n_reads && (!start_inds || !end_inds)) return fail(QMCP_EINVAL, "null buffer");
    if (n_reads > (1ull << 30)) return fail(QMCP_ERANGE, "n_reads exceeds 2^30 per call");
    c->mask_reads = 0;
    const size_t n = (size_t)n_reads;
    const size_t words = (n + 63) / 64;
    TRY(ensure(c, c->in_starts, n * sizeof(uint32_t)));
    TRY(ensure(c, c->in_ends, n * sizeof(uint32_t)));
    TRY(ensure(c, c->mask, words * sizeof(uint64_t)));
    // chunks of 256 Ki reads (2 MiB of staging, 1 MiB per copy); thread t takes chunks t, t + T, ... and
    // owns two staging slots, so no slot is ever shared: before reusing a slot it waits for the copy it
    // issued from it two chunks ago
    constexpr size_t kChunk = 1u << 18;
    const size_t n_chunks = (n + kChunk - 1) / kChunk;
    unsigned want = 8;
    if (const char* e = std::getenv("QMCP_HIP_HOST_THREADS")) want = (unsigned)std::strtoul(e, nullptr, 10);
    const unsigned hw = std::thread::hardware_concurrency();
    unsigned T = want < 1 ? 1 : want;
    if (hw != 0 && T > hw) T = hw;
    if (T > n_chunks) T = (unsigned)(n_chunks ? n_chunks : 1);
    const size_t stage_words = (size_t)T * 2 * 2 * kChunk;  // T threads x 2 slots x (starts + ends)
    if (c->h_stage_words < stage_words) {
        if (c->h_stage) HIP_TRY(hipHostFree(c->h_stage));
        c->h_stage = nullptr;
        c->h_stage_words = 0;
        HIP_TRY(hipHostMalloc((void**)&c->h_stage, stage_words * sizeof(uint32_t), hipHostMallocDefault));
        c->h_stage_words = stage_words;
    }
    if (c->h_mask_words < words) {
        if (c->h_mask) HIP_TRY(hipHostFree(c->h_mask));
        c->h_mask = nullptr;
        c->h_mask_words = 0;
        HIP_TRY(hipHostMalloc((void**)&c->h_mask, (words ? words : 1) * sizeof(uint64_t), hipHostMallocDefault));
        c->h_mask_words = words ? words : 1;
    }
    while (c->stage_ev.size() < (size_t)T * 2) {
        hipEvent_t e = nullptr;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->stage_ev.push_back(e);
    }
    unsigned n_streams = 4;
    if (const char* e = std::getenv("QMCP_HIP_COPY_STREAMS")) n_streams = (unsigned)std::strtoul(e, nullptr, 10);
    if (n_streams < 1) n_streams = 1;
    if (n_streams > 8) n_streams = 8;
    while (c->stage_streams.size() < n_streams) {
        hipStream_t st = nullptr;
        hipEvent_t e = nullptr;
        HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        c->stage_streams.push_back(st);
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->stage_done.push_back(e);
    }
    const clock::time_point t_copy = clock::now();
    // One span for every read (all of reads-gen's inputs): only the starts cross the link and the device
    // rebuilds the ends.  Taken on the evidence of the first reads, checked on all of them while they are
    // narrowed; a call that turns out mixed after all sends its ends in a second pass.
    uint64_t span0 = 0;
    bool send_starts_only = n != 0 && end_inds[0] >= start_inds[0] && end_inds[0] - start_inds[0] < (1ull << 24) &&
                            std::getenv("QMCP_HIP_HOST_BOTH_COLUMNS") == nullptr;
    if (send_starts_only) {
        span0 = end_inds[0] - start_inds[0];
        const size_t probe = n < 4096 ? n : 4096;
        uint64_t differs = 0;
        for (size_t i = 0; i < probe; ++i) differs |= (end_inds[i] - start_inds[i]) ^ span0;
        send_starts_only = differs == 0;
    }
    std::atomic<uint64_t> high_bits{0}, span_differs{0};
    std::atomic<int> hip_failed{0};
    enum Pass { kBothColumns, kStartsChecked, kEndsOnly };
    auto worker = [&](unsigned t, Pass pass) {
        if (hipSetDevice(c->device) != hipSuccess) { hip_failed = 1; return; }
        uint64_t hi = 0, differs = 0;
        unsigned use = 0;
        hipStream_t cs = c->stage_streams[t % n_streams];
        for (size_t k = t; k < n_chunks; k += T, ++use) {
            const unsigned slot = use & 1u;
            hipEvent_t ev = c->stage_ev[(size_t)t * 2 + slot];
            if (use >= 2 && hipEventSynchronize(ev) != hipSuccess) { hip_failed = 1; return; }
            uint32_t* ss = c->h_stage + ((size_t)t * 2 + slot) * 2 * kChunk;
            uint32_t* ee = ss + kChunk;
            const size_t lo = k * kChunk, cnt = (lo + kChunk <= n ? kChunk : n - lo);
            const uint64_t* s64 = start_inds + lo;
            const uint64_t* e64 = end_inds + lo;
            bool ok = true;
            if (pass == kBothColumns) {
                for (size_t i = 0; i < cnt; ++i) {  // (branch-free: the range check is one OR per element)
                    const uint64_t a = s64[i], b = e64[i];
                    hi |= a | b;
                    ss[i] = (uint32_t)a;
                    ee[i] = (uint32_t)b;
                }
                ok = hipMemcpyAsync((uint32_t*)c->in_starts.p + lo, ss, cnt * sizeof(uint32_t), hipMemcpyHostToDevice, cs) == hipSuccess &&
                     hipMemcpyAsync((uint32_t*)c->in_ends.p + lo, ee, cnt * sizeof(uint32_t), hipMemcpyHostToDevice, cs) == hipSuccess;
            } else if (pass == kStartsChecked) {
                for (size_t i = 0; i < cnt; ++i) {
                    const uint64_t a = s64[i], b = e64[i];
                    hi |= a | b;
                    differs |= (b - a) ^ span0;
                    ss[i] = (uint32_t)a;
                }
                ok = hipMemcpyAsync((uint32_t*)c->in_starts.p + lo, ss, cnt * sizeof(uint32_t), hipMemcpyHostToDevice, cs) == hipSuccess;
            } else {
                for (size_t i = 0; i < cnt; ++i) ee[i] = (uint32_t)e64[i];  // (range-checked in the first pass)
                ok = hipMemcpyAsync((uint32_t*)c->in_ends.p + lo, ee, cnt * sizeof(uint32_t), hipMemcpyHostToDevice, cs) == hipSuccess;
            }
            if (!ok || hipEventRecord(ev, cs) != hipSuccess) { hip_failed = 1; return; }
        }
        // the staging slots are this thread's own in every pass: drain its last two copies before another pass reuses them
        for (unsigned u = 0; u < 2 && u < use; ++u)
            if (pass != kBothColumns && hipEventSynchronize(c->stage_ev[(size_t)t * 2 + u]) != hipSuccess) { hip_failed = 1; return; }
        high_bits.fetch_or(hi >> 32);
        span_differs.fetch_or(differs);
    };
    auto run_pass = [&](Pass pass) {
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < T; ++t) pool.emplace_back(worker, t, pass);
        worker(0, pass);
        for (auto& th : pool) th.join();
    };
    run_pass(send_starts_only ? kStartsChecked : kBothColumns);
    if (send_starts_only && span_differs.load() != 0 && !hip_failed.load() && high_bits.load() == 0) {
        send_starts_only = false;
        run_pass(kEndsOnly);
    }
    if (hip_failed.load()) return fail(QMCP_EHIP, "staging copy failed: %s", hipGetErrorString(hipGetLastError()));
    if (high_bits.load() != 0) {
        for (unsigned i = 0; i < n_streams; ++i) (void)hipStreamSynchronize(c->stage_streams[i]);
        return fail(QMCP_ERANGE, "a read coordinate exceeds 2^32 - 1");
    }
    // the solve follows the copies: an event edge from every copy stream to the solver stream
    for (unsigned i = 0; i < n_streams; ++i) {
        HIP_TRY(hipEventRecord(c->stage_done[i], c->stage_streams[i]));
        HIP_TRY(hipStreamWaitEvent(c->stream, c->stage_done[i], 0));
    }
    if (send_starts_only) qmcp::launch_fill_ends(c->stream, (const uint32_t*)c->in_starts.p, (uint32_t)n, (uint32_t)span0,
                                                 (uint32_t*)c->in_ends.p);
    for (unsigned i = 0; i < n_streams; ++i) HIP_TRY(hipStreamSynchronize(c->stage_streams[i]));  // (for the breakdown)
    const float ms_copy = ms_since(t_copy);
    const clock::time_point t_solve = clock::now();
    TRY(solve_on_device(c, (const uint32_t*)c->in_starts.p, (const uint32_t*)c->in_ends.p, contig_read_offsets,
                        contig_lengths, n_contigs, n_reads, max_coverage, (uint64_t*)c->mask.p, stats));
    const float ms_solve = ms_since(t_solve);
    c->mask_reads = n_reads;
    const clock::time_point t_d2h = clock::now();
    if (words && keep_mask_out) {  // (NULL: the caller will ask for qmcp_hip_kept_indices_host instead)
        HIP_TRY(hipMemcpyAsync(c->h_mask, c->mask.p, words * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        std::memcpy(keep_mask_out, c->h_mask, words * sizeof(uint64_t));
    }
    const float ms_d2h = ms_since(t_d2h);
    if (stats) { stats->ms_h2d = ms_copy; stats->ms_d2h = ms_d2h; stats->columns_sent = send_starts_only ? 1u : 2u; }
    if (breakdown) {
        breakdown->ms_total = ms_since(t_begin);
        breakdown->ms_narrow_h2d = ms_copy;
        breakdown->ms_solve = ms_solve;
        breakdown->ms_d2h = ms_d2h;
        breakdown->host_threads = T;
        breakdown->chunks = (uint32_t)n_chunks;
        breakdown->columns_sent = send_starts_only ? 1u : 2u;
    }
    return QMCP_OK;
}

int qmcp_hip_kept_indices_host(qmcp_hip_ctx* c, uint64_t n_reads, uint64_t* indices_out, uint64_t capacity,
                               uint64_t* n_out) {
    TRY(use_device(c));
    if (!n_out) return fail(QMCP_EINVAL, "null n_out");
    *n_out = 0;
    if (n_reads == 0) return QMCP_OK;
    if (c->mask_reads != n_reads || !c->mask.p)
        return fail(QMCP_EINVAL, "the context holds no keep mask of %llu reads (call a host solve first)",
                    (unsigned long long)n_reads);
    const uint32_t words = (uint32_t)((n_reads + 63) / 64);
    TRY(ensure(c, c->f_words, ((size_t)words + 2) * sizeof(uint32_t)));
    TRY(ensure(c, c->spine, (size_t)(qmcp::scan_spine_entries(words + 1) + 1) * sizeof(uint32_t) + 16));
    qmcp::launch_word_popcounts(c->stream, (const uint64_t*)c->mask.p, words, (uint32_t*)c->f_words.p);
    qmcp::launch_exclusive_scan(c->stream, (const uint32_t*)c->f_words.p, words, (uint32_t*)c->f_words.p,
                                (uint32_t*)c->spine.p, true);
    HIP_TRY(hipGetLastError());
    uint32_t total = 0;
    HIP_TRY(hipMemcpyAsync(&total, (uint32_t*)c->f_words.p + words, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    *n_out = total;
    if (total == 0) return QMCP_OK;
    if (!indices_out || capacity < total) return fail(QMCP_EINVAL, "indices_out holds %llu entries, %u are kept",
                                                      (unsigned long long)capacity, total);
    TRY(ensure(c, c->kidx, (size_t)total * sizeof(uint64_t)));
    qmcp::launch_mask_to_indices(c->stream, (const uint64_t*)c->mask.p, words, (const uint32_t*)c->f_words.p,
                                 (unsigned long long*)c->kidx.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(indices_out, c->kidx.p, (size_t)total * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return QMCP_OK;
}

int qmcp_hip_coverage_host(qmcp_hip_ctx* c, const uint32_t* starts, const uint32_t* ends,
                           uint64_t n_reads, const uint64_t* contig_read_offsets,
                           const uint32_t* contig_lengths, uint32_t n_contigs, uint32_t* cov_out) {
    if (!cov_out) return fail(QMCP_EINVAL, "null buffer");
    return coverage_common(c, starts, ends, n_reads, contig_read_offsets, contig_lengths, n_contigs,
                           nullptr, cov_out);
}

int qmcp_hip_filtered_coverage_host(qmcp_hip_ctx* c, const uint32_t* starts, const uint32_t* ends,
                                    uint64_t n_reads, const uint64_t* contig_read_offsets,
                                    const uint32_t* contig_lengths, uint32_t n_contigs,
                                    const uint64_t* keep_mask, uint32_t* cov_out) {
    if ((!keep_mask && n_reads) || !cov_out) return fail(QMCP_EINVAL, "null buffer");
    return coverage_common(c, starts, ends, n_reads, contig_read_offsets, contig_lengths, n_contigs,
                           keep_mask, cov_out);
}

int qmcp_hip_demand_host(qmcp_hip_ctx* c, const uint32_t* starts, const uint32_t* ends, uint64_t n_reads,
                         uint32_t ref_genome_length, uint32_t max_coverage, int32_t* b_out, int32_t* d_out) {
    if (!b_out || !d_out) return fail(QMCP_EINVAL, "null buffer");
    if (ref_genome_length == 0) return fail(QMCP_EINVAL, "ref_genome_length == 0");
    const uint64_t offs[2] = {0, n_reads};
    TRY(coverage_common(c, starts, ends, n_reads, offs, &ref_genome_length, 1, nullptr, nullptr));
    const size_t nb = ((size_t)ref_genome_length + 1) * sizeof(int32_t);
    TRY(ensure(c, c->ecnt, nb));  // free after the coverage: b
    TRY(ensure(c, c->eoff, nb));  //                          d
    qmcp::launch_b_and_demand(c->stream, (const uint32_t*)c->cov.p, ref_genome_length, max_coverage,
                              (int32_t*)c->ecnt.p, (int32_t*)c->eoff.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(b_out, c->ecnt.p, nb, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(d_out, c->eoff.p, nb, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return QMCP_OK;
}

int qmcp_hip_complete_pairs_device(qmcp_hip_ctx* c, uint64_t* d_keep_mask, uint64_t n_reads,
                                   void* hip_stream) {
    TRY(use_device(c));
    const uint64_t words = (n_reads + 63) / 64;
    if (words == 0) return QMCP_OK;
    if (!d_keep_mask) return fail(QMCP_EINVAL, "null mask");
    if (words > 0xFFFFFFFFull) return fail(QMCP_ERANGE, "mask too large");
    TRY(order_after(c, hip_stream));
    qmcp::launch_complete_pairs(c->stream, d_keep_mask, (uint32_t)words, n_reads);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    return QMCP_OK;
}

int qmcp_hip_complete_pairs_host(qmcp_hip_ctx* c, uint64_t* keep_mask, uint64_t n_reads) {
    TRY(use_device(c));
    const size_t words = (size_t)((n_reads + 63) / 64);
    if (words == 0) return QMCP_OK;
    if (!keep_mask) return fail(QMCP_EINVAL, "null mask");
    TRY(ensure(c, c->mask, words * 8));
    c->mask_reads = n_reads;
    HIP_TRY(hipMemcpyAsync(c->mask.p, keep_mask, words * 8, hipMemcpyHostToDevice, c->stream));
    qmcp::launch_complete_pairs(c->stream, (uint64_t*)c->mask.p, (uint32_t)words, n_reads);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(keep_mask, c->mask.p, words * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return QMCP_OK;
}

int qmcp_hip_amplicon_filter_host(qmcp_hip_ctx* c, const uint32_t* starts, const uint32_t* ends,
                                  const uint32_t* seq_lengths, const uint32_t* qualities,
                                  uint64_t n_reads, const uint32_t* amp_starts,
                                  const uint32_t* amp_ends, uint32_t n_amplicons,
                                  uint32_t min_length, uint32_t min_mapq, uint64_t* pair_keep_out) {
    TRY(use_device(c));
    const uint64_t n_pairs = n_reads / 2;
    const size_t words = (size_t)((n_pairs + 63) / 64);
    if (words == 0) return QMCP_OK;
    if (!starts || !ends || !pair_keep_out || (n_amplicons && (!amp_starts || !amp_ends)))
        return fail(QMCP_EINVAL, "null buffer");
    const size_t nb = (size_t)n_reads * 4;
    TRY(ensure(c, c->in_starts, nb));
    TRY(ensure(c, c->in_ends, nb));
    HIP_TRY(hipMemcpyAsync(c->in_starts.p, starts, nb, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->in_ends.p, ends, nb, hipMemcpyHostToDevice, c->stream));
    const uint32_t* d_len = nullptr;
    const uint32_t* d_q = nullptr;
    if (seq_lengths) {
        TRY(ensure(c, c->in_aux0, nb));
        HIP_TRY(hipMemcpyAsync(c->in_aux0.p, seq_lengths, nb, hipMemcpyHostToDevice, c->stream));
        d_len = (const uint32_t*)c->in_aux0.p;
    }
    if (qualities) {
        TRY(ensure(c, c->in_aux1, nb));
        HIP_TRY(hipMemcpyAsync(c->in_aux1.p, qualities, nb, hipMemcpyHostToDevice, c->stream));
        d_q = (const uint32_t*)c->in_aux1.p;
    }
    TRY(ensure(c, c->amp, (size_t)2 * (n_amplicons + 1) * 4));
    uint32_t* d_as = (uint32_t*)c->amp.p;
    uint32_t* d_ae = d_as + n_amplicons + 1;
    if (n_amplicons) {
        HIP_TRY(hipMemcpyAsync(d_as, amp_starts, (size_t)n_amplicons * 4, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(d_ae, amp_ends, (size_t)n_amplicons * 4, hipMemcpyHostToDevice, c->stream));
    }
    TRY(ensure(c, c->mask, words * 8));
    c->mask_reads = 0;  // (the buffer now holds pair bits)
    qmcp::launch_amplicon_filter(c->stream, (const uint32_t*)c->in_starts.p,
                                 (const uint32_t*)c->in_ends.p, d_len, d_q, n_pairs, d_as, d_ae,
                                 n_amplicons, min_length, min_mapq, (uint64_t*)c->mask.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(pair_keep_out, c->mask.p, words * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return QMCP_OK;
}

int qmcp_hip_filter_solve_host(qmcp_hip_ctx* c, const uint32_t* starts, const uint32_t* ends,
                               const uint32_t* seq_lengths, const uint32_t* qualities,
                               uint64_t n_reads, const uint32_t* amp_starts,
                               const uint32_t* amp_ends, uint32_t n_amplicons, uint32_t min_length,
                               uint32_t min_mapq, uint32_t ref_genome_length, uint32_t max_coverage,
                               int complete_pairs, uint64_t* keep_mask_out,
                               uint64_t* pairs_filtered_out, qmcp_hip_stats* stats) {
    TRY(use_device(c));
    if (n_reads & 1ull) return fail(QMCP_EINVAL, "n_reads must be even (reads come in mate pairs)");
    if (n_reads > (1ull << 30)) return fail(QMCP_ERANGE, "n_reads exceeds 2^30 per call");
    const size_t words = (size_t)((n_reads + 63) / 64);
    if (n_reads && (!starts || !ends || !keep_mask_out)) return fail(QMCP_EINVAL, "null buffer");
    if (n_amplicons && (!amp_starts || !amp_ends)) return fail(QMCP_EINVAL, "null amplicon table");
    if (pairs_filtered_out) *pairs_filtered_out = 0;
    if (stats) std::memset(stats, 0, sizeof(*stats));
    if (n_reads == 0) return QMCP_OK;
    const uint64_t n_pairs = n_reads / 2;
    const size_t pwords = (size_t)((n_pairs + 63) / 64);
    const size_t nb = (size_t)n_reads * 4;
    hipStream_t st = c->stream;
    TRY(ensure(c, c->in_starts, nb));
    TRY(ensure(c, c->in_ends, nb));
    TRY(ensure(c, c->f_starts, nb));
    TRY(ensure(c, c->f_ends, nb));
    TRY(ensure(c, c->f_map, (size_t)n_pairs * 4 + 16));
    TRY(ensure(c, c->f_words, (pwords + 2) * 4));
    TRY(ensure(c, c->f_mask, pwords * 8 + 16));
    TRY(ensure(c, c->mask, words * 8));
    TRY(ensure(c, c->cov, words * 8 + 16));  // compact-index keep mask
    TRY(ensure(c, c->spine, (size_t)(qmcp::scan_spine_entries((uint32_t)pwords + 1) + 1) * 4 + 16));
    uint32_t sent_columns = 2;
    TRY(upload_columns(c, starts, ends, n_reads, &sent_columns));
    const uint32_t* d_len = nullptr;
    const uint32_t* d_q = nullptr;
    if (seq_lengths) {
        TRY(ensure(c, c->in_aux0, nb));
        HIP_TRY(hipMemcpyAsync(c->in_aux0.p, seq_lengths, nb, hipMemcpyHostToDevice, st));
        d_len = (const uint32_t*)c->in_aux0.p;
    }
    if (qualities) {
        TRY(ensure(c, c->in_aux1, nb));
        HIP_TRY(hipMemcpyAsync(c->in_aux1.p, qualities, nb, hipMemcpyHostToDevice, st));
        d_q = (const uint32_t*)c->in_aux1.p;
    }
    // 1. FILTER predicate per pair.  Without amplicons (AmpliconBehaviour::IGNORE) one interval
    //    covering every coordinate stands in for the amplicon set.
    TRY(ensure(c, c->amp, (size_t)2 * ((size_t)n_amplicons + 2) * 4));
    uint32_t* d_as = (uint32_t*)c->amp.p;
    uint32_t* d_ae = d_as + n_amplicons + 2;
    uint32_t n_amp_eff = n_amplicons;
    if (n_amplicons) {
        HIP_TRY(hipMemcpyAsync(d_as, amp_starts, (size_t)n_amplicons * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d_ae, amp_ends, (size_t)n_amplicons * 4, hipMemcpyHostToDevice, st));
    } else {
        const uint32_t everything[2] = {0u, 0xFFFFFFFFu};
        HIP_TRY(hipMemcpyAsync(d_as, &everything[0], 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d_ae, &everything[1], 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));  // `everything` lives on this stack frame
        n_amp_eff = 1;
    }
    qmcp::launch_amplicon_filter(st, (const uint32_t*)c->in_starts.p, (const uint32_t*)c->in_ends.p,
                                 d_len, d_q, n_pairs, d_as, d_ae, n_amp_eff, min_length, min_mapq,
                                 (uint64_t*)c->f_mask.p);
    // 2. compaction: per-word popcounts -> exclusive scan -> scatter of surviving pairs
    qmcp::launch_word_popcounts(st, (const uint64_t*)c->f_mask.p, (uint32_t)pwords, (uint32_t*)c->f_words.p);
    qmcp::launch_exclusive_scan(st, (const uint32_t*)c->f_words.p, (uint32_t)pwords, (uint32_t*)c->f_words.p,
                                (uint32_t*)c->spine.p, true);
    HIP_TRY(hipGetLastError());
    uint32_t n_surv_pairs = 0;
    HIP_TRY(hipMemcpyAsync(&n_surv_pairs, (uint32_t*)c->f_words.p + pwords, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (pairs_filtered_out) *pairs_filtered_out = n_pairs - n_surv_pairs;
    qmcp::launch_compact_pairs(st, (const uint32_t*)c->in_starts.p, (const uint32_t*)c->in_ends.p,
                               (const uint64_t*)c->f_mask.p, (const uint32_t*)c->f_words.p, n_pairs,
                               (uint32_t*)c->f_starts.p, (uint32_t*)c->f_ends.p, (uint32_t*)c->f_map.p);
    HIP_TRY(hipGetLastError());
    // 3. solve the survivors (device-resident), 4. complete mates, 5. back to original indices
    const uint64_t n_c = 2ull * n_surv_pairs;
    const uint64_t offs[2] = {0, n_c};
    uint64_t* d_mask_c = (uint64_t*)c->cov.p;
    TRY(solve_on_device(c, (const uint32_t*)c->f_starts.p, (const uint32_t*)c->f_ends.p, offs,
                        &ref_genome_length, 1, n_c, max_coverage, d_mask_c, stats));
    const uint32_t words_c = (uint32_t)((n_c + 63) / 64);
    if (complete_pairs && words_c) {
        qmcp::launch_complete_pairs(st, d_mask_c, words_c, n_c);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipMemsetAsync(c->mask.p, 0, words * 8, st));
    if (n_c) {
        qmcp::launch_expand_mask(st, d_mask_c, (const uint32_t*)c->f_map.p, (uint32_t)n_c,
                                 (uint64_t*)c->mask.p);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipMemcpyAsync(keep_mask_out, c->mask.p, words * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    c->mask_reads = n_reads;
    if (stats) stats->columns_sent = sent_columns;
    return QMCP_OK;
}

}  // extern "C"

// ------------------------------------------------------------------ several devices, one call
struct qmcp_hip_multi {
    std::vector<qmcp_hip_ctx*> ctx;
    std::vector<std::vector<uint64_t>> local_mask;  // per device, reused across calls
};

namespace {

// cost model of a device's share (measured; the same numbers as
// genome-downsampler_amd/sharding.py): per read for the bandwidth-bound stages, per position of the
// LONGEST contig for the sweep (a device's chains run side by side)
constexpr double kNsPerRead = 0.008, kNsPerPosition = 1.5, kNsPerPositionStretches = 0.012;

// (sharding.py: share_sweeps_as_stretches / share_cost) a share's sweep is cut into stretches exactly when the
// solver would cut it: the AGGREGATE depth of everything the device owns (launch_uniform_sweep above)
bool share_sweeps_as_stretches(double reads, double positions, size_t n_contigs, uint32_t span, uint32_t M) {
    if (span == 0 || M == 0 || positions <= 0 || n_contigs >= 256) return false;
    const double depth = reads * (double)span / (positions * (double)M);
    if (depth <= kSpecMinDepth) return positions >= 128.0 * (double)span;  // nearly every window has a real cut
    return depth < kSpecDepth && positions >= 8.0 * (double)spec_burn_blocks(depth) * (double)span;
}
double share_cost(double reads, double positions, double longest, size_t n_contigs, uint32_t span, uint32_t M) {
    if (share_sweeps_as_stretches(reads, positions, n_contigs, span, M))
        return kNsPerRead * reads + kNsPerPositionStretches * positions;
    return kNsPerRead * reads + kNsPerPosition * longest;
}

void assign_contigs_by_cost(const uint64_t* roff, const uint32_t* lengths, uint32_t n_contigs, int n_dev,
                            uint32_t span, uint32_t M, std::vector<std::vector<uint32_t>>& owned) {
    owned.assign((size_t)n_dev, {});
    std::vector<uint32_t> order(n_contigs);
    for (uint32_t c = 0; c < n_contigs; ++c) order[c] = c;
    auto n_reads_of = [&](uint32_t c) { return (double)(roff[c + 1] - roff[c]); };
    auto alone = [&](uint32_t c) { return share_cost(n_reads_of(c), (double)lengths[c], (double)lengths[c], 1, span, M); };
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return alone(a) > alone(b); });
    std::vector<double> reads((size_t)n_dev, 0.0), longest((size_t)n_dev, 0.0), positions((size_t)n_dev, 0.0);
    for (uint32_t c : order) {
        int best = 0;
        double best_cost = 0;
        for (int d = 0; d < n_dev; ++d) {
            const double cost = share_cost(reads[d] + n_reads_of(c), positions[d] + (double)lengths[c],
                                           std::max(longest[d], (double)lengths[c]), owned[d].size() + 1, span, M);
            if (d == 0 || cost < best_cost) { best = d; best_cost = cost; }
        }
        owned[best].push_back(c);
        reads[best] += n_reads_of(c);
        positions[best] += (double)lengths[c];
        longest[best] = std::max(longest[best], (double)lengths[c]);
    }
    for (auto& o : owned) std::sort(o.begin(), o.end());
}

// OR `count` bits of src, from bit src_bit on, into dst from bit dst_bit on
void or_bits(uint64_t* dst, uint64_t dst_bit, const uint64_t* src, uint64_t src_bit, uint64_t count) {
    while (count != 0) {
        const unsigned so = (unsigned)(src_bit & 63), dof = (unsigned)(dst_bit & 63);
        unsigned take = 64 - (so > dof ? so : dof);  // stay inside one word on both sides
        if ((uint64_t)take > count) take = (unsigned)count;
        uint64_t v = src[src_bit >> 6] >> so;
        if (take < 64) v &= (1ull << take) - 1ull;
        dst[dst_bit >> 6] |= v << dof;
        src_bit += take; dst_bit += take; count -= take;
    }
}

}  // namespace

extern "C" {

int qmcp_hip_multi_create(const int* devices, int n_devices, qmcp_hip_multi** out) {
    if (!out) return fail(QMCP_EINVAL, "out is null");
    *out = nullptr;
    if (!devices || n_devices <= 0) return fail(QMCP_EINVAL, "no devices given");
    qmcp_hip_multi* m = new (std::nothrow) qmcp_hip_multi();
    if (!m) return fail(QMCP_ENOMEM, "host allocation failed");
    for (int i = 0; i < n_devices; ++i) {
        qmcp_hip_ctx* c = nullptr;
        const int rc = qmcp_hip_create(devices[i], &c);
        if (rc != QMCP_OK) { qmcp_hip_multi_destroy(m); return rc; }
        m->ctx.push_back(c);
    }
    m->local_mask.resize((size_t)n_devices);
    *out = m;
    return QMCP_OK;
}

void qmcp_hip_multi_destroy(qmcp_hip_multi* m) {
    if (!m) return;
    for (qmcp_hip_ctx* c : m->ctx) qmcp_hip_destroy(c);
    delete m;
}

int qmcp_hip_multi_solve_host(qmcp_hip_multi* m, const uint32_t* starts, const uint32_t* ends, uint64_t n_reads,
                              const uint64_t* contig_read_offsets, const uint32_t* contig_lengths,
                              uint32_t n_contigs, uint32_t max_coverage, uint64_t* keep_mask_out,
                              qmcp_hip_stats* per_device_stats, int* contig_device_out) {
    if (!m || m->ctx.empty()) return fail(QMCP_EINVAL, "null multi-device handle");
    if (n_reads && (!starts || !ends || !keep_mask_out)) return fail(QMCP_EINVAL, "null buffer");
    Problem pr;
    TRY(check_problem(contig_read_offsets, contig_lengths, n_contigs, n_reads, pr));
    const int n_dev = (int)m->ctx.size();
    std::vector<std::vector<uint32_t>> owned;
    // (the first read's span stands for the read length in the cost model; a mix of lengths only shifts balance)
    const uint32_t span0 = n_reads != 0 && ends[0] >= starts[0] ? ends[0] - starts[0] + 1 : 0u;
    assign_contigs_by_cost(contig_read_offsets, contig_lengths, n_contigs, n_dev, span0, max_coverage, owned);
    if (contig_device_out)
        for (int d = 0; d < n_dev; ++d)
            for (uint32_t c : owned[d]) contig_device_out[c] = d;
    const size_t words = (size_t)((n_reads + 63) / 64);
    std::memset(keep_mask_out, 0, words * sizeof(uint64_t));
    std::vector<int> rcs((size_t)n_dev, QMCP_OK);
    std::vector<std::string> msgs((size_t)n_dev);
    auto worker = [&](int d) {
        qmcp_hip_ctx* c = m->ctx[d];
        auto body = [&]() -> int {
            TRY(use_device(c));
            if (c->pending) return fail(QMCP_EINVAL, "a solve is pending on a context of this handle");
            const std::vector<uint32_t>& mine = owned[d];
            std::vector<uint64_t> loffs(mine.size() + 1, 0);
            std::vector<uint32_t> llens(mine.size());
            for (size_t i = 0; i < mine.size(); ++i) {
                loffs[i + 1] = loffs[i] + (contig_read_offsets[mine[i] + 1] - contig_read_offsets[mine[i]]);
                llens[i] = contig_lengths[mine[i]];
            }
            const uint64_t ln = loffs.back();
            if (per_device_stats) std::memset(&per_device_stats[d], 0, sizeof(qmcp_hip_stats));
            if (mine.empty() || ln == 0) return QMCP_OK;
            TRY(ensure(c, c->in_starts, (size_t)ln * sizeof(uint32_t)));
            TRY(ensure(c, c->in_ends, (size_t)ln * sizeof(uint32_t)));
            const size_t lwords = (size_t)((ln + 63) / 64);
            TRY(ensure(c, c->mask, lwords * sizeof(uint64_t)));
            c->mask_reads = 0;
            // a device's reads are its contigs' slices of the caller's arrays, copied one contig at a time
            // straight to their place in the local problem (no host-side concatenation)
            for (size_t i = 0; i < mine.size(); ++i) {
                const uint64_t lo = contig_read_offsets[mine[i]], cnt = loffs[i + 1] - loffs[i];
                if (cnt == 0) continue;
                HIP_TRY(hipMemcpyAsync((uint32_t*)c->in_starts.p + loffs[i], starts + lo, (size_t)cnt * sizeof(uint32_t),
                                       hipMemcpyHostToDevice, c->stream));
                HIP_TRY(hipMemcpyAsync((uint32_t*)c->in_ends.p + loffs[i], ends + lo, (size_t)cnt * sizeof(uint32_t),
                                       hipMemcpyHostToDevice, c->stream));
            }
            TRY(solve_on_device(c, (const uint32_t*)c->in_starts.p, (const uint32_t*)c->in_ends.p, loffs.data(),
                                llens.data(), (uint32_t)mine.size(), ln, max_coverage, (uint64_t*)c->mask.p,
                                per_device_stats ? &per_device_stats[d] : nullptr));
            std::vector<uint64_t>& lm = m->local_mask[d];
            lm.resize(lwords);
            HIP_TRY(hipMemcpyAsync(lm.data(), c->mask.p, lwords * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            return QMCP_OK;
        };
        rcs[d] = body();
        if (rcs[d] != QMCP_OK) msgs[d] = g_err;  // (thread-local: carry it to the caller's thread)
    };
    {
        std::vector<std::thread> pool;
        for (int d = 1; d < n_dev; ++d) pool.emplace_back(worker, d);
        worker(0);
        for (auto& th : pool) th.join();
    }
    for (int d = 0; d < n_dev; ++d)
        if (rcs[d] != QMCP_OK) return fail(rcs[d], "device %d of the handle: %s", d, msgs[d].c_str());
    // merge: every contig's bits from its device's local mask to its global ReadIndex positions
    for (int d = 0; d < n_dev; ++d) {
        uint64_t local_bit = 0;
        for (uint32_t c : owned[d]) {
            const uint64_t cnt = contig_read_offsets[c + 1] - contig_read_offsets[c];
            if (cnt) or_bits(keep_mask_out, contig_read_offsets[c], m->local_mask[d].data(), local_bit, cnt);
            local_bit += cnt;
        }
    }
    return QMCP_OK;
}

}  // extern "C"
