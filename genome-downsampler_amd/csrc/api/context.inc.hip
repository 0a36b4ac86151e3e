// context.inc.hip -- part of qmcp_api.hip (one translation unit).
// The solver context, its device arena, per-kernel timing spans, problem checks, contig tables, small stage helpers.
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
    qmcp_hip_options opt = {};      // which kernels and routes a solve takes where the data would decide (qmcp_hip_set_options)
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
    // ... and a call of this shape settled within this many queued rounds: the next one queues them -- and the ranking
    // behind them -- without waiting for the device in between (qmcp_hip_solve_device_begin returns at once; the state
    // words are looked at when the solve is collected, and a call that turns out to need more is solved again the
    // blocking way)
    uint64_t nu_need_n = 0, nu_need_ltot = 0;
    uint32_t nu_need_ell = 0, nu_need_M = 0, nu_need_rounds = 0;
    bool nu_deferred = false;       // the pending solve's rounds were queued unseen
    DevBuf nu_exc, nu_nadj, nu_ce, nu_state, nu_sus, nu_ckpt, nu_prev;
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
static uint32_t rank_min_reads(const qmcp_hip_ctx* c) { return c->opt.rank_min_reads ? c->opt.rank_min_reads : (1u << 17); }

// shortest span the event-driven sweep is used for: its scratch is 256 bytes per block, i.e. grows as
// the span shrinks; at 32 positions it is 8 bytes per position, what the bucket offsets themselves take
static uint32_t ev_min_span() { return 32u; }

float elapsed(hipEvent_t a, hipEvent_t b) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, a, b) != hipSuccess) return 0.f;
    return ms;
}
