// host_entries.inc.hip -- part of qmcp_api.hip (one translation unit).
// Stage probes and host-column upload helpers (anonymous namespace), then the extern "C" entry points of include/qmcp_hip.h.
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
    bool speculate = n_reads >= (1u << 20) && !c->opt.host_both_columns;
    for (size_t i = 0; speculate && i < 4096; ++i) speculate = ends[i] - starts[i] == span0;
    if (speculate) {
        unsigned T = 8;
        if (c->opt.host_threads) T = c->opt.host_threads;
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

void qmcp_hip_default_options(qmcp_hip_options* out) {
    if (!out) return;
    std::memset(out, 0, sizeof(*out));
    out->struct_size = (uint32_t)sizeof(*out);
}

int qmcp_hip_set_options(qmcp_hip_ctx* c, const qmcp_hip_options* options) {
    if (!c || !options) return fail(QMCP_EINVAL, "null argument");
    if (c->pending) return fail(QMCP_EINVAL, "a solve is pending on this context (call qmcp_hip_solve_end first)");
    if (options->struct_size == 0 || options->struct_size > sizeof(qmcp_hip_options))
        return fail(QMCP_EINVAL, "qmcp_hip_options.struct_size %u is not a size this build knows (%zu at most)",
                    options->struct_size, sizeof(qmcp_hip_options));
    if (options->sweep < QMCP_SWEEP_AUTO || options->sweep > QMCP_SWEEP_EVENTS) return fail(QMCP_EINVAL, "options.sweep out of range");
    qmcp_hip_default_options(&c->opt);
    std::memcpy(&c->opt, options, options->struct_size);   // (an older caller's shorter struct: the rest stays "the library chooses")
    c->opt.struct_size = (uint32_t)sizeof(qmcp_hip_options);
    // (what a context remembered about earlier calls was learnt under other options)
    c->spiky_known = false;
    c->spec_hopeless_n = 0;
    c->nu_failed_n = 0;
    return QMCP_OK;
}

int qmcp_hip_get_options(qmcp_hip_ctx* c, qmcp_hip_options* out) {
    if (!c || !out) return fail(QMCP_EINVAL, "null argument");
    *out = c->opt;
    return QMCP_OK;
}

void qmcp_hip_destroy(qmcp_hip_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->stream2) (void)hipStreamSynchronize(c->stream2);
    DevBuf* bufs[] = {&c->roff, &c->poff, &c->stats, &c->cstart, &c->boff, &c->ecnt, &c->eoff,
                      &c->selend, &c->spine, &c->hist, &c->spine2, &c->hist2, &c->keys[0], &c->keys[1], &c->vals[0],
                      &c->vals[1], &c->in_starts, &c->in_ends, &c->in_aux0, &c->in_aux1, &c->mask,
                      &c->cov, &c->amp, &c->scalars, &c->next_head, &c->ranges, &c->rankamb, &c->pm_desc, &c->pm_work, &c->segs, &c->specsnap, &c->specflags, &c->rings, &c->evpk, &c->evlast, &c->kidx, &c->f_starts, &c->f_ends, &c->f_map, &c->f_words, &c->f_mask, &c->nu_exc, &c->nu_nadj, &c->nu_ce, &c->nu_state, &c->nu_sus, &c->nu_ckpt, &c->nu_prev};
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
    if (c->opt.host_threads) want = c->opt.host_threads;
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
    if (c->opt.copy_streams) n_streams = c->opt.copy_streams;
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
                            !c->opt.host_both_columns;
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
