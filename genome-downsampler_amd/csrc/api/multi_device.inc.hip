// multi_device.inc.hip -- part of qmcp_api.hip (one translation unit).
// Several devices behind one call: contig assignment by cost, one context and host thread per device, host-side merge of the masks.
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
    return depth < kSpecDepth && positions >= 8.0 * (double)spec_burn_blocks(spec_depth_in_sigma(depth, M)) * (double)span;
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
