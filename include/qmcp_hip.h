/*
 * qmcp_hip.h -- C ABI of the MI355X-native quasi-MCP coverage-downsampling solver.
 *
 * This is the drop-in boundary for the one hot path this repository accelerates:
 * the `-a quasi-mcp-*` solver behind the reference's plugin surface
 *
 *     qmcp::Solver::solve(uint32_t max_coverage, bam_api::BamApi&)
 *         -> std::unique_ptr<std::vector<bam_api::ReadIndex>>
 *     (reference: libs/qmcp-solver/include/qmcp-solver/solver.hpp:13-20,
 *      registered by name in src/solver_manager.hpp:18-27).
 *
 * The reference has no FFI of its own (it is one C++ binary).  A maintainer drops
 * this library in by adding one `qmcp::Solver` subclass that narrows
 * `SOAPairedReads::start_inds/end_inds` (libs/bam-api/include/bam-api/soa_paired_reads.hpp:19-24)
 * to uint32 and calls `qmcp_hip_solve_host`; that adapter ships in
 * genome-downsampler_amd/host/ and is shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++ or torch types cross this boundary
 *   - read i is the inclusive interval [starts[i], ends[i]] on its contig
 *     (end_ind is inclusive in the reference: libs/bam-api/src/read.cpp:13)
 *   - reads are grouped by contig: contig c owns reads
 *     [contig_read_offsets[c], contig_read_offsets[c+1]); n_contigs == 1 reproduces the
 *     reference exactly (it is single-contig: libs/bam-api/src/bam_api.cpp:422)
 *   - the result is a keep bitmask: bit (i & 63) of word (i >> 6) is set iff read i is kept;
 *     expanding it in ascending order gives the reference's `Solution` vector
 *     (quasi_mcp_cpu_max_flow_solver.cpp:89-100)
 *   - every function returns QMCP_OK (0) or a negative QMCP_E* code; the message for the
 *     calling thread's last failure is available from qmcp_hip_last_error().  Nothing here
 *     terminates the process (the reference's CUDA path calls std::terminate():
 *     libs/qmcp-solver/include/qmcp-solver/cuda_helpers.cuh:13-22 -- the C++ adapter
 *     re-creates that behaviour on top of the status code).
 *   - there is no CPU fallback: without a usable HIP device every entry point fails.
 */
#ifndef QMCP_HIP_H
#define QMCP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QMCP_HIP_ABI_VERSION 5

enum {
    QMCP_OK = 0,
    QMCP_EINVAL = -1,      /* bad argument (null pointer, n_contigs == 0, offsets not monotone ...) */
    QMCP_EREAD = -2,       /* a read has start > end or end >= contig length                       */
    QMCP_ERANGE = -3,      /* problem exceeds the 32-bit coordinate / count limits of this build   */
    QMCP_ENODEVICE = -4,   /* no HIP device / device index out of range                            */
    QMCP_EHIP = -5,        /* a HIP runtime call failed (message has the hipError string)          */
    QMCP_ENOMEM = -6       /* device or host allocation failed                                     */
};

/* Selection paths the solver can take per call (reported in qmcp_hip_stats.path). */
enum {
    QMCP_PATH_NONE = 0,
    QMCP_PATH_UNIFORM = 1, /* all reads of the call have one span: block-parallel sweep            */
    QMCP_PATH_GENERAL = 2, /* mixed spans: event-driven priority sweep                             */
    QMCP_PATH_NEAR_UNIFORM = 3 /* one dominant span and a few shorter reads on deep data: the one-span sweep
                                  over the regular reads, the others selected as verified exceptions       */
};

/* Opaque solver context: owns one HIP stream and a reusable device arena.  Mirrors the
 * lifetime rules of a reference solver instance (constructed once, `solve` called many
 * times: src/tests/coverage_tester.cpp:30-34); creating it is the first and only point
 * where the GPU is touched (reference solvers are constructed eagerly even for --help:
 * src/app.hpp:35, so the C++ adapter creates the context lazily on first solve). */
typedef struct qmcp_hip_ctx qmcp_hip_ctx;

typedef struct qmcp_hip_stats {
    uint64_t n_reads;
    uint64_t n_kept;          /* popcount of the keep mask                                        */
    uint64_t total_length;    /* sum of contig lengths                                            */
    uint32_t n_contigs;
    uint32_t path;            /* QMCP_PATH_*                                                      */
    uint32_t min_span;        /* min / max of (end - start + 1) over the call                     */
    uint32_t max_span;
    uint32_t sort_passes;     /* radix passes of the bucketing stage (1 = one range partition,    */
                              /* the large uniform-span route; >= 2 = LSD radix sort)             */
    uint32_t sweep_stretches; /* chains the sweep ran side by side: the non-empty contigs, or more    */
                              /* where cut points (coverage <= M) split them                       */
    float ms_total;           /* device time of the whole solve (HIP events on the solver stream) */
    float ms_prepare;         /* validate + span reduction + per-position start/end counts        */
    float ms_scan;            /* prefix scans -> bucket offsets / coverage                        */
    float ms_sort;            /* radix bucketing of reads by (start, span)                        */
    float ms_sweep;           /* selection sweep                                                  */
    float ms_mark;            /* keep-mask emission                                               */
    float ms_h2d;             /* host entry point only                                            */
    float ms_d2h;             /* host entry point only                                            */
    uint32_t columns_sent;    /* host entry points only: 1 = every read has one span, only the starts
                                 crossed the link and the device rebuilt the ends; 2 = both columns  */
    uint32_t spec_boundaries; /* stretches that started at a speculative boundary (data a few times
                                 deeper than M: no cut point, but the sweep forgets its start)      */
    uint32_t spec_mismatches; /* of those, how many disagreed with the stretch before them: the parts
                                 of the genome they lie in were swept again with three times the
                                 run-in                                                              */
    uint32_t spec_retry_mismatches; /* ... and how many disagreed in that sweep: those parts were
                                 swept exactly                                                       */
    uint32_t sweep_blocks_changed; /* event-driven sweep (deep data): blocks of one read length's positions that
                                 changed the kept profile -- the chain's serial work is ~170 instructions per
                                 changed block + ~60 per 16 blocks tested; block-scan sweeps: blocks redone in
                                 the general form                                                     */
    uint32_t sweep_blocks;    /* ... of this many blocks swept                                       */
    uint32_t arena_grown_mid_solve; /* device buffers that had to grow after the solve's first launch (a
                                 stall on queued work); 0 from the second call of a shape on         */
    uint32_t near_uniform_exceptions; /* QMCP_PATH_NEAR_UNIFORM (also when the route was tried and given up for
                                 the mixed-span one): reads shorter than the dominant span            */
    uint32_t near_uniform_selected;   /* ... of those, kept                                              */
    uint32_t near_uniform_rounds;     /* ... sweeps it took (1 = no exception was wanted by the sweep)   */
    uint32_t near_uniform_giveup;     /* 0, or why the route handed the call to the mixed-span one: QMCP_NU_GIVEUP_* */
} qmcp_hip_stats;

/* qmcp_hip_stats.near_uniform_giveup */
enum {
    QMCP_NU_GIVEUP_NONE = 0,
    QMCP_NU_GIVEUP_NOT_TRIED = 1,      /* switched off, small call, span or M outside the event-driven sweep, too shallow */
    QMCP_NU_GIVEUP_LONGER_READS = 2,   /* the dominant span is not the longest (a deletion lengthens a read)              */
    QMCP_NU_GIVEUP_TOO_MANY = 3,       /* more than a tenth of the reads are exceptions, or the list overflowed           */
    QMCP_NU_GIVEUP_HEAVY_RANGE = 4,    /* one position range holds too many reads for the ranked route                    */
    QMCP_NU_GIVEUP_UNMODELLED = 5,     /* a run of used-up buckets without an anchor, or too many suspects / neighbours   */
    QMCP_NU_GIVEUP_BUDGET = 6,         /* the rounds did not settle within the budget                                     */
    QMCP_NU_GIVEUP_REMEMBERED = 7      /* an earlier call of this shape on this context did not settle                    */
};

int qmcp_hip_abi_version(void);
const char* qmcp_hip_last_error(void);

/* Number of HIP devices visible to the process (0 if none); never initialises a context. */
int qmcp_hip_device_count(void);

int qmcp_hip_create(int device, qmcp_hip_ctx** out_ctx);
void qmcp_hip_destroy(qmcp_hip_ctx* ctx);

/* Per-context options: which of the (all exact) kernels and routes a solve takes where the library would otherwise
 * choose by the data, and the host entries' threads -- the counterpart of the reference solver's setters
 * (libs/qmcp-solver/include/qmcp-solver/quasi_mcp_cuda_max_flow_solver.hpp:30-31: set_block_size, set_kernel_cycles).
 * Every choice gives the same keep mask; the options exist for tests (every route is forced and compared with the
 * oracle), measurements and debugging.  0 means "the library chooses" in every field.
 * qmcp_hip_create initialises a context's options from the defaults and then from the environment variables named below
 * (a debug override, read once, there and nowhere else); qmcp_hip_set_options replaces them. */
typedef struct qmcp_hip_options {
    uint32_t struct_size;         /* sizeof(qmcp_hip_options) of the caller's build (the struct may grow at its end)      */
    int32_t pass_major;           /* range-ranked route: -1 the range-major form, +1 the pass-major form wherever its hard
                                     limits allow (QMCP_HIP_PM=0|1)                                                       */
    int32_t sweep;                /* one-length sweep: QMCP_SWEEP_* (QMCP_HIP_SWEEP=fast|gen|ev)                          */
    int32_t cut_points;           /* split contigs at cut points: -1 never, +1 always (QMCP_HIP_CUTS=0|1)                 */
    int32_t speculation;          /* speculative stretch boundaries: -1 never, +1 at any depth (QMCP_HIP_SPEC=0|1)        */
    uint32_t speculation_run_in;  /* blocks of run-in of the first tier (QMCP_HIP_SPEC_BURN)                              */
    int32_t near_uniform;         /* near-uniform route: -1 off (QMCP_HIP_NEAR=0)                                         */
    uint32_t near_uniform_rounds; /* its budget of rounds (QMCP_HIP_NEAR_ROUNDS)                                          */
    float near_uniform_min_depth; /* sigma depth below which it is not tried (default 1.5: DESIGN.md 4.1; the mean coverage in
                                     units of M where M = 50) (QMCP_HIP_NEAR_MIN_DEPTH)                                  */
    int32_t near_uniform_debug;   /* 1: what every pair of rounds did, to stderr (QMCP_HIP_NEAR_DEBUG); 2 (lab): the rounds
                                     after the first sweep every stretch again, not only those a selection reaches     */
    int32_t force_sort_route;     /* 1: the keep mask from the radix sort even where the ranked route applies
                                     (QMCP_HIP_NO_RANK)                                                                   */
    int32_t keep_expand;          /* 1: the event-driven sweep always expands its output (QMCP_HIP_EXPAND)                */
    int32_t mixed_sweep_in_lds;   /* 1: the LDS-cached mixed-span sweep instead of the register-resident one
                                     (QMCP_HIP_GENERAL_LDS)                                                               */
    uint32_t rank_min_reads;      /* calls below this many reads take the sort-based route (QMCP_HIP_RANK_MIN; 2^17)      */
    uint32_t host_threads;        /* host entries: threads that narrow / check the columns (QMCP_HIP_HOST_THREADS; 8)     */
    uint32_t copy_streams;        /* host entries: copy streams (QMCP_HIP_COPY_STREAMS)                                   */
    int32_t host_both_columns;    /* 1: host entries always send starts AND ends (QMCP_HIP_HOST_BOTH_COLUMNS)             */
} qmcp_hip_options;
enum { QMCP_SWEEP_AUTO = 0, QMCP_SWEEP_FAST = 1, QMCP_SWEEP_GENERAL = 2, QMCP_SWEEP_EVENTS = 3 };
void qmcp_hip_default_options(qmcp_hip_options* out);            /* all zero but struct_size                          */
int qmcp_hip_set_options(qmcp_hip_ctx* ctx, const qmcp_hip_options* options);
int qmcp_hip_get_options(qmcp_hip_ctx* ctx, qmcp_hip_options* out);

/* Instrumentation (the reference's only timing is the "solve took" wall-clock log line,
 * src/app.cpp:132-139).  With profiling on, kernel launches of a solve are bracketed by HIP events
 * on the stream they run on: enabled == 1 every kernel, enabled == 2 only the selection sweep (the
 * event records cost a few microseconds of device idle time per bracket).
 * qmcp_hip_kernel_times writes one line per kernel, "name<TAB>launches<TAB>total_ms", accumulated
 * since profiling was last switched on, and returns the number of lines (negative on error). */
int qmcp_hip_set_profiling(qmcp_hip_ctx* ctx, int enabled);
int qmcp_hip_kernel_times(qmcp_hip_ctx* ctx, char* buf, size_t cap);

/* Replaces QuasiMcpCpuMaxFlowSolver::solve / QuasiMcpCudaMaxFlowSolver::solve
 * (libs/qmcp-solver/src/quasi_mcp_cpu_max_flow_solver.cpp:11-28,
 *  libs/qmcp-solver/src/quasi_mcp_cuda_max_flow_solver.cu:319-435) for host-resident reads.
 * keep_mask_out has ceil(n_reads / 64) words and is fully overwritten.  stats may be NULL.
 * Limits (QMCP_ERANGE beyond them): 2^30 reads and 2^31 - 2 bases per call; reads of one length per
 * call take the block sweep (any length up to 512 bases, longer ones and any mix of lengths the event
 * sweeps, reads up to 2^24 - 1 bases); 2^28 reads per contig on the block sweep. */
int qmcp_hip_solve_host(qmcp_hip_ctx* ctx,
                        const uint32_t* starts, const uint32_t* ends, uint64_t n_reads,
                        const uint64_t* contig_read_offsets, const uint32_t* contig_lengths,
                        uint32_t n_contigs, uint32_t max_coverage,
                        uint64_t* keep_mask_out, qmcp_hip_stats* stats);

/* The same for callers that hold the reference's own columns: SOAPairedReads::start_inds / end_inds are
 * std::vector<size_t> (libs/bam-api/include/bam-api/soa_paired_reads.hpp:19-24, read.hpp:11-13), i.e.
 * 64-bit.  The narrowing to uint32 (the reference's CUDA solver narrows too:
 * quasi_mcp_cuda_max_flow_solver.hpp:19) runs here, chunk by chunk on several host threads into pinned
 * staging owned by the context, each chunk's host-to-device copy issued as soon as it is narrowed, so
 * the span the reference times as "solve took" (src/app.cpp:132-139) is the PCIe transfer plus little.
 * A coordinate above 2^32 - 1 fails with QMCP_ERANGE.  When every read of the call has one span (checked
 * on all of them while they are narrowed) only the starts cross the link and the device rebuilds the
 * ends; qmcp_hip_solve_host does the same for calls of 2^20 reads or more (host threads check while
 * the starts are copied).  `breakdown` (may be NULL) receives host wall-clock milliseconds of the
 * call's parts. */
typedef struct qmcp_hip_host_breakdown {
    float ms_total;        /* the whole call                                                        */
    float ms_narrow_h2d;   /* narrowing + host-to-device copies (overlapped with each other)        */
    float ms_solve;        /* enqueue to completion of the device solve                             */
    float ms_d2h;          /* keep mask to the host                                                 */
    uint32_t host_threads; /* threads that narrowed                                                 */
    uint32_t chunks;
    uint32_t columns_sent; /* 1: every read has one span, only the starts crossed the link (the device
                              rebuilt the ends); 2: starts and ends                                   */
} qmcp_hip_host_breakdown;
int qmcp_hip_solve_host64(qmcp_hip_ctx* ctx,
                          const uint64_t* start_inds, const uint64_t* end_inds, uint64_t n_reads,
                          const uint64_t* contig_read_offsets, const uint32_t* contig_lengths,
                          uint32_t n_contigs, uint32_t max_coverage,
                          uint64_t* keep_mask_out, qmcp_hip_stats* stats,
                          qmcp_hip_host_breakdown* breakdown);

/* Solution of the last qmcp_hip_solve_host / _host64 / complete_pairs_host call on this context as the
 * reference returns it: the ascending ReadIndex list of obtain_sequence
 * (quasi_mcp_cpu_max_flow_solver.cpp:89-100), expanded from the context's keep mask on the device
 * (per-word popcounts, a scan, one scatter) and copied out -- for a plugin adapter this replaces a host
 * loop over the mask.  `capacity` entries at least stats.n_kept; *n_out receives the count. */
int qmcp_hip_kept_indices_host(qmcp_hip_ctx* ctx, uint64_t n_reads, uint64_t* indices_out, uint64_t capacity,
                               uint64_t* n_out);

/* Same solve with reads and mask already resident in this context's device memory
 * (d_* are device pointers; contig tables stay on the host).  `hip_stream` is a
 * hipStream_t the caller's producer work was enqueued on, or NULL: the solve is ordered
 * after it and the call returns after the solve has completed on the device. */
int qmcp_hip_solve_device(qmcp_hip_ctx* ctx,
                          const uint32_t* d_starts, const uint32_t* d_ends, uint64_t n_reads,
                          const uint64_t* contig_read_offsets, const uint32_t* contig_lengths,
                          uint32_t n_contigs, uint32_t max_coverage,
                          uint64_t* d_keep_mask_out, void* hip_stream, qmcp_hip_stats* stats);

/* The same solve in two halves, for callers that keep more than one solve in flight (one per
 * context: two contexts on one device let the selection sweep of one call -- a serial chain on a
 * few compute units -- run beside the bandwidth-bound stages of the next).  _begin orders the solve
 * after `hip_stream`, enqueues all of it and returns without waiting for the device (it does wait
 * for one 16-byte read-back that picks the kernels; the device keeps working on other contexts
 * meanwhile).  That holds for calls whose reads have ONE length (QMCP_PATH_UNIFORM).  A call with other
 * lengths is decided by what the device finds: on QMCP_PATH_NEAR_UNIFORM the FIRST call of a shape (reads, positions,
 * dominant length, M) waits for the device two to three times and once more per pair of rounds (each round's outcome
 * decides whether another is queued) -- _begin then returns when most of the solve has RUN.  The context remembers in
 * how many rounds the shape settled (up to eight), and the next call of it queues that many rounds and the ranking
 * behind them without looking: _begin returns at once again, _end looks at the route's state words and, should
 * the call have needed more rounds than were queued, solves it again the blocking way before it returns (bench.py
 * reports both rates for cfg4 with 1 % of the reads shortened: other_configs.cfg4_1pct_clipped.pipelined_ms 1.75
 * against device_ms 2.27; before round 4's second half 2.24).  On QMCP_PATH_GENERAL _begin waits once more where it
 * samples the lengths.  A context's first call
 * of a shape may also grow its arena after work is queued (stats.arena_grown_mid_solve), which waits for every
 * stream of the context; the second call of the shape does not.  _end waits for the solve and fills `stats`.  One pending solve per context: a second
 * _begin, or any other entry point of the same context, before _end fails with QMCP_EINVAL.
 * The reference has no counterpart: its solve is one blocking call (src/app.cpp:132-139). */
int qmcp_hip_solve_device_begin(qmcp_hip_ctx* ctx,
                                const uint32_t* d_starts, const uint32_t* d_ends, uint64_t n_reads,
                                const uint64_t* contig_read_offsets, const uint32_t* contig_lengths,
                                uint32_t n_contigs, uint32_t max_coverage,
                                uint64_t* d_keep_mask_out, void* hip_stream);
int qmcp_hip_solve_end(qmcp_hip_ctx* ctx, qmcp_hip_stats* stats);

/* Several devices behind one call (the reference has no counterpart: src/solver_manager.hpp:18-27
 * registers single-device solvers).  Contigs are independent problems (the reference is single-contig,
 * libs/bam-api/src/bam_api.cpp:422), so they are dealt to the devices -- by a cost of reads plus the
 * longest contig a device owns (its sweep chains run side by side), longest-processing-time first --
 * and every device solves its share in its own context on its own host thread: no data-path exchange
 * between devices.  The per-device keep masks are merged into global ReadIndex bit positions on the
 * host (contig boundaries need not be multiples of 64).  `devices` may name a device more than once
 * (separate contexts on it).  per_device_stats: n_devices entries or NULL; contig_device_out:
 * n_contigs entries (index into `devices`) or NULL. */
typedef struct qmcp_hip_multi qmcp_hip_multi;
int qmcp_hip_multi_create(const int* devices, int n_devices, qmcp_hip_multi** out);
void qmcp_hip_multi_destroy(qmcp_hip_multi* m);
int qmcp_hip_multi_solve_host(qmcp_hip_multi* m,
                              const uint32_t* starts, const uint32_t* ends, uint64_t n_reads,
                              const uint64_t* contig_read_offsets, const uint32_t* contig_lengths,
                              uint32_t n_contigs, uint32_t max_coverage,
                              uint64_t* keep_mask_out, qmcp_hip_stats* per_device_stats,
                              int* contig_device_out);

/* Stage probe for parity tests of the deterministic half of the reference solver:
 * writes cov[p] for every base of every contig (contigs concatenated, sum(contig_lengths)
 * entries) -- the array BamApi::find_input_cover returns (libs/bam-api/src/bam_api.cpp:275-286)
 * and from which b and d of create_b_function / create_demand_function
 * (quasi_mcp_cpu_max_flow_solver.cpp:58-87) follow as b[p+1] = min(cov[p], M). */
int qmcp_hip_coverage_host(qmcp_hip_ctx* ctx,
                           const uint32_t* starts, const uint32_t* ends, uint64_t n_reads,
                           const uint64_t* contig_read_offsets, const uint32_t* contig_lengths,
                           uint32_t n_contigs, uint32_t* cov_out);

/* Coverage of the kept subset only (BamApi::find_filtered_cover, bam_api.cpp:288-301);
 * keep_mask is a host bitmask as produced by qmcp_hip_solve_host. */
int qmcp_hip_filtered_coverage_host(qmcp_hip_ctx* ctx,
                                    const uint32_t* starts, const uint32_t* ends, uint64_t n_reads,
                                    const uint64_t* contig_read_offsets,
                                    const uint32_t* contig_lengths, uint32_t n_contigs,
                                    const uint64_t* keep_mask, uint32_t* cov_out);

/* Stage probe: the capped coverage b and the demand d of the reference's flow network for one
 * contig, computed on the device -- create_b_function (quasi_mcp_cpu_max_flow_solver.cpp:58-73):
 * b[0] = 0, b[p + 1] = min(cov[p], M); create_demand_function (:75-87): d[0] = -b[1],
 * d[i] = b[i] - b[i + 1] for 1 <= i < n, d[n] = b[n].  Both outputs have ref_genome_length + 1
 * entries (the reference's std::vector<int>). */
int qmcp_hip_demand_host(qmcp_hip_ctx* ctx,
                         const uint32_t* starts, const uint32_t* ends, uint64_t n_reads,
                         uint32_t ref_genome_length, uint32_t max_coverage,
                         int32_t* b_out, int32_t* d_out);

/* BamApi::find_pairs (libs/bam-api/src/bam_api.cpp:239-273) on the bitmask: mates sit at
 * indices (2q, 2q+1) (bam_api.cpp:456-461), so completing pairs is an OR inside each
 * aligned bit pair.  In place on a device mask of ceil(n_reads/64) words. */
int qmcp_hip_complete_pairs_device(qmcp_hip_ctx* ctx, uint64_t* d_keep_mask, uint64_t n_reads,
                                   void* hip_stream);
int qmcp_hip_complete_pairs_host(qmcp_hip_ctx* ctx, uint64_t* keep_mask, uint64_t n_reads);

/* Amplicon FILTER pre-pass (BamApi::should_be_filtered_out with AmpliconBehaviour::FILTER,
 * libs/bam-api/src/bam_api.cpp:311-319; Amplicon::includes amplicon.cpp:5-7;
 * AmpliconSet::member_includes_both amplicon_set.cpp:5-9; min length / min MAPQ
 * bam_api.cpp:321-327).  Pair q = reads (2q, 2q+1).  pair_keep_out gets one bit per pair:
 * set iff the pair survives (both mates inside one amplicon [amp_start, amp_end] inclusive,
 * both seq_lengths >= min_length, both qualities >= min_mapq).  seq_lengths / qualities may
 * be NULL (that filter is then skipped, as with the reference's defaults of 0). */
int qmcp_hip_amplicon_filter_host(qmcp_hip_ctx* ctx,
                                  const uint32_t* starts, const uint32_t* ends,
                                  const uint32_t* seq_lengths, const uint32_t* qualities,
                                  uint64_t n_reads,
                                  const uint32_t* amp_starts, const uint32_t* amp_ends,
                                  uint32_t n_amplicons, uint32_t min_length, uint32_t min_mapq,
                                  uint64_t* pair_keep_out);

/* The device-resident part of App::execute around the solver (src/app.cpp:113-142) for one
 * contig, in one call: FILTER pre-pass as in qmcp_hip_amplicon_filter_host (n_amplicons == 0
 * skips the amplicon predicate, i.e. AmpliconBehaviour::IGNORE; seq_lengths / qualities may be
 * NULL), compaction of the surviving pairs on the device (what BamApi does while ingesting:
 * only accepted pairs are appended, bam_api.cpp:434-461), the solve on the survivors, optional
 * mate completion (BamApi::find_pairs, src/app.cpp:141), and the result expressed over the
 * ORIGINAL read indices.  keep_mask_out: ceil(n_reads/64) words.  pairs_filtered_out (may be
 * NULL) receives the number of pairs the pre-pass dropped (BamApi::get_filtered_out_reads
 * counts their reads).  n_reads must be even (whole pairs). */
int qmcp_hip_filter_solve_host(qmcp_hip_ctx* ctx,
                               const uint32_t* starts, const uint32_t* ends,
                               const uint32_t* seq_lengths, const uint32_t* qualities,
                               uint64_t n_reads,
                               const uint32_t* amp_starts, const uint32_t* amp_ends,
                               uint32_t n_amplicons, uint32_t min_length, uint32_t min_mapq,
                               uint32_t ref_genome_length, uint32_t max_coverage, int complete_pairs,
                               uint64_t* keep_mask_out, uint64_t* pairs_filtered_out,
                               qmcp_hip_stats* stats);

#ifdef __cplusplus
}
#endif
#endif /* QMCP_HIP_H */
