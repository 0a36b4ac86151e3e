/*
 * qmcp_oracle.c -- CPU oracle for the quasi-MCP coverage-downsampling solver path.
 *
 * TEST INFRASTRUCTURE ONLY.  Loaded by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py -- never by the product path.
 *
 * What it restates (reference = migoox/genome-downsampler, paths relative to its root):
 *
 *   deterministic half of `quasi-mcp-cpu`
 *     create_b_function        libs/qmcp-solver/src/quasi_mcp_cpu_max_flow_solver.cpp:58-73
 *     create_demand_function   ...:75-87
 *     create_network_flow_graph (exact arc emission order)  ...:30-56
 *     obtain_sequence (kept = reads whose arc carries flow) ...:89-100
 *   validity checker of the reference's only test
 *     BamApi::find_input_cover / find_filtered_cover  libs/bam-api/src/bam_api.cpp:275-301
 *     CoverageTester::is_out_cover_valid              src/tests/coverage_tester.cpp:95-107
 *   "next" rows
 *     BamApi::find_pairs                               libs/bam-api/src/bam_api.cpp:239-273
 *     amplicon FILTER predicate   bam_api.cpp:311-327, amplicon.cpp:5-7, amplicon_set.cpp:5-9
 *
 * The max-flow itself (quasi_mcp_cpu_max_flow_solver.cpp:19-20) lives in a third-party
 * dependency that is absent from the reference tree: Google OR-Tools, C++ release v9.9.3963
 * (pinned at scripts/install_libs.sh:109-111), `operations_research::SimpleMaxFlow`
 * (a sequential push-relabel).  Its published contract is "return a maximum s-t flow"; the
 * flow is NOT unique for this network: any subset F of reads with
 * cov_F(p) >= min(cov(p), M) for every base p is the support of a maximum flow (all source
 * arcs saturate; the free back arcs p+1 -> p absorb the surplus).  Which maximum flow
 * OR-Tools returns is an artefact of its arc order / queue discipline, and the reference's
 * own tests pin only the coverage inequality (coverage_tester.cpp:101-107,116,134,153).
 *
 * PARITY STATUS
 *   - b, d, the arc list and the input generator are pinned bit-exactly by golden vectors
 *     captured from the reference's own sources (tests/golden/).
 *   - the kept-read set is pinned to be *a maximum flow of the reference's network*:
 *     qmcp_oracle_check_flow() rebuilds the flow from the kept set on the exact reference
 *     graph and checks capacity, conservation and that every source arc is saturated
 *     (value == capacity of the cut around s, hence maximum); qmcp_oracle_maxflow_value()
 *     is an independent Dinic on the same arc list for small cases.
 *   - bit-identity with the particular flow OR-Tools v9.9 would return: PARITY UNPINNED
 *     (dependency unavailable here and on the GPU box; no golden kept set exists in the
 *     reference).  Among all maximum flows this oracle fixes one canonical answer:
 *
 *       sweep p = 0 .. L-1;  deficit = min(cov(p), M) - (#selected reads covering p);
 *       if deficit > 0 select that many not-yet-selected reads covering p, preferring
 *       the largest end, then the largest start, then the smallest read index.
 *
 *     This is the classical greedy for interval multi-cover; it returns a minimum-
 *     cardinality valid subset (what the reference's `mcp-cpu` optimises,
 *     mcp_cpu_cost_scaling_solver.cpp:45-48), checked against exhaustive search in tests/.
 *     The HIP solver must reproduce this set bit for bit.
 */
#include "qmcp_oracle.h"

#include <limits.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- FNV-1a-64 helpers */
/* h = 0xcbf29ce484222325; for each v (as u64): h = (h ^ v) * 0x100000001b3 */
uint64_t qmcp_oracle_fnv_init(void) { return 0xcbf29ce484222325ULL; }
uint64_t qmcp_oracle_fnv_mix(uint64_t h, uint64_t v) { return (h ^ v) * 0x100000001b3ULL; }

uint64_t qmcp_oracle_reads_fnv(const uint32_t* starts, const uint32_t* ends,
                               const uint32_t* qualities, uint64_t n) {
    uint64_t h = qmcp_oracle_fnv_init();
    for (uint64_t i = 0; i < n; ++i) {
        h = qmcp_oracle_fnv_mix(h, starts[i]);
        h = qmcp_oracle_fnv_mix(h, ends[i]);
        h = qmcp_oracle_fnv_mix(h, qualities ? qualities[i] : 0);
    }
    return h;
}

uint64_t qmcp_oracle_mask_fnv(const uint64_t* mask, uint64_t n_reads) {
    /* hash of the ascending kept ReadIndex list (the reference's Solution vector) */
    uint64_t h = qmcp_oracle_fnv_init();
    for (uint64_t i = 0; i < n_reads; ++i)
        if ((mask[i >> 6] >> (i & 63)) & 1ULL) h = qmcp_oracle_fnv_mix(h, i);
    return h;
}

static int reads_ok(const uint32_t* starts, const uint32_t* ends, uint64_t n, uint32_t ref_len) {
    for (uint64_t i = 0; i < n; ++i)
        if (starts[i] > ends[i] || ends[i] >= ref_len) return 0;
    return 1;
}

/* ---------------------------------------------------------------- b, d, arcs */

/* create_b_function, quasi_mcp_cpu_max_flow_solver.cpp:58-73: per-base increments of
 * b[j+1] for j in [start, end], then cap at M.  b has ref_len+1 entries, b[0] == 0. */
int qmcp_oracle_b_function(const uint32_t* starts, const uint32_t* ends, uint64_t n,
                           uint32_t ref_len, uint32_t M, int32_t* b) {
    if (!reads_ok(starts, ends, n, ref_len)) return -2;
    memset(b, 0, ((size_t)ref_len + 1) * sizeof(int32_t));
    for (uint64_t i = 0; i < n; ++i)
        for (uint32_t j = starts[i]; j <= ends[i]; ++j) ++b[(size_t)j + 1];
    for (size_t i = 0; i < (size_t)ref_len + 1; ++i)
        if ((uint32_t)b[i] > M) b[i] = (int32_t)M;
    return 0;
}

/* create_demand_function, :75-87: in place, ascending, loop bound i < ref_len, so
 * d[ref_len] keeps the capped b value and d[0] = -b[1]. */
void qmcp_oracle_demand_function(int32_t* b, uint32_t ref_len) {
    if (ref_len == 0) return;
    int32_t b_1 = b[1];
    for (size_t i = 1; i < (size_t)ref_len; ++i) b[i] = b[i] - b[i + 1];
    b[0] = -b_1;
}

/* create_network_flow_graph, :30-56.  Arc id == emission index:
 *   [0, N)        read arcs  start -> end+1, cap 1          (arc id == ReadIndex)
 *   [N, N+n)      back arcs  i+1 -> i, cap INT64_MAX, i = 0..n-1
 *   then for i = 0..n: d>0 -> i -> t cap d;  d<0 -> s -> i cap -d   (s = n+1, t = n+2) */
int qmcp_oracle_graph(const uint32_t* starts, const uint32_t* ends, uint64_t n,
                      uint32_t ref_len, uint32_t M, qmcp_oracle_graph_info* info,
                      int64_t* arcs_out) {
    int32_t* d = (int32_t*)malloc(((size_t)ref_len + 1) * sizeof(int32_t));
    if (!d) return -6;
    int rc = qmcp_oracle_b_function(starts, ends, n, ref_len, M, d);
    if (rc) { free(d); return rc; }
    qmcp_oracle_demand_function(d, ref_len);

    uint64_t h = qmcp_oracle_fnv_init();
    uint64_t na = 0, nt = 0;
    int64_t supply = 0, demand = 0;
    const int64_t s = (int64_t)ref_len + 1, t = s + 1;
#define EMIT(tail, head, cap)                                            \
    do {                                                                 \
        h = qmcp_oracle_fnv_mix(h, (uint64_t)(tail));                    \
        h = qmcp_oracle_fnv_mix(h, (uint64_t)(head));                    \
        h = qmcp_oracle_fnv_mix(h, (uint64_t)(cap));                     \
        if (arcs_out) {                                                  \
            arcs_out[3 * na] = (tail);                                   \
            arcs_out[3 * na + 1] = (head);                               \
            arcs_out[3 * na + 2] = (cap);                                \
        }                                                                \
        ++na;                                                            \
    } while (0)
    for (uint64_t i = 0; i < n; ++i) EMIT((int64_t)starts[i], (int64_t)ends[i] + 1, (int64_t)1);
    for (int64_t i = 0; i < (int64_t)ref_len; ++i) EMIT(i + 1, i, INT64_MAX);
    for (int64_t i = 0; i <= (int64_t)ref_len; ++i) {
        if (d[i] > 0) { EMIT(i, t, (int64_t)d[i]); demand += d[i]; ++nt; }
        else if (d[i] < 0) { EMIT(s, i, (int64_t)-d[i]); supply += -d[i]; ++nt; }
    }
#undef EMIT
    free(d);
    if (info) {
        info->n_arcs = na; info->n_terminal_arcs = nt;
        info->total_supply = supply; info->total_demand = demand; info->arc_fnv = h;
    }
    return 0;
}

/* ---------------------------------------------------------------- canonical selection */

typedef struct { const uint32_t* st; const uint32_t* en; } key_ctx;

/* a has priority over b: larger end, then larger start, then smaller index */
static inline int before(const key_ctx* k, uint32_t a, uint32_t b) {
    if (k->en[a] != k->en[b]) return k->en[a] > k->en[b];
    if (k->st[a] != k->st[b]) return k->st[a] > k->st[b];
    return a < b;
}
static void heap_push(const key_ctx* k, uint32_t* heap, uint64_t* size, uint32_t v) {
    uint64_t i = (*size)++;
    while (i > 0) {
        uint64_t p = (i - 1) >> 1;
        if (!before(k, v, heap[p])) break;
        heap[i] = heap[p]; i = p;
    }
    heap[i] = v;
}
static uint32_t heap_pop(const key_ctx* k, uint32_t* heap, uint64_t* size) {
    uint32_t top = heap[0];
    uint32_t v = heap[--(*size)];
    uint64_t i = 0, n = *size;
    for (;;) {
        uint64_t c = 2 * i + 1;
        if (c >= n) break;
        if (c + 1 < n && before(k, heap[c + 1], heap[c])) ++c;
        if (!before(k, heap[c], v)) break;
        heap[i] = heap[c]; i = c;
    }
    if (n) heap[i] = v;
    return top;
}

/* One contig.  Sets bit (mask_bit_base + i) for every kept read i; the mask must be
 * zeroed by the caller.  This plays the role of SimpleMaxFlow::Solve + obtain_sequence
 * (quasi_mcp_cpu_max_flow_solver.cpp:19-20,89-100) under the canonical rule in the header. */
int qmcp_oracle_select(const uint32_t* starts, const uint32_t* ends, uint64_t n,
                       uint32_t ref_len, uint32_t M, uint64_t* keep_mask, uint64_t mask_bit_base) {
    if (n >= 0xFFFFFFFFULL) return -3;
    if (!reads_ok(starts, ends, n, ref_len)) return -2;
    if (n == 0 || ref_len == 0) return 0;
    const size_t L = ref_len;
    /* need[p] = min(cov[p], M) = b[p+1] of create_b_function, via a difference array
     * (same values as the per-base loop; tests compare the two) */
    int64_t* diff = (int64_t*)calloc(L + 1, sizeof(int64_t));
    uint32_t* need = (uint32_t*)malloc(L * sizeof(uint32_t));
    uint64_t* off = (uint64_t*)calloc(L + 2, sizeof(uint64_t));
    uint32_t* order = (uint32_t*)malloc((size_t)n * sizeof(uint32_t));
    uint32_t* heap = (uint32_t*)malloc((size_t)n * sizeof(uint32_t));
    uint32_t* expire = (uint32_t*)calloc(L + 1, sizeof(uint32_t));
    if (!diff || !need || !off || !order || !heap || !expire) {
        free(diff); free(need); free(off); free(order); free(heap); free(expire);
        return -6;
    }
    for (uint64_t i = 0; i < n; ++i) { diff[starts[i]] += 1; diff[(size_t)ends[i] + 1] -= 1; }
    int64_t run = 0;
    for (size_t p = 0; p < L; ++p) {
        run += diff[p];
        need[p] = run > (int64_t)M ? M : (uint32_t)run;
    }
    /* bucket reads by start (stable) */
    for (uint64_t i = 0; i < n; ++i) off[(size_t)starts[i] + 1]++;
    for (size_t p = 0; p < L; ++p) off[p + 1] += off[p];
    {
        uint64_t* cur = (uint64_t*)malloc((L + 1) * sizeof(uint64_t));
        if (!cur) { free(diff); free(need); free(off); free(order); free(heap); free(expire); return -6; }
        memcpy(cur, off, (L + 1) * sizeof(uint64_t));
        for (uint64_t i = 0; i < n; ++i) order[cur[starts[i]]++] = (uint32_t)i;
        free(cur);
    }
    key_ctx k = { starts, ends };
    uint64_t hsize = 0;
    int64_t cur_cov = 0;
    for (size_t p = 0; p < L; ++p) {
        for (uint64_t j = off[p]; j < off[p + 1]; ++j) heap_push(&k, heap, &hsize, order[j]);
        int64_t deficit = (int64_t)need[p] - cur_cov;
        while (deficit > 0) {
            /* need <= cov guarantees a live candidate exists */
            uint32_t r = heap_pop(&k, heap, &hsize);
            if (ends[r] < p) continue; /* expired while waiting in the pool */
            uint64_t bit = mask_bit_base + r;
            keep_mask[bit >> 6] |= 1ULL << (bit & 63);
            expire[ends[r]]++;
            ++cur_cov;
            --deficit;
        }
        cur_cov -= expire[p];
    }
    free(diff); free(need); free(off); free(order); free(heap); free(expire);
    return 0;
}

/* Multi-contig wrapper with the signature of qmcp_hip_solve_host (include/qmcp_hip.h):
 * each contig is an independent reference-style solve (the reference is single-contig,
 * libs/bam-api/src/bam_api.cpp:422); ReadIndex stays global. */
int qmcp_oracle_solve(const uint32_t* starts, const uint32_t* ends, uint64_t n_reads,
                      const uint64_t* contig_read_offsets, const uint32_t* contig_lengths,
                      uint32_t n_contigs, uint32_t max_coverage, uint64_t* keep_mask_out) {
    if (!contig_read_offsets || !contig_lengths || n_contigs == 0 || !keep_mask_out) return -1;
    if (contig_read_offsets[0] != 0 || contig_read_offsets[n_contigs] != n_reads) return -1;
    memset(keep_mask_out, 0, (size_t)((n_reads + 63) / 64) * sizeof(uint64_t));
    for (uint32_t c = 0; c < n_contigs; ++c) {
        uint64_t lo = contig_read_offsets[c], hi = contig_read_offsets[c + 1];
        if (hi < lo) return -1;
        int rc = qmcp_oracle_select(starts + lo, ends + lo, hi - lo, contig_lengths[c],
                                    max_coverage, keep_mask_out, lo);
        if (rc) return rc;
    }
    return 0;
}

/* ---------------------------------------------------------------- validity checker */

/* find_input_cover (keep_mask == NULL) / find_filtered_cover, bam_api.cpp:275-301:
 * per-base increments into an array of ref_len entries. */
int qmcp_oracle_cover(const uint32_t* starts, const uint32_t* ends, uint64_t n,
                      uint32_t ref_len, const uint64_t* keep_mask, uint64_t mask_bit_base,
                      uint32_t* cov_out) {
    if (!reads_ok(starts, ends, n, ref_len)) return -2;
    memset(cov_out, 0, (size_t)ref_len * sizeof(uint32_t));
    for (uint64_t i = 0; i < n; ++i) {
        if (keep_mask) {
            uint64_t bit = mask_bit_base + i;
            if (!((keep_mask[bit >> 6] >> (bit & 63)) & 1ULL)) continue;
        }
        for (uint32_t j = starts[i]; j <= ends[i]; ++j) cov_out[j]++;
    }
    return 0;
}

/* is_out_cover_valid, coverage_tester.cpp:95-107: min(in, M) <= out everywhere */
int qmcp_oracle_is_out_cover_valid(const uint32_t* in_cover, const uint32_t* out_cover,
                                   uint32_t ref_len, uint32_t M) {
    for (size_t p = 0; p < ref_len; ++p) {
        uint32_t capped = in_cover[p] < M ? in_cover[p] : M;
        if (!(capped <= out_cover[p])) return 0;
    }
    return 1;
}

/* Flow certificate: put flow 1 on the arc of every kept read, saturate every terminal arc,
 * route x_i = out_cov(i) - min(cov(i), M) on back arc i+1 -> i, and check capacity
 * (x_i >= 0) and conservation at every node 0..n of the reference graph (:30-56).
 * Returns 1 when the kept set is the support of a maximum flow, 0 otherwise. */
int qmcp_oracle_check_flow(const uint32_t* starts, const uint32_t* ends, uint64_t n,
                           uint32_t ref_len, uint32_t M, const uint64_t* keep_mask,
                           uint64_t mask_bit_base, int64_t* flow_value_out) {
    const size_t L = ref_len;
    int32_t* d = (int32_t*)malloc((L + 1) * sizeof(int32_t));
    int64_t* kstart = (int64_t*)calloc(L + 2, sizeof(int64_t));
    int64_t* kend1 = (int64_t*)calloc(L + 2, sizeof(int64_t));
    int64_t* x = (int64_t*)calloc(L + 1, sizeof(int64_t));
    uint32_t* incov = (uint32_t*)malloc((L ? L : 1) * sizeof(uint32_t));
    uint32_t* outcov = (uint32_t*)malloc((L ? L : 1) * sizeof(uint32_t));
    int ok = 1;
    if (!d || !kstart || !kend1 || !x || !incov || !outcov) { ok = 0; goto done; }
    if (qmcp_oracle_b_function(starts, ends, n, ref_len, M, d)) { ok = 0; goto done; }
    qmcp_oracle_demand_function(d, ref_len);
    qmcp_oracle_cover(starts, ends, n, ref_len, NULL, 0, incov);
    qmcp_oracle_cover(starts, ends, n, ref_len, keep_mask, mask_bit_base, outcov);
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t bit = mask_bit_base + i;
        if ((keep_mask[bit >> 6] >> (bit & 63)) & 1ULL) {
            kstart[starts[i]]++;
            kend1[(size_t)ends[i] + 1]++;
        }
    }
    for (size_t i = 0; i < L; ++i) {
        int64_t nd = incov[i] < M ? incov[i] : M;
        x[i] = (int64_t)outcov[i] - nd;
        if (x[i] < 0) ok = 0; /* back-arc flow must be non-negative */
    }
    int64_t value = 0;
    for (size_t i = 0; i <= L; ++i) {
        int64_t in = kend1[i] + (i < L ? x[i] : 0) + (d[i] < 0 ? -(int64_t)d[i] : 0);
        int64_t out = kstart[i] + (i >= 1 ? x[i - 1] : 0) + (d[i] > 0 ? (int64_t)d[i] : 0);
        if (in != out) ok = 0;
        if (d[i] < 0) value += -(int64_t)d[i];
    }
    if (flow_value_out) *flow_value_out = value;
done:
    free(d); free(kstart); free(kend1); free(x); free(incov); free(outcov);
    return ok;
}

/* ---------------------------------------------------------------- independent max-flow */
/* Dinic on the reference's exact arc list; for small instances in tests only (it plays
 * SimpleMaxFlow::Solve's published contract: the VALUE of a maximum flow is unique). */
typedef struct { int to; int64_t cap; } dedge;
typedef struct {
    int nv, ne; dedge* e; int* head; int* nxt; int* level; int* it;
} dgraph;
static void dg_add(dgraph* g, int u, int v, int64_t c) {
    g->e[g->ne].to = v; g->e[g->ne].cap = c; g->nxt[g->ne] = g->head[u]; g->head[u] = g->ne++;
    g->e[g->ne].to = u; g->e[g->ne].cap = 0; g->nxt[g->ne] = g->head[v]; g->head[v] = g->ne++;
}
static int dg_bfs(dgraph* g, int s, int t, int* queue) {
    for (int i = 0; i < g->nv; ++i) g->level[i] = -1;
    int qh = 0, qt = 0; queue[qt++] = s; g->level[s] = 0;
    while (qh < qt) {
        int u = queue[qh++];
        for (int a = g->head[u]; a >= 0; a = g->nxt[a])
            if (g->e[a].cap > 0 && g->level[g->e[a].to] < 0) {
                g->level[g->e[a].to] = g->level[u] + 1; queue[qt++] = g->e[a].to;
            }
    }
    return g->level[t] >= 0;
}
static int64_t dg_dfs(dgraph* g, int u, int t, int64_t f) {
    if (u == t) return f;
    for (int* a = &g->it[u]; *a >= 0; *a = g->nxt[*a]) {
        dedge* ed = &g->e[*a];
        if (ed->cap > 0 && g->level[ed->to] == g->level[u] + 1) {
            int64_t got = dg_dfs(g, ed->to, t, f < ed->cap ? f : ed->cap);
            if (got > 0) { ed->cap -= got; g->e[*a ^ 1].cap += got; return got; }
        }
    }
    return 0;
}
int64_t qmcp_oracle_maxflow_value(const uint32_t* starts, const uint32_t* ends, uint64_t n,
                                  uint32_t ref_len, uint32_t M) {
    qmcp_oracle_graph_info info;
    size_t max_arcs = (size_t)n + 2 * (size_t)ref_len + 1;
    int64_t* arcs = (int64_t*)malloc(3 * max_arcs * sizeof(int64_t));
    if (!arcs) return -1;
    if (qmcp_oracle_graph(starts, ends, n, ref_len, M, &info, arcs)) { free(arcs); return -1; }
    dgraph g;
    g.nv = (int)ref_len + 3; g.ne = 0;
    g.e = (dedge*)malloc(2 * info.n_arcs * sizeof(dedge));
    g.nxt = (int*)malloc(2 * info.n_arcs * sizeof(int));
    g.head = (int*)malloc(g.nv * sizeof(int));
    g.level = (int*)malloc(g.nv * sizeof(int));
    g.it = (int*)malloc(g.nv * sizeof(int));
    int* queue = (int*)malloc(g.nv * sizeof(int));
    for (int i = 0; i < g.nv; ++i) g.head[i] = -1;
    for (uint64_t a = 0; a < info.n_arcs; ++a)
        dg_add(&g, (int)arcs[3 * a], (int)arcs[3 * a + 1], arcs[3 * a + 2]);
    int s = (int)ref_len + 1, t = s + 1;
    int64_t flow = 0;
    while (dg_bfs(&g, s, t, queue)) {
        memcpy(g.it, g.head, g.nv * sizeof(int));
        int64_t f;
        while ((f = dg_dfs(&g, s, t, INT64_MAX)) > 0) flow += f;
    }
    free(arcs); free(g.e); free(g.nxt); free(g.head); free(g.level); free(g.it); free(queue);
    return flow;
}

/* ---------------------------------------------------------------- "next" rows */

/* BamApi::find_pairs, bam_api.cpp:239-273: every kept id brings its mate
 * (id+1 if is_first_read else id-1; mates are adjacent, first mate on the even index). */
void qmcp_oracle_find_pairs(uint64_t* keep_mask, uint64_t n_reads) {
    for (uint64_t i = 0; i + 1 < n_reads; i += 2) {
        int a = (int)((keep_mask[i >> 6] >> (i & 63)) & 1ULL);
        int b = (int)((keep_mask[(i + 1) >> 6] >> ((i + 1) & 63)) & 1ULL);
        if (a | b) {
            keep_mask[i >> 6] |= 1ULL << (i & 63);
            keep_mask[(i + 1) >> 6] |= 1ULL << ((i + 1) & 63);
        }
    }
}

/* should_be_filtered_out with AmpliconBehaviour::FILTER, bam_api.cpp:311-327:
 * a pair survives iff both mates have quality >= min_mapq, seq_length >= min_length and
 * some amplicon includes both (amplicon.cpp:5-7: start <= r.start && r.end <= end). */
void qmcp_oracle_amplicon_filter(const uint32_t* starts, const uint32_t* ends,
                                 const uint32_t* seq_lengths, const uint32_t* qualities,
                                 uint64_t n_reads, const uint32_t* amp_starts,
                                 const uint32_t* amp_ends, uint32_t n_amplicons,
                                 uint32_t min_length, uint32_t min_mapq, uint64_t* pair_keep_out) {
    uint64_t n_pairs = n_reads / 2;
    memset(pair_keep_out, 0, (size_t)((n_pairs + 63) / 64) * sizeof(uint64_t));
    for (uint64_t q = 0; q < n_pairs; ++q) {
        uint64_t i = 2 * q, j = i + 1;
        int ok = 1;
        if (qualities && !(qualities[i] >= min_mapq && qualities[j] >= min_mapq)) ok = 0;
        if (seq_lengths && !(seq_lengths[i] >= min_length && seq_lengths[j] >= min_length)) ok = 0;
        int in_one = 0;
        for (uint32_t a = 0; a < n_amplicons && !in_one; ++a)
            in_one = amp_starts[a] <= starts[i] && ends[i] <= amp_ends[a] &&
                     amp_starts[a] <= starts[j] && ends[j] <= amp_ends[a];
        if (ok && in_one) pair_keep_out[q >> 6] |= 1ULL << (q & 63);
    }
}
