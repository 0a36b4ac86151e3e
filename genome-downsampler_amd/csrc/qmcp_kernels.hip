// qmcp_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the quasi-MCP solver path.
//
// Everything here is integer scatter / scan / selection: HBM- or latency-bound, no MFMA.
// wave = 64 lanes throughout.  The path these kernels replace is the reference's
//   coverage build        quasi_mcp_cpu_max_flow_solver.cpp:58-73 (O(N*len) per-base loop)
//   max-flow + readout    quasi_mcp_cpu_max_flow_solver.cpp:19-20,89-100
//   (CUDA equivalent      quasi_mcp_cuda_max_flow_solver.cu:12-79,319-435)
// with the canonical selection rule stated in oracle/qmcp_oracle.c.
//
// Coordinates: contigs are concatenated into one global axis; gpos = pos_offset[c] + pos,
// Ltot = sum of contig lengths.  Reads never cross a contig, so per-position prefix counts
// taken over the global axis cancel exactly at contig borders.
//
// One translation unit, kept in parts under kernels/ (included below, in dependency order):
//   wave_primitives          DPP helpers, wave reductions and scans
//   prepare_scan             k_prepare (validate, statistics, partition histogram), exclusive scan
//   radix_sort               LSD radix passes (sort-based routes)
//   bucket_offsets           sort-based routes: bucket offsets from the sorted keys (heads + reverse min-scan)
//   ranked_route             range partition (one or two levels), per-range offsets, ordered ranking
//   pass_major               the same route without the range-major copy: one pass over the reads sorts every pass of
//                            8 192 in place; the per-range kernels walk its slices (included last: uses the launchers' helpers)
//   sweep_uniform            block forms of the uniform-span sweep, single-wave kernel
//   sweep_segments           cut points (coverage <= M): contigs split into independently swept stretches
//   sweep_uniform_pipelines  seven-wave pipelines: fast form with checked fallback, all-general form
//   sweep_uniform_events     event-driven form for deep data: pack (whole chip), chain (one wave per contig,
//                            LDS-DMA ring), expand (whole chip)
//   sweep_mixed              mixed-span event sweeps (register-resident, LDS-cached, plain)
//   mark_and_next_rows       keep-mask emission for the sort-based routes, coverage probes, FILTER,
//                            pair compaction / completion
//   launchers                host-side launch wrappers declared in qmcp_kernels.h
//   near_uniform             one dominant span + a few shorter reads: the one-span sweep over the regular reads, the
//                            exceptions verified against it and selected one event at a time (included last)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qmcp_kernels.h"

namespace qmcp {

static constexpr int kWave = 64;
// "no constraint".  Real counts stay < 2^28 per contig (checked by the host).  In the map
// algebra below b, u and v never accumulate (a composite's b is <= its first element's b, its
// u and v are <= its last element's), so the largest intermediate is u + b <= 2 kInf + 2^29
// < 2^32: nothing wraps, no saturation is needed, and anything >= kInf just means "infinite"
// (every finite value is < 2^29 < kInf).
static constexpr uint32_t kInf = 0x40000000u;

#include "kernels/wave_primitives.inc.hip"
#include "kernels/prepare_scan.inc.hip"
#include "kernels/radix_sort.inc.hip"
#include "kernels/ranked_route.inc.hip"
#include "kernels/bucket_offsets.inc.hip"
#include "kernels/sweep_uniform.inc.hip"
#include "kernels/sweep_segments.inc.hip"
#include "kernels/sweep_uniform_pipelines.inc.hip"
#include "kernels/sweep_uniform_events.inc.hip"
#include "kernels/sweep_mixed.inc.hip"
#include "kernels/mark_and_next_rows.inc.hip"
#include "kernels/launchers.inc.hip"
#include "kernels/pass_major.inc.hip"
#include "kernels/near_uniform.inc.hip"

}  // namespace qmcp
