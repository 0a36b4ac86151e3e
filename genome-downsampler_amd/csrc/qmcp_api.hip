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
#include <cmath>
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

// One translation unit, kept in parts under api/ (included below, in dependency order):
//   context             the solver context, its device arena, timing spans, problem checks, small stage helpers
//   uniform_sweep       which one-length sweep a call takes and its launches (pipelines, event-driven form, stretches,
//                       speculative tiers)
//   near_uniform_sizes  the near-uniform route's sizes, round budget and buffers
//   solve_head          a solve's head: arena sizing, the range-ranked route's producers, bucket offsets; the pass-major
//                       ranking
//   near_uniform_route  the near-uniform route's half of a solve's tail
//   solve_tail          a solve's tail, collection, context creation, enqueue / complete
//   host_entries        probes, column upload, the extern "C" entry points
//   multi_device        several devices behind one call
#include "api/context.inc.hip"
#include "api/uniform_sweep.inc.hip"
#include "api/near_uniform_sizes.inc.hip"
#include "api/solve_head.inc.hip"
#include "api/near_uniform_route.inc.hip"
#include "api/solve_tail.inc.hip"
#include "api/host_entries.inc.hip"
#include "api/multi_device.inc.hip"
