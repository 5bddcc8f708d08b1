// Internal interface between the C ABI (mpcx_api.hip) and the kernels.
#ifndef MPCX_KERNELS_H
#define MPCX_KERNELS_H

#include <hip/hip_runtime.h>
#include <stddef.h>

#include "../../include/mpcx.h"

// doubles of workspace per stage and lane for a kernel built for `nh` obstacle rows
size_t mpcx_stage_doubles(int nh);

// The workspace starts with a header: the problem counter (byte 0), the options (byte 256), and the hand-over
// area through which wavefronts pass old problems to worker wavefronts (byte 512 on): three counters (queue
// head, queue tail, problems finished; 128 bytes apart), a ring of MPCX_HANDOVER_SLOTS descriptor numbers, a
// busy word per descriptor, and the descriptors (32 doubles each: what a lane knows of its problem).
#define MPCX_HANDOVER_SLOTS 8192
#define MPCX_HO_RING 512
#define MPCX_HO_BUSY (MPCX_HO_RING + 4 * MPCX_HANDOVER_SLOTS)
#define MPCX_HO_DESC (MPCX_HO_BUSY + 4 * MPCX_HANDOVER_SLOTS)
#define MPCX_HO_BYTES (MPCX_HO_DESC + 256 * MPCX_HANDOVER_SLOTS)
#define MPCX_WS_HEADER_BYTES (512 + MPCX_HO_BYTES)

// Launch the fused SQP solve on `n_waves` persistent 64-lane workgroups.  All
// pointers inside `in` / `out` and `workspace` are device memory; `workspace`
// holds MPCX_WS_HEADER_BYTES (problem counter, options, hand-over area) + n_waves * ((N+1) + slots * ceil((N+1)/64)) *
// mpcx_stage_doubles(nh) * 64 doubles: per wavefront N+1 stage records for 64 problems, then `slots`
// regions to which its last problems are moved once the queue is empty (stages contiguous).
// Each wavefront works on at most `lanes` (1..64) problems at a time; with `coop_max` or fewer
// of them in progress it takes them one at a time, a lane per stage (0 = never); sweep L is
// done that way whenever `lin_max` or fewer problems of the wavefront wait for it.
// coop_layout = 1 (only with lanes <= coop_max and
// lanes * ceil((N+1)/64) <= N+1): the wavefront never runs the one-lane-per-problem sweeps and its
// workspace is laid out per problem, stages contiguous, so that its accesses coalesce.
hipError_t mpcx_launch_sqp(int nh, const mpcx_batch &in, const mpcx_opts &o, const mpcx_result &out,
                           double *workspace, int n_waves, int lanes, int coop_max, int lin_max, int coop_layout, int slots, hipStream_t stream);
// (old problems are handed over to worker wavefronts when n_waves is a full grid of 64-problem wavefronts;
// see mpcx_launch_sqp)

// initial guess of obstacle-free problems from the problem table (device pointers)
hipError_t mpcx_launch_guess(const double *pdata, const int32_t *n_obs, const double *width, int B, int N, double *xguess,
                             hipStream_t stream);

// outcome counters of a solved batch (device pointers; stats = 8 x int64, zeroed by the launcher)
hipError_t mpcx_launch_stats(const int32_t *status, const int32_t *sqp_iter, const int32_t *qp_iter, int B,
                             long long *stats, hipStream_t stream);

// sin/cos exactly as the solver computes them (arithmetic parity tests)
hipError_t mpcx_launch_probe(const double *xin, int n, double *s, double *c, hipStream_t stream);

#endif
