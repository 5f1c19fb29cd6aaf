/*
 * whitted_fast.hip -- the trace kernels in FAST arithmetic: FMAs exactly where the source writes them,
 * native v_rcp_f32 / v_sqrt_f32 / v_rsq_f32 (1 ulp).  This is the envelope an OpenCL device build of
 * the reference is allowed (x/y <= 2.5 ulp, sqrt <= 3 ulp, contraction permitted), and is the default,
 * benchmarked path.  Build (build.py): hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
 * -fno-slp-vectorize -- the SAME flags as the strict build; the compiler contracts nothing by itself.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "whitted_params.h"
#define WT_STRICT 0
#define WT_NS wt_fast
#define WT_LAUNCH_TRACE wt_fast_launch_trace
#define WT_LAUNCH_RAYGEN wt_fast_launch_raygen
#define WT_LAUNCH_SCHED wt_fast_launch_sched
#define WT_LAUNCH_UNIT wt_fast_launch_unit
#define WT_LAUNCH_UNIT_SCENE wt_fast_launch_unit_scene
#include "whitted_launch.inc"
