/*
 * whitted_strict.hip -- the same kernels in STRICT arithmetic: built with
 * -ffp-contract=off, IEEE-rounded divide and sqrt.  Tracks the un-contracted CPU oracle
 * (oracle/whitted_oracle.c) to within libm ulps; used for the tight parity tests and
 * selectable with clw_ext_set_strict / CLWRAP_STRICT=1.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "whitted_params.h"
#define WT_STRICT 1
#define WT_NS wt_strict
#define WT_LAUNCH_TRACE wt_strict_launch_trace
#define WT_LAUNCH_RAYGEN wt_strict_launch_raygen
#define WT_LAUNCH_SCHED wt_strict_launch_sched
#define WT_LAUNCH_UNIT wt_strict_launch_unit
#define WT_LAUNCH_UNIT_SCENE wt_strict_launch_unit_scene
#include "whitted_launch.inc"
