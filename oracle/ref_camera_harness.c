/*
 * ref_camera_harness.c -- TEST INFRASTRUCTURE ONLY (oracle/).  Not part of the product path.
 *
 * Calls the reference's OWN host camera code -- rinit_camera + rgen_perspective of
 * /root/reference/src/cpu_ray.c (cpu_ray.c:24-35, 42-106), compiled where it lies next to this file by
 * oracle/Makefile into oracle/_ref/libref_cpu_ray.so -- so that oracle/gen_golden.py can record its outputs as
 * fixtures (tests/golden/camera.npz).  The product's clw_host_perspective (csrc/host_camera.c) and the oracle's
 * wo_perspective are then compared with those bytes.
 *
 * rgen_perspective falls off its end without a return statement for the cameras it accepts (cpu_ray.c:106), so its
 * return value is only meaningful where it says `return false`; the harness reports the outputs, not that value.
 */
#include "cpu_ray.h"
#include <string.h>

/* out: im_corner[3], origin[3], up[3], right[3], w_factor, h_factor (14 floats, zero-filled first) */
void ref_perspective(const float origin[3], const float look[3], float fov, float focal, unsigned width, unsigned height,
                     float out[14]) {
    cl_float3 o, l, corner, cam_origin, up, right;
    cl_float wf = 0.0f, hf = 0.0f;
    memset(&corner, 0, sizeof corner); memset(&cam_origin, 0, sizeof cam_origin);
    memset(&up, 0, sizeof up); memset(&right, 0, sizeof right);
    o.x = origin[0]; o.y = origin[1]; o.z = origin[2];
    l.x = look[0]; l.y = look[1]; l.z = look[2];
    rcamera cam = rinit_camera(o, l, fov, focal);
    (void)rgen_perspective(&cam, &corner, &cam_origin, &up, &right, &wf, &hf, width, height);
    out[0] = corner.x; out[1] = corner.y; out[2] = corner.z;
    out[3] = cam_origin.x; out[4] = cam_origin.y; out[5] = cam_origin.z;
    out[6] = up.x; out[7] = up.y; out[8] = up.z;
    out[9] = right.x; out[10] = right.y; out[11] = right.z;
    out[12] = wf; out[13] = hf;
}
