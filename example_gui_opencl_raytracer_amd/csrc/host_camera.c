/*
 * host_camera.c -- camera -> the six raygen scalars/vectors, for callers that do not link
 * the reference's src/cpu_ray.c (the bench driver, bench.py, the tests).
 *
 * Same results as the reference's rinit_camera + rgen_perspective
 * (reference src/cpu_ray.c:8-35, 42-106): the look direction is normalised with a
 * double-precision sqrt, the half-angle and tan() go through double and are stored to
 * float, `right`/`up` are not re-normalised, and the corner is
 * dir*focal - right*image_w/2 + up*image_h/2.  Built with -ffp-contract=off.
 */
#include "../../include/hip_wrap_ext.h"
#include <float.h>
#include <math.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

int clw_host_perspective(const float origin[3], const float look[3], float fov, float focal,
                         uint32_t width, uint32_t height, clw_camera* out) {
    /* rinit_camera: dir = look / |look| */
    float inv = 1 / sqrt(look[0] * look[0] + look[1] * look[1] + look[2] * look[2]);
    float dir[3] = {look[0] * inv, look[1] * inv, look[2] * inv};

    out->width = width;
    out->height = height;
    for (int k = 0; k < 3; k++) out->origin[k] = origin[k];

    int is_180 = fov - 180.0f <= FLT_EPSILON && fov - 180.0f >= 0;
    if (is_180 || fov <= FLT_EPSILON || (dir[0] == 0.0f && dir[1] == 1.0f && dir[2] == 0.0f)) return 0;

    float half_fov = (fov / 360.0f) * M_PI;
    float aspect = (float)height / (float)width;
    float fov_tan = tan(half_fov);
    float image_w = fov_tan * focal * 2;
    float image_h = aspect * image_w;
    out->w_factor = image_w / width;
    out->h_factor = image_h / height;

    float fwd[3] = {dir[0] * -1.0f, dir[1] * -1.0f, dir[2] * -1.0f};
    /* right = (0,1,0) x forward, up = forward x right, written out like the reference */
    float right[3] = {1.0f * fwd[2] - 0.0f * fwd[1], 0.0f * fwd[0] - 0.0f * fwd[2], 0.0f * fwd[1] - 1.0f * fwd[0]};
    float up[3] = {fwd[1] * right[2] - fwd[2] * right[1], fwd[2] * right[0] - fwd[0] * right[2],
                   fwd[0] * right[1] - fwd[1] * right[0]};
    for (int k = 0; k < 3; k++) {
        float centre = -fwd[k] * focal;
        out->right[k] = right[k];
        out->up[k] = up[k];
        out->im_corner[k] = centre - right[k] * image_w / 2 + up[k] * image_h / 2;
    }
    return 1;
}
