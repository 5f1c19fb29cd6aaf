/*
 * scene_prep.c -- raw wire structs -> prepared geometry stream (see whitted_params.h).
 *
 * Built with gcc -ffp-contract=off: every value below is an fp32 expression the
 * reference evaluates per test or per hit, hoisted to once per scene with the same
 * operations in the same order:
 *   r*r                               reference primitives.cl:175   (intersect_sphere)
 *   (rgb * intensity) * (1/pi)        primitives.cl:287, raytracing.cl:118
 *   tangent basis b0, b1 of a plane   primitives.cl:226-236         (plane_texture_pixel)
 * Struct offsets: reference src/cl/types.cl:4-59.
 */
#include "scene_prep.h"
#include <string.h>

#define INV_PI_F 0.31830988618379067154f

static float ldf(const uint8_t* p, size_t off) { float f; memcpy(&f, p + off, 4); return f; }
static uint32_t ldu(const uint8_t* p, size_t off) { uint32_t u; memcpy(&u, p + off, 4); return u; }

static void cross(const float a[3], const float b[3], float out[3]) {
    out[0] = a[1] * b[2] - a[2] * b[1];
    out[1] = a[2] * b[0] - a[0] * b[2];
    out[2] = a[0] * b[1] - a[1] * b[0];
}

size_t wprep_geom_f4(uint32_t ns, uint32_t np, uint32_t nl) { return (size_t)ns + 2 * (size_t)np + 2 * (size_t)nl; }

void wprep_build(const uint8_t* spheres, uint32_t ns, const uint8_t* planes, uint32_t np,
                 const uint8_t* lights, uint32_t nl, float* geom, float* ptex) {
    float* g = geom;
    for (uint32_t i = 0; i < ns; i++, g += 4) {
        const uint8_t* s = spheres + 96 * (size_t)i;
        float r = ldf(s, 16);
        float r2 = r * r;
        uint32_t transparent = ldu(s, 32 + 32); /* material.transperent */
        uint32_t bits;
        memcpy(&bits, &r2, 4);
        bits &= 0x7FFFFFFFu;
        if (transparent) bits |= 0x80000000u;    /* sign bit of r*r carries the flag */
        g[0] = ldf(s, 0); g[1] = ldf(s, 4); g[2] = ldf(s, 8);
        memcpy(&g[3], &bits, 4);
    }
    for (uint32_t i = 0; i < np; i++, g += 8) {
        const uint8_t* p = planes + 96 * (size_t)i;
        float n[3] = {ldf(p, 0), ldf(p, 4), ldf(p, 8)};
        int32_t texture_id = (int32_t)ldu(p, 32 + 48);
        float scale = ldf(p, 32 + 52);
        g[0] = n[0]; g[1] = n[1]; g[2] = n[2]; g[3] = texture_id >= 0 ? 1.0f : 0.0f;
        g[4] = ldf(p, 16); g[5] = ldf(p, 20); g[6] = ldf(p, 24); g[7] = 0.0f;

        /* first axis e in {X,Y,Z} with (e x n).x + (e x n).y + (e x n).z != 0 (component-sum test
         * kept as is); b0 = e x n, b1 = n x b0.  No axis qualifies -> zeros (uninitialised in the
         * reference). */
        static const float axes[3][3] = {{1.0f, 0.0f, 0.0f}, {0.0f, 1.0f, 0.0f}, {0.0f, 0.0f, 1.0f}};
        float b0[3] = {0, 0, 0}, b1[3] = {0, 0, 0};
        for (int a = 0; a < 3; a++) {
            float cr[3];
            cross(axes[a], n, cr);
            float sum = 1.0f * cr[0] + 1.0f * cr[1] + 1.0f * cr[2];
            if (sum == 0.0f) continue;
            memcpy(b0, cr, sizeof b0);
            cross(n, cr, b1);
            break;
        }
        float* t = ptex + 8 * (size_t)i;
        t[0] = b0[0]; t[1] = b0[1]; t[2] = b0[2]; t[3] = scale;
        t[4] = b1[0]; t[5] = b1[1]; t[6] = b1[2];
        memcpy(&t[7], &texture_id, 4);
    }
    for (uint32_t i = 0; i < nl; i++, g += 8) {
        const uint8_t* l = lights + 48 * (size_t)i;
        float r = ldf(l, 16), intensity = ldf(l, 20);
        g[0] = ldf(l, 0); g[1] = ldf(l, 4); g[2] = ldf(l, 8); g[3] = r * r;
        for (int k = 0; k < 3; k++) {
            float c = ldf(l, 32 + 4 * (size_t)k);
            g[4 + k] = (c * intensity) * INV_PI_F;
        }
        g[7] = r;
    }
}
