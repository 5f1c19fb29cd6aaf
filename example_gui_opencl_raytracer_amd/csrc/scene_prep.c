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
#include <math.h>
#include <stdlib.h>

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

/* ---- light / plane side table ------------------------------------------------------------------------
 * For every chunk of three lights c and every plane p one float4 {mu0, mu1, mu2, 1/|n|}: mu_k is the signed distance of
 * light 3c+k's centre from the plane, pulled towards zero by the light's radius and by a slack for every rounding on the
 * way (0 when the light sphere touches or straddles the plane, or does not exist).  |mu_k| > 0 therefore means: every
 * sample point on that light lies strictly on ONE side of the plane, at least |mu_k| away from it.  The trace kernel
 * uses it to skip, for a whole wavefront, plane tests of shadow rays that cannot come out "blocked" (the shading point
 * and the light are on the same side): whitted_trace.inc, wt_shadow_batch.  Pure work-skipping: the table never enters
 * any arithmetic a pixel depends on. */
size_t wprep_lpt_f4(uint32_t np, uint32_t nl) {
    size_t n = (size_t)((nl + 2) / 3) * np;
    return n <= 64 ? n : 0;                       /* big light x plane products: no table, no skipping */
}
void wprep_build_lpt(const uint8_t* planes, uint32_t np, const uint8_t* lights, uint32_t nl, float* out) {
    if (!wprep_lpt_f4(np, nl)) return;
    for (uint32_t c = 0; c < (nl + 2) / 3; c++)
        for (uint32_t p = 0; p < np; p++) {
            const uint8_t* pl = planes + 96 * (size_t)p;
            double n[3] = {ldf(pl, 0), ldf(pl, 4), ldf(pl, 8)}, p0[3] = {ldf(pl, 16), ldf(pl, 20), ldf(pl, 24)};
            double nn = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
            float* o = out + 4 * ((size_t)c * np + p);
            for (int k = 0; k < 3; k++) {
                o[k] = 0.0f;
                uint32_t l = 3 * c + (uint32_t)k;
                if (l >= nl || !(nn > 0.0) || !isfinite(nn)) continue;
                const uint8_t* li = lights + 48 * (size_t)l;
                double L[3] = {ldf(li, 0), ldf(li, 4), ldf(li, 8)}, r = fabs((double)ldf(li, 16));
                double sd = ((L[0] - p0[0]) * n[0] + (L[1] - p0[1]) * n[1] + (L[2] - p0[2]) * n[2]) / nn;
                double mag = fabs(L[0]) + fabs(L[1]) + fabs(L[2]) + fabs(p0[0]) + fabs(p0[1]) + fabs(p0[2]);
                double mu = fabs(sd) - r * (1.0 + 1e-5) - 1e-5 * mag - 1e-30;
                if (isfinite(mu) && mu > 0.0) o[k] = (float)(sd > 0 ? mu : -mu) * (1.0f - 1e-6f);
            }
            o[3] = (nn > 0.0 && isfinite(nn)) ? (float)(1.0 / nn) : 0.0f;
        }
}

/* ---- uniform grid ---------------------------------------------------------------------------------- */
static void sphere_cells(const uint8_t* s, const wprep_grid* g, int lo[3], int hi[3]) {
    float r = ldf(s, 16);
    for (int a = 0; a < 3; a++) {
        float c = ldf(s, 4 * (size_t)a);
        /* conservative: the box is widened by a margin far larger than the traversal's rounding error */
        float pad = 1e-3f * g->cell[a] + 1e-5f * (fabsf(c) + r) + g->reg_pad;
        int l = (int)floorf((c - r - pad - g->gmin[a]) * g->inv[a]);
        int h = (int)floorf((c + r + pad - g->gmin[a]) * g->inv[a]);
        lo[a] = l < 0 ? 0 : (l >= g->res[a] ? g->res[a] - 1 : l);
        hi[a] = h < 0 ? 0 : (h >= g->res[a] ? g->res[a] - 1 : h);
    }
}

size_t wprep_grid_plan(const uint8_t* spheres, uint32_t ns, float density, float reg_pad, wprep_grid* g) {
    g->reg_pad = (reg_pad > 0.0f && isfinite(reg_pad)) ? reg_pad * 1.001f + 1e-6f : 0.0f;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = 0; i < ns; i++) {
        const uint8_t* s = spheres + 96 * (size_t)i;
        float r = ldf(s, 16);
        for (int a = 0; a < 3; a++) {
            float c = ldf(s, 4 * (size_t)a);
            if (c - r < mn[a]) mn[a] = c - r;
            if (c + r > mx[a]) mx[a] = c + r;
        }
    }
    float ext[3], vol = 1.0f, big = 0.0f;
    for (int a = 0; a < 3; a++) { ext[a] = mx[a] - mn[a]; if (ext[a] > big) big = ext[a]; }
    if (!(big > 0.0f)) big = 1.0f;
    for (int a = 0; a < 3; a++) {
        /* every sphere lies strictly inside the grid bounds -- by a margin of ITS axis (sphere_cells pads by 1e-3 cell + 1e-5 of the
         * coordinate): a flat field of spheres (C4: 100 x 0.6 x 100) must not get a slab of empty space above it that every rising
         * shadow ray then walks through, cell by cell, testing the spheres below it (the margin used to be 1 % of the LARGEST extent:
         * 1.0 above a layer 0.6 thick; 9.5 -> 5 cells per shadow ray at C4) */
        float pad = 1e-2f * ext[a] + 2e-5f * fmaxf(fabsf(mn[a]), fabsf(mx[a])) + 1e-3f + 1.01f * g->reg_pad;
        mn[a] -= pad; mx[a] += pad; ext[a] = mx[a] - mn[a];
        vol *= ext[a];
    }
    /* cubic cells about one mean sphere diameter wide (finer cells make every sphere span several of them, coarser
     * ones put several spheres in a cell), kept between 1/8 and 4 spheres per cell on average; `density` scales
     * the side (tuning knob; the shim passes 1.6); 1..1024 cells per axis (10-bit cell boxes), <= 2^22 cells */
    double diam = 0.0;
    for (uint32_t i = 0; i < ns; i++) diam += 2.0 * ldf(spheres + 96 * (size_t)i, 16);
    diam /= (double)(ns ? ns : 1);
    float side_lo = cbrtf(vol * 0.125f / (float)(ns ? ns : 1)), side_hi = cbrtf(vol * 4.0f / (float)(ns ? ns : 1));
    float side = (float)diam * density;
    if (!(side > side_lo)) side = side_lo;
    if (side > side_hi) side = side_hi;
    uint64_t total = 1;
    for (int a = 0; a < 3; a++) {
        int n = (int)(ext[a] / side + 0.5f);
        if (n < 1) n = 1;
        if (n > 1024) n = 1024;
        g->res[a] = n;
        total *= (uint64_t)n;
    }
    while (total > (1u << 22)) {                /* coarsen the finest axes until the table fits */
        int a = 0;
        for (int k = 1; k < 3; k++) if (g->res[k] > g->res[a]) a = k;
        total /= (uint64_t)g->res[a];
        g->res[a] = (g->res[a] + 1) / 2;
        total *= (uint64_t)g->res[a];
    }
    for (int a = 0; a < 3; a++) {
        g->gmin[a] = mn[a];
        g->cell[a] = ext[a] / (float)g->res[a];
        g->inv[a] = 1.0f / g->cell[a];
    }
    g->ncells = (uint32_t)total;
    size_t pairs = 0;
    for (uint32_t i = 0; i < ns; i++) {
        int lo[3], hi[3];
        sphere_cells(spheres + 96 * (size_t)i, g, lo, hi);
        pairs += (size_t)(hi[0] - lo[0] + 1) * (size_t)(hi[1] - lo[1] + 1) * (size_t)(hi[2] - lo[2] + 1);
    }
    return pairs;
}

void wprep_grid_fill(const uint8_t* spheres, uint32_t ns, const wprep_grid* g, uint32_t* start, uint32_t* items,
                     uint32_t* box) {
    memset(start, 0, sizeof(uint32_t) * ((size_t)g->ncells + 1));
    for (uint32_t i = 0; i < ns; i++) {         /* pass 1: counts (shifted by one for the prefix sum) */
        int lo[3], hi[3];
        sphere_cells(spheres + 96 * (size_t)i, g, lo, hi);
        box[2 * i] = (uint32_t)lo[0] | (uint32_t)lo[1] << 10 | (uint32_t)lo[2] << 20;
        box[2 * i + 1] = (uint32_t)hi[0] | (uint32_t)hi[1] << 10 | (uint32_t)hi[2] << 20;
        for (int z = lo[2]; z <= hi[2]; z++)
            for (int y = lo[1]; y <= hi[1]; y++)
                for (int x = lo[0]; x <= hi[0]; x++)
                    start[((size_t)z * g->res[1] + y) * g->res[0] + x + 1]++;
    }
    for (uint32_t c = 0; c < g->ncells; c++) start[c + 1] += start[c];
    uint32_t* cur = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)g->ncells);
    memcpy(cur, start, sizeof(uint32_t) * (size_t)g->ncells);
    for (uint32_t i = 0; i < ns; i++) {         /* pass 2: ascending sphere index inside every cell */
        int lo[3], hi[3];
        sphere_cells(spheres + 96 * (size_t)i, g, lo, hi);
        for (int z = lo[2]; z <= hi[2]; z++)
            for (int y = lo[1]; y <= hi[1]; y++)
                for (int x = lo[0]; x <= hi[0]; x++)
                    items[cur[((size_t)z * g->res[1] + y) * g->res[0] + x]++] = i;
    }
    free(cur);
}
