/*
 * whitted_oracle.c -- TEST INFRASTRUCTURE ONLY (oracle/).  Not part of the product path.
 *
 * Scalar plain-C restatement of the reference's per-pixel Whitted trace, one
 * function per reference function, each citing the reference file:line it follows
 * (paths relative to /root/reference).  It is the checker the HIP path is compared
 * against, and (with OpenMP over pixels) the timed CPU baseline.
 *
 * PINNED BY (see tests/test_oracle_*.py):
 *   1. oracle/_ref/libref_cl.so -- the reference's own .cl sources compiled for the
 *      host in place (oracle/ref_harness.cpp).  Built with the same arithmetic flags
 *      (-O2 -ffp-contract=off) this file matches it BIT-FOR-BIT on whole frames and on
 *      per-function vectors; that comparison runs wherever _ref exists.
 *   2. tests/golden/ -- frames and per-function vectors generated from (1) by
 *      oracle/gen_golden.py and committed, so the pin also holds on the GPU box.
 *   3. the reference's only committed output, out/scene.png (800x600, depth 15, real
 *      assets): compared statistically where /root/reference exists.
 *
 * Arithmetic: fp32 evaluated strictly left to right exactly as the reference's
 * expressions parse, no FMA contraction, glibc sinf/cosf/powf/sqrtf; the two fp64
 * multiplies of the light sampling are kept (raytracing.cl:99-100).
 * Deliberate, documented choices where OpenCL leaves the result undefined:
 *   - (int)float out of int range / NaN: saturating, NaN -> 0 (what AMD hardware's
 *     v_cvt_i32_f32 does); every occurrence is counted (int_cast_oor).
 *   - image reads outside the image: clamped to the edge and counted (oob_reads).
 *   - plane_texture_pixel with no qualifying axis / map_to_cube with no qualifying
 *     face (uninitialised locals in the reference): zeros.
 */
#define _DEFAULT_SOURCE /* M_PI */
#include "whitted_oracle.h"
#include <math.h>
#include <float.h>
#include <string.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define EPSILON 0.001f                 /* primitives.cl:5 */
#define INV_PI_F 0.31830988618379067154f /* INVERSE_SQUARE_LIGHT = M_1_PI_F, primitives.cl:6 */
#define TRANSPARENT_THROUGH 0.8f       /* primitives.cl:7 */
/* The factor as a run-time value (same arithmetic: one float multiply): the reference's only committed output, out/scene.png,
 * was rendered by a version of the kernels in which a transparent sphere did NOT attenuate a shadow ray (factor 1.0) -- found
 * by a one-parameter search, tests/test_reference_fixture.py: 88.5 % of its pixels are reproduced with 0.8, 99.3 % with 1.0 -- so
 * the known-answer test of that fixture sets 1.0; everything else runs with the source's 0.8. */
/* ---- libm call log and override (wo_trace_libm): which sinf / cosf / powf inputs one pixel evaluates, and with whose results.  The device
 * libm (ocml) rounds a few per cent of all inputs one ulp away from glibc, and now and then such an ulp decides a shadow sample or an
 * 8-bit truncation; tests/test_gpu_parity.py::pin_strict_residual uses this to PROVE that nothing else separates the strict build from
 * this oracle: the pixel is re-traced here with the DEVICE's results substituted for glibc's and must come out as the GPU's pixel, bit for bit.
 * Rows of four floats {tag, a, b, result}: tag 0 = the xorshift pair of one soft-shadow sample (theta = fl32(2 pi a), phi = fl32(pi b));
 * tags 1..4 = sinf(phi), cosf(phi), sinf(theta), cosf(theta) with a = the angle; tag 5 = powf(a, b).  An override table has the same rows. */
static __thread float* wo_libm_log = 0;
static __thread uint32_t wo_libm_cap = 0, wo_libm_n = 0;
static __thread const float* wo_libm_ovr = 0;
static __thread uint32_t wo_libm_ovr_n = 0;
static inline void wo_libm_row(float tag, float a, float b, float r) {
    if (!wo_libm_log) return;
    if (wo_libm_n < wo_libm_cap) { float* q = wo_libm_log + 4 * (size_t)wo_libm_n; q[0] = tag; q[1] = a; q[2] = b; q[3] = r; }
    wo_libm_n++;
}
/* the value this build uses for libm call (tag, a, b): glibc's, or the override table's entry for exactly these input bits */
static inline float wo_libm_call(float tag, float a, float b, float glibc) {
    float r = glibc;
    for (uint32_t k = 0; k < wo_libm_ovr_n; k++) {
        const float* q = wo_libm_ovr + 4 * (size_t)k;
        if (q[0] == tag && !memcmp(&q[1], &a, 4) && !memcmp(&q[2], &b, 4)) { r = q[3]; break; }
    }
    wo_libm_row(tag, a, b, r);
    return r;
}
static float wo_through = TRANSPARENT_THROUGH;
void wo_set_transparent_through(float t) { wo_through = t; }
#define DEFAULT_N 1.0f                 /* raytracing.cl:7 */
#define SOFT_SHADOWS 2                 /* raytracing.cl:10 */

typedef struct { float x, y, z; } v3;

/* ---- decision margins (diagnostic build only: -DWO_MARGINS -> liboracle_margins.so) -------------------------------
 * Every comparison whose outcome selects a different code path or a different texel records how close its operands
 * were, as a relative margin |a - b| / (|a| + |b|); wo_render_margins returns the per-pixel minimum per site class.
 * A pixel with a margin near the rounding error (2^-24) is one where two correct evaluations of the reference may
 * legitimately disagree: tests/golden/masks.npz uses this for the ill-conditioned sphere discriminants that a
 * sub-pixel jitter of the camera cannot expose (far origins: b*b and 4ac agree to 7 digits, their difference is
 * rounding noise).  The plain build compiles all of this away. */
enum { WO_M_DISC = 0, WO_M_ROOT = 1, WO_M_PLANE = 2, WO_M_NEAREST = 3, WO_M_SHADOW_T = 4, WO_M_CAST = 5, WO_M_FACE = 6,
       WO_M_TIR = 7, WO_M_SITES = 8 };
#ifdef WO_MARGINS
static __thread float* wo_margin_row = 0;
static inline void wo_margin(int site, float a, float b) {
    if (!wo_margin_row) return;
    float den = fabsf(a) + fabsf(b);
    float m = den > 0.0f ? fabsf(a - b) / den : 1.0f;
    if (m == m && m < wo_margin_row[site]) wo_margin_row[site] = m;
}
#define WO_MARGIN(site, a, b) wo_margin(site, a, b)
#else
#define WO_MARGIN(site, a, b) ((void)0)
#endif

/* ---- wire structs (types.cl:4-59; host mirrors cpu_obj.h:10-48) -------- */
typedef struct {
    float rgb[4];
    float ambient, diffuse, specular;
    uint32_t shininess;
    uint32_t transperent, dielectric;
    float n, reflectivity;
    int32_t texture_id;
    float texture_scale;
    uint32_t pad_[2];
} w_material; /* 64 B */
typedef struct { float origin[4]; float radius; uint32_t pad_[3]; w_material material; } w_sphere; /* 96 B */
typedef struct { float normal[4]; float point[4]; w_material material; } w_plane;                  /* 96 B */
typedef struct { float origin[4]; float radius, intensity; uint32_t pad_[2]; float rgb[4]; } w_light; /* 48 B */

typedef char chk_mat[sizeof(w_material) == 64 ? 1 : -1];
typedef char chk_sph[sizeof(w_sphere) == 96 ? 1 : -1];
typedef char chk_pln[sizeof(w_plane) == 96 ? 1 : -1];
typedef char chk_lgt[sizeof(w_light) == 48 ? 1 : -1];

/* ---- vector helpers: the OpenCL built-ins as oracle/clc_host.h defines them ---- */
static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 ld3(const float* p) { return V(p[0], p[1], p[2]); }
static inline v3 add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mulv(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 muls(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 divs(v3 a, float s) { return V(a.x / s, a.y / s, a.z / s); }
static inline float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 cross(v3 a, v3 b) {
    return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline v3 normalize(v3 v) { return divs(v, sqrtf(dot(v, v))); }
static inline float distance(v3 a, v3 b) { v3 d = sub(a, b); return sqrtf(dot(d, d)); }
static inline float clamp01(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
/* OpenCL max() as clc_host.h spells it: a NaN second operand propagates */
static inline float fmaxf_like(float a, float b) { return a > b ? a : b; }

typedef struct { v3 origin, dir; } ray_t;

/* float -> int with the documented saturating semantics */
static inline int f2i(float f, wo_counters* c) {
    if (f != f) { if (c) c->int_cast_oor++; return 0; }
    if (f >= 2147483648.0f) { if (c) c->int_cast_oor++; return 2147483647; }
    if (f < -2147483648.0f) { if (c) c->int_cast_oor++; return (-2147483647 - 1); }
    WO_MARGIN(WO_M_CAST, f + fminf(f - floorf(f), ceilf(f) - f), f);
    return (int)f;
}

/* read_imagei on a raw RGBA8 layer stack (image built by opencl_wrap.c:212-332) */
static inline const uint8_t* texel(const uint8_t* base, int w, int h, int layers, int x, int y,
                                   int l, wo_counters* c) {
    if (x < 0 || y < 0 || l < 0 || x >= w || y >= h || l >= layers) {
        if (c) c->oob_reads++;
        x = x < 0 ? 0 : (x >= w ? w - 1 : x);
        y = y < 0 ? 0 : (y >= h ? h - 1 : y);
        l = l < 0 ? 0 : (l >= layers ? layers - 1 : l);
    }
    return base + 4 * ((size_t)l * (size_t)w * (size_t)h + (size_t)y * (size_t)w + (size_t)x);
}

/* ---- primitives.cl:14-109  map_to_cube --------------------------------- */
static void map_to_cube(v3 dir, uint32_t face_size, int32_t uv[2], wo_counters* c) {
    float x = dir.x, y = dir.y, z = dir.z;
    float ax = fabsf(x), ay = fabsf(y), az = fabsf(z);
    int xp = x > 0, yp = y > 0, zp = z > 0;
    float max_axis = 0.0f, uc = 0.0f, vc = 0.0f;
    uint32_t shift_u = 0, shift_v = 0;
    /* six independent ifs: later faces overwrite earlier ones on ties (primitives.cl:33-101) */
    if (xp && ax >= ay && ax >= az)  { max_axis = ax; uc = -z; vc = y;  shift_u = face_size * 2; shift_v = face_size * 1; }
    if (!xp && ax >= ay && ax >= az) { max_axis = ax; uc = z;  vc = y;  shift_v = face_size * 1; }
    if (yp && ay >= ax && ay >= az)  { max_axis = ay; uc = x;  vc = -z; shift_u = face_size; shift_v = face_size * 2; }
    if (!yp && ay >= ax && ay >= az) { max_axis = ay; uc = x;  vc = z;  shift_u = face_size; }
    if (zp && az >= ax && az >= ay)  { max_axis = az; uc = x;  vc = y;  shift_u = face_size; shift_v = face_size * 1; }
    if (!zp && az >= ax && az >= ay) { max_axis = az; uc = -x; vc = y;  shift_u = face_size * 3; shift_v = face_size * 1; }
    WO_MARGIN(WO_M_FACE, ax, ay); WO_MARGIN(WO_M_FACE, ax, az); WO_MARGIN(WO_M_FACE, ay, az);
    float fu = 0.5f * (uc / max_axis + 1.0f);
    float fv = 0.5f * (vc / max_axis + 1.0f);
    uv[0] = f2i((float)shift_u + fu * (float)face_size, c);
    uv[1] = f2i((float)shift_v + fv * (float)face_size, c);
}

/* ---- primitives.cl:116-125  xorshift32: returns a value in [0,4) ------- */
static inline float xorshift32(uint32_t* state) {
    uint32_t x = *state;
    x ^= x << 13;
    x ^= x >> 17;
    x ^= x << 5;
    *state = x;
    return ((float)x) / 2147483648.0f * 2.0f;
}

/* ---- primitives.cl:127-130  reflect ------------------------------------ */
static inline v3 reflect(v3 i, v3 n) {
    float cosI = -dot(n, i);
    return add(i, muls(n, 2 * cosI));
}

/* ---- primitives.cl:132-144  refract (NaN vector on total internal reflection) */
static inline v3 refract(float n1, float n2, v3 i, v3 nrm) {
    float n = n1 / n2;
    float cosI = -dot(nrm, i);
    float sinT2 = n * n * (1.0f - cosI * cosI);
    WO_MARGIN(WO_M_TIR, sinT2, 1.0f);
    if (sinT2 > 1.0f) return V(NAN, NAN, NAN);
    float cosT = sqrtf(1.0f - sinT2);
    return add(muls(i, n), muls(nrm, n * cosI - cosT));
}

/* ---- primitives.cl:146-160  compute_schlick ---------------------------- */
static inline float schlick(float n1, float n2, v3 i, v3 nrm) {
    float r0 = (n1 - n2) / (n1 + n2);
    r0 *= r0;
    float cosX = -dot(nrm, i);
    if (n1 > n2) {
        float n = n1 / n2;
        float sinT2 = n * n * (1.0f - cosX * cosX);
        if (sinT2 > 1.0f) return 1.0f;
        cosX = sqrtf(1.0f - sinT2);
    }
    float x = 1.0f - cosX;
    return r0 + (1.0f - r0) * x * x * x * x * x;
}

/* ---- primitives.cl:162-168  euclidean_modulo --------------------------- */
static inline int euclidean_modulo(int a, int b) {
    int m = a % b;
    if (m < 0) m = (b < 0) ? m - b : m + b;
    return m;
}

/* ---- primitives.cl:170-195  intersect_sphere --------------------------- */
static inline int intersect_sphere(const ray_t* r, v3 center, float radius, float* t,
                                   wo_counters* cnt) {
    if (cnt) cnt->sphere_tests++;
    v3 v = sub(r->origin, center);
    float a = dot(r->dir, r->dir);
    float b = dot(muls(v, 2), r->dir);
    float c = dot(v, v) - radius * radius;
    float D = b * b - 4 * a * c;
    WO_MARGIN(WO_M_DISC, b * b, 4 * a * c);
    if (D < 0) return 0;
    D = sqrtf(D);
    WO_MARGIN(WO_M_ROOT, -b, D);
    float t2 = ((-b - D) / (2 * a) < 0) ? (-b + D) / (2 * a) : (-b - D) / (2 * a);
    if (t2 <= 0) return 0;
    *t = t2;
    return 1;
}

/* ---- primitives.cl:197-215  intersect_plane (two-sided) ---------------- */
static inline int intersect_plane(const ray_t* r, v3 n, v3 p0, float* t, wo_counters* cnt) {
    if (cnt) cnt->plane_tests++;
    float b = dot(r->dir, n);
    if (b == 0) return 0;
    float t2 = dot(sub(p0, r->origin), n) / b;
    WO_MARGIN(WO_M_PLANE, dot(p0, n), dot(r->origin, n));
    WO_MARGIN(WO_M_PLANE, b + 1.0f, 1.0f);
    if (t2 <= 0) return 0;
    *t = t2;
    return 1;
}

/* ---- primitives.cl:217-259  plane_texture_pixel ------------------------ */
static v3 plane_texture_pixel(const w_plane* pl, v3 p, const wo_scene* sc, wo_counters* cnt) {
    const v3 axes[3] = {{1.0f, 0.0f, 0.0f}, {0.0f, 1.0f, 0.0f}, {0.0f, 0.0f, 1.0f}};
    v3 nrm = ld3(pl->normal);
    v3 b0 = V(0, 0, 0), b1 = V(0, 0, 0);
    for (int i = 0; i < 3; i++) {
        v3 cr = cross(axes[i], nrm);
        if (dot(V(1.0f, 1.0f, 1.0f), cr) == 0.0f) continue; /* component-sum test, kept */
        b0 = cr;
        b1 = cross(nrm, cr);
        break;
    }
    float ui = dot(b0, p) * pl->material.texture_scale;
    float vi = dot(b1, p) * pl->material.texture_scale;
    int tx = euclidean_modulo(f2i(ui, cnt), sc->tex_w);
    int ty = euclidean_modulo(f2i(vi, cnt), sc->tex_h);
    if (cnt) cnt->texel_fetches++;
    const uint8_t* px = texel(sc->tex, sc->tex_w, sc->tex_h, sc->tex_layers, tx, ty,
                              pl->material.texture_id, cnt);
    return V((float)px[0] / 255.0f, (float)px[1] / 255.0f, (float)px[2] / 255.0f);
}

/* ---- primitives.cl:262-318  findLightIntersection ---------------------- */
static int find_light(const ray_t* r, const wo_scene* sc, v3* color, wo_counters* cnt) {
    const w_light* L = (const w_light*)sc->lights;
    const w_sphere* S = (const w_sphere*)sc->spheres;
    const w_plane* P = (const w_plane*)sc->planes;
    int did = 0;
    float t = INFINITY;
    v3 col = V(0, 0, 0);
    if (cnt) cnt->light_probes++;
    for (uint32_t i = 0; i < sc->nl; i++) {
        float _t;
        int hit = intersect_sphere(r, ld3(L[i].origin), L[i].radius, &_t, cnt);
        if (!hit || _t >= t) continue;
        t = _t;
        v3 ip = add(r->origin, muls(r->dir, t));
        float d = distance(r->origin, ip);
        /* "(1/d*d)" parses as (1/d)*d -- not inverse-square; kept (primitives.cl:287) */
        col = muls(muls(muls(ld3(L[i].rgb), L[i].intensity), INV_PI_F), (1 / d * d));
        did = 1;
    }
    if (!did) return 0;
    for (uint32_t i = 0; i < sc->ns; i++) {
        float _t;
        int hit = intersect_sphere(r, ld3(S[i].origin), S[i].radius, &_t, cnt);
        if (hit && _t <= t && !S[i].material.transperent) return 0;
    }
    for (uint32_t i = 0; i < sc->np; i++) {
        float _t;
        int hit = intersect_plane(r, ld3(P[i].normal), ld3(P[i].point), &_t, cnt);
        if (hit && _t <= t) return 0;
    }
    *color = col;
    return 1;
}

/* ---- primitives.cl:322-394  findSolidIntersection ---------------------- */
static int find_solid(const ray_t* r, const wo_scene* sc, v3* point, v3* normal, w_material* mat,
                      wo_counters* cnt) {
    const w_sphere* S = (const w_sphere*)sc->spheres;
    const w_plane* P = (const w_plane*)sc->planes;
    int did = 0;
    float t = INFINITY;
    v3 nrm = V(0, 0, 0), ip = V(0, 0, 0);
    w_material m;
    memset(&m, 0, sizeof m);
    if (cnt) cnt->segments++;
    for (uint32_t i = 0; i < sc->ns; i++) {
        float _t;
        int hit = intersect_sphere(r, ld3(S[i].origin), S[i].radius, &_t, cnt);
        if (hit && t < INFINITY) WO_MARGIN(WO_M_NEAREST, _t, t);
        if (!hit || _t >= t) continue; /* strict: ties keep the earlier primitive */
        t = _t;
        ip = add(r->origin, muls(r->dir, t));
        nrm = normalize(sub(ip, ld3(S[i].origin)));
        ip = add(ip, muls(nrm, EPSILON));
        m = S[i].material;
        did = 1;
    }
    for (uint32_t i = 0; i < sc->np; i++) {
        float _t;
        int hit = intersect_plane(r, ld3(P[i].normal), ld3(P[i].point), &_t, cnt);
        if (hit && t < INFINITY) WO_MARGIN(WO_M_NEAREST, _t, t);
        if (!hit || _t >= t) continue;
        t = _t;
        ip = add(r->origin, muls(r->dir, t));
        nrm = ld3(P[i].normal); /* never flipped toward the ray */
        m = P[i].material;
        if (P[i].material.texture_id >= 0) {
            /* fetched for every improving plane, from the pre-offset point (primitives.cl:374-380) */
            v3 c = plane_texture_pixel(&P[i], ip, sc, cnt);
            m.rgb[0] = c.x; m.rgb[1] = c.y; m.rgb[2] = c.z;
        }
        ip = add(ip, muls(nrm, EPSILON));
        did = 1;
    }
    if (!did) return 0;
    *point = ip;
    *normal = nrm;
    *mat = m;
    return 1;
}

/* ---- primitives.cl:396-442  testShadowPath ----------------------------- */
static float shadow_path(v3 to, v3 from, const wo_scene* sc, wo_counters* cnt) {
    const w_sphere* S = (const w_sphere*)sc->spheres;
    const w_plane* P = (const w_plane*)sc->planes;
    ray_t r;
    r.origin = from;
    r.dir = normalize(sub(to, from));
    float t = distance(to, from);
    float opacity = 1.0f;
    if (cnt) cnt->shadow_rays++;
    for (uint32_t i = 0; i < sc->ns; i++) {
        float _t;
        int hit = intersect_sphere(&r, ld3(S[i].origin), S[i].radius, &_t, cnt);
        if (hit) WO_MARGIN(WO_M_SHADOW_T, _t, t);
        if (!hit || _t >= t) continue;
        if (S[i].material.transperent) { opacity *= wo_through; continue; }
        return 0.0f;
    }
    for (uint32_t i = 0; i < sc->np; i++) {
        float _t;
        int hit = intersect_plane(&r, ld3(P[i].normal), ld3(P[i].point), &_t, cnt);
        if (hit) WO_MARGIN(WO_M_SHADOW_T, _t, t);
        if (!hit || _t >= t) continue;
        return 0.0f;
    }
    return opacity;
}

/* ---- raygen.cl:5-25 ----------------------------------------------------- */
static inline ray_t raygen_one(const wo_camera* cam, uint64_t id64) {
    uint32_t id = (uint32_t)id64;
    float w = (float)(id % cam->width);
    float h = (float)(id / cam->width);
    v3 vec = sub(add(ld3(cam->im_corner), muls(muls(ld3(cam->right), cam->w_factor), w)),
                 muls(muls(ld3(cam->up), cam->h_factor), h));
    ray_t r;
    r.dir = normalize(vec);
    r.origin = ld3(cam->origin);
    return r;
}

/* ---- raytracing.cl:14-195  raytracer, one work-item --------------------- */
typedef struct { v3 origin, dir, rgb; int depth; } stack_ray;

static uint32_t trace_pixel(uint32_t id, ray_t primary, const wo_scene* sc, int max_depth,
                            float* out_rgb, wo_counters* cnt) {
    const w_light* L = (const w_light*)sc->lights;
    stack_ray rs[WO_MAX_DEPTH];
    float n_stack[WO_MAX_DEPTH];
    float f_stack[WO_MAX_DEPTH];
    uint32_t rand_state = id; /* id 0 is the xorshift fixed point (raytracing.cl:33) */
    uint32_t sp = 1;
    rs[0].origin = primary.origin; rs[0].dir = primary.dir; rs[0].rgb = V(0, 0, 0); rs[0].depth = 0;
    n_stack[0] = DEFAULT_N;
    f_stack[0] = 1.0f;

    while (sp > 0) {
        while (rs[sp - 1].depth < max_depth) {
            stack_ray* top = &rs[sp - 1];
            ray_t r; r.origin = top->origin; r.dir = top->dir;
            v3 ip, nrm; w_material m;
            v3 lc;
            if (find_light(&r, sc, &lc, cnt)) {                       /* :48-54 */
                top->rgb = add(top->rgb, muls(lc, f_stack[sp - 1]));
                break;
            }
            if (!find_solid(&r, sc, &ip, &nrm, &m, cnt)) {            /* :56-81 skybox */
                int32_t uv[2];
                map_to_cube(top->dir, (uint32_t)(sc->sky_w / 4), uv, cnt);
                if (cnt) cnt->sky_fetches++;
                const uint8_t* px = texel(sc->sky, sc->sky_w, sc->sky_h, 1, uv[0], sc->sky_h - uv[1], 0, cnt);
                v3 pf = V((float)px[0] / 255.0f, (float)px[1] / 255.0f, (float)px[2] / 255.0f);
                top->rgb = add(top->rgb, muls(pf, f_stack[sp - 1]));
                break;
            }
            if (cnt) cnt->shaded_hits++;
            /* ambient: f * material.rgb * ambient (:83-84) */
            top->rgb = add(top->rgb, muls(muls(ld3(m.rgb), f_stack[sp - 1]), m.ambient));

            for (uint32_t i = 0; i < sc->nl; i++) {                   /* :87-136 */
                v3 lo = ld3(L[i].origin);
                float soft = 0.0f;
                v3 shadow_dir = normalize(sub(lo, ip));
                for (int j = 0; j < SOFT_SHADOWS; j++) {
                    /* fp64 multiplies rounded to fp32 (M_PI is a double constant) */
                    const float u1 = xorshift32(&rand_state), u2 = xorshift32(&rand_state);
                    float theta = (float)(2 * M_PI * (double)u1);
                    float phi = (float)(M_PI * (double)u2);
                    float sp_ = sinf(phi), cp_ = cosf(phi), st_ = sinf(theta), ct_ = cosf(theta);
                    if (wo_libm_log || wo_libm_ovr) {
                        wo_libm_row(0.0f, u1, u2, 0.0f);
                        sp_ = wo_libm_call(1.0f, phi, 0.0f, sp_); cp_ = wo_libm_call(2.0f, phi, 0.0f, cp_);
                        st_ = wo_libm_call(3.0f, theta, 0.0f, st_); ct_ = wo_libm_call(4.0f, theta, 0.0f, ct_);
                    }
                    float x = L[i].radius * sp_ * ct_;
                    float y = L[i].radius * sp_ * st_;
                    float z = L[i].radius * cp_;
                    v3 sample = add(lo, V(x, y, z));
                    soft += shadow_path(sample, ip, sc, cnt);
                }
                float ssr = soft / (float)SOFT_SHADOWS;
                float d = distance(lo, ip);
                v3 light_rgb = divs(muls(muls(muls(ld3(L[i].rgb), L[i].intensity), INV_PI_F), 1.0f), d * d);
                light_rgb = muls(light_rgb, ssr);
                v3 v = normalize(sub(top->origin, ip));
                v3 h = normalize(add(v, shadow_dir));
                /* spec/diffuse do not multiply by the material colour (:129-135) */
                float spec_f = powf(fmaxf_like(0.0f, dot(nrm, h)), (float)m.shininess);
                if (wo_libm_log || wo_libm_ovr) spec_f = wo_libm_call(5.0f, fmaxf_like(0.0f, dot(nrm, h)), (float)m.shininess, spec_f);
                top->rgb = add(top->rgb, muls(muls(light_rgb, f_stack[sp - 1] * m.specular), spec_f));
                float diff_f = fmaxf_like(0.0f, dot(nrm, shadow_dir));
                top->rgb = add(top->rgb, muls(muls(light_rgb, f_stack[sp - 1] * m.diffuse), diff_f));
            }

            v3 incident = top->dir;                                   /* :139-159 */
            float n1 = n_stack[sp - 1];
            float n2 = m.n;
            n2 = (n1 == DEFAULT_N) ? n2 : DEFAULT_N;
            float reflect_amount = m.reflectivity;
            if (m.dielectric) {
                float fr = schlick(n1, n2, incident, nrm);
                reflect_amount = m.reflectivity + (1.0f - m.reflectivity) * fr;
            }
            float old_f = f_stack[sp - 1];
            f_stack[sp - 1] *= reflect_amount;
            top->dir = reflect(top->dir, nrm);
            top->origin = ip;
            top->depth++;

            if (m.transperent && sp < (uint32_t)max_depth && reflect_amount < 1.0f) { /* :161-179 */
                rs[sp] = rs[sp - 1];
                if (n1 < n2) {
                    rs[sp].origin = sub(rs[sp].origin, muls(nrm, 2 * EPSILON));
                } else {
                    nrm = muls(nrm, -1);
                }
                f_stack[sp] = old_f * (1.0f - reflect_amount);
                rs[sp].rgb = V(0, 0, 0);
                n_stack[sp] = n2;
                rs[sp].dir = refract(n1, n2, incident, nrm);
                if (rs[sp].dir.x != rs[sp].dir.x) { if (cnt) cnt->tir_drops++; continue; }
                sp++;
                if (cnt) { cnt->pushes++; if (sp > cnt->max_stack) cnt->max_stack = sp; }
            }
        }
        if (sp == 1) break;                                           /* :183-190 */
        rs[sp - 2].rgb = add(rs[sp - 2].rgb, rs[sp - 1].rgb);
        sp--;
    }
    if (out_rgb) { out_rgb[0] = rs[0].rgb.x; out_rgb[1] = rs[0].rgb.y; out_rgb[2] = rs[0].rgb.z; }
    /* truncating pack (:193-194) */
    float r = clamp01(rs[0].rgb.x) * 255.0f, g = clamp01(rs[0].rgb.y) * 255.0f, b = clamp01(rs[0].rgb.z) * 255.0f;
    return 0u << 24 | (uint32_t)r << 16 | (uint32_t)g << 8 | (uint32_t)b;
}

/* ---- drivers ------------------------------------------------------------ */
static void counters_add(wo_counters* d, const wo_counters* s) {
    d->segments += s->segments; d->light_probes += s->light_probes; d->shadow_rays += s->shadow_rays;
    d->sky_fetches += s->sky_fetches; d->texel_fetches += s->texel_fetches;
    d->sphere_tests += s->sphere_tests; d->plane_tests += s->plane_tests;
    d->shaded_hits += s->shaded_hits; d->pushes += s->pushes; d->tir_drops += s->tir_drops;
    d->int_cast_oor += s->int_cast_oor; d->oob_reads += s->oob_reads;
    if (s->max_stack > d->max_stack) d->max_stack = s->max_stack;
}

int wo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

static int render_impl(const wo_camera* cam, const float* rays16, const wo_scene* sc, int depth,
                       uint64_t id_begin, uint64_t id_end, uint32_t* out, float* out_rgb,
                       wo_counters* counters, int threads) {
    if (depth < 1 || depth > WO_MAX_DEPTH || id_end < id_begin) return -1;
    wo_counters total;
    memset(&total, 0, sizeof total);
    int64_t n = (int64_t)(id_end - id_begin);
#ifdef _OPENMP
    int nt = threads > 0 ? threads : omp_get_max_threads();
#pragma omp parallel num_threads(nt)
#endif
    {
        wo_counters local;
        memset(&local, 0, sizeof local);
        wo_counters* lc = counters ? &local : NULL;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 256)
#endif
        for (int64_t k = 0; k < n; k++) {
            uint64_t id = id_begin + (uint64_t)k;
            ray_t pr;
            if (rays16) {
                const float* r = rays16 + 16 * (size_t)k; /* rray: origin@0, dir@16 B (types.cl:50-57) */
                pr.origin = ld3(r);
                pr.dir = ld3(r + 4);
            } else {
                pr = raygen_one(cam, id);
            }
            out[k] = trace_pixel((uint32_t)id, pr, sc, depth, out_rgb ? out_rgb + 3 * (size_t)k : NULL, lc);
        }
#ifdef _OPENMP
#pragma omp critical
#endif
        { if (counters) counters_add(&total, &local); }
    }
    (void)threads;
    if (counters) *counters = total;
    return 0;
}

/* glibc's own results for the three libm functions of the path (tests/golden/libm_divergence.json pins where ocml differs) */
float wo_libm_sinf(float x) { return sinf(x); }
float wo_libm_cosf(float x) { return cosf(x); }
float wo_libm_powf(float x, float y) { return powf(x, y); }
/* the sample angles of raytracing.cl:99-100: fl32(2 pi u) / fl32(pi u), the products taken in fp64 */
float wo_libm_angle(float u, int full) { return (float)((full ? 2 * M_PI : M_PI) * (double)u); }

/* one pixel with its libm calls logged (rows of 4 floats, see wo_libm_row): returns the number of rows the pixel produced (<= cap stored) */
int wo_trace_libm(const wo_camera* cam, const wo_scene* sc, int depth, uint64_t id, uint32_t* out, float* log, uint32_t cap,
                  const float* overrides, uint32_t n_overrides) {
    if (depth < 1 || depth > WO_MAX_DEPTH) return -1;
    wo_libm_log = log; wo_libm_cap = cap; wo_libm_n = 0;
    wo_libm_ovr = overrides; wo_libm_ovr_n = overrides ? n_overrides : 0;
    *out = trace_pixel((uint32_t)id, raygen_one(cam, id), sc, depth, NULL, NULL);
    wo_libm_log = 0; wo_libm_ovr = 0; wo_libm_ovr_n = 0;
    return (int)wo_libm_n;
}

int wo_render(const wo_camera* cam, const wo_scene* sc, int depth, uint64_t id_begin,
              uint64_t id_end, uint32_t* out, float* out_rgb, wo_counters* counters, int threads) {
    return render_impl(cam, NULL, sc, depth, id_begin, id_end, out, out_rgb, counters, threads);
}

#ifdef WO_MARGINS
/* wo_render plus margins[WO_M_SITES * n]: per pixel the smallest relative margin seen at each site class (1 = none). */
int wo_render_margins(const wo_camera* cam, const wo_scene* sc, int depth, uint64_t id_begin, uint64_t id_end,
                      uint32_t* out, float* margins) {
    if (depth < 1 || depth > WO_MAX_DEPTH || id_end < id_begin) return -1;
    int64_t n = (int64_t)(id_end - id_begin);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 256)
#endif
    for (int64_t k = 0; k < n; k++) {
        float* row = margins + WO_M_SITES * (size_t)k;
        for (int s = 0; s < WO_M_SITES; s++) row[s] = 1.0f;
        wo_margin_row = row;
        out[k] = trace_pixel((uint32_t)(id_begin + (uint64_t)k), raygen_one(cam, id_begin + (uint64_t)k), sc, depth, NULL, NULL);
        wo_margin_row = 0;
    }
    return 0;
}
#endif

int wo_trace_rays(const float* rays16, const wo_scene* sc, int depth, uint64_t id_begin,
                  uint64_t id_end, uint32_t* out, float* out_rgb, wo_counters* counters,
                  int threads) {
    return render_impl(NULL, rays16, sc, depth, id_begin, id_end, out, out_rgb, counters, threads);
}

void wo_raygen(const wo_camera* cam, uint64_t id_begin, uint64_t id_end, float* rays16) {
    for (uint64_t id = id_begin; id < id_end; id++) {
        float* r = rays16 + 16 * (size_t)(id - id_begin);
        ray_t pr = raygen_one(cam, id);
        memset(r, 0, 64);
        r[0] = pr.origin.x; r[1] = pr.origin.y; r[2] = pr.origin.z;
        r[4] = pr.dir.x; r[5] = pr.dir.y; r[6] = pr.dir.z;
        /* rgb@32 = 0, depth@48 = 0 */
    }
}

/* ---- cpu_ray.c:8-18 normalize (double sqrt, float result) --------------- */
void wo_normalize3(const float v[3], float out[3]) {
    float number = 1 / sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    out[0] = v[0] * number; out[1] = v[1] * number; out[2] = v[2] * number;
}

/* ---- cpu_ray.c:42-106 rgen_perspective ---------------------------------- */
int wo_perspective(const float origin[3], const float dir[3], float fov, float focal,
                   uint32_t width, uint32_t height, wo_camera* out) {
    out->width = width; out->height = height;
    memcpy(out->origin, origin, 12);
    int is_180 = fov - 180.0f <= FLT_EPSILON && fov - 180.0f >= 0;
    if (is_180 || fov <= FLT_EPSILON || (dir[0] == 0.0f && dir[1] == 1.0f && dir[2] == 0.0f)) return 0;
    /* mixed float/double exactly as the reference: M_PI and tan are double, results stored to float */
    float half_radians_fov = (fov / 360.0f) * M_PI;
    float aspect_ratio = (float)height / (float)width;
    float fov_tan = tan(half_radians_fov);
    float image_width = fov_tan * focal * 2;
    float image_height = aspect_ratio * image_width;
    out->w_factor = image_width / width;
    out->h_factor = image_height / height;
    float fwd[3] = {dir[0] * -1.0f, dir[1] * -1.0f, dir[2] * -1.0f};
    const float top[3] = {0.0f, 1.0f, 0.0f};
    float right[3] = {top[1] * fwd[2] - top[2] * fwd[1], top[2] * fwd[0] - top[0] * fwd[2],
                      top[0] * fwd[1] - top[1] * fwd[0]};
    float up[3] = {fwd[1] * right[2] - fwd[2] * right[1], fwd[2] * right[0] - fwd[0] * right[2],
                   fwd[0] * right[1] - fwd[1] * right[0]};
    float center[3] = {-fwd[0] * focal, -fwd[1] * focal, -fwd[2] * focal};
    for (int k = 0; k < 3; k++) {
        out->right[k] = right[k];
        out->up[k] = up[k];
        out->im_corner[k] = center[k] - right[k] * image_width / 2 + up[k] * image_height / 2;
    }
    return 1;
}

/* ---- per-function entry points ------------------------------------------ */
int wo_intersect_sphere(const float o[3], const float d[3], const float c[3], float r, float* t) {
    ray_t ray = {ld3(o), ld3(d)};
    float tt = 0.0f;
    int hit = intersect_sphere(&ray, ld3(c), r, &tt, NULL);
    *t = hit ? tt : 0.0f;
    return hit;
}
int wo_intersect_plane(const float o[3], const float d[3], const float n[3], const float p0[3], float* t) {
    ray_t ray = {ld3(o), ld3(d)};
    float tt = 0.0f;
    int hit = intersect_plane(&ray, ld3(n), ld3(p0), &tt, NULL);
    *t = hit ? tt : 0.0f;
    return hit;
}
static void st3(float* o, v3 v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; }
void wo_reflect(const float i[3], const float n[3], float out[3]) { st3(out, reflect(ld3(i), ld3(n))); }
void wo_refract(float n1, float n2, const float i[3], const float n[3], float out[3]) {
    st3(out, refract(n1, n2, ld3(i), ld3(n)));
}
float wo_schlick(float n1, float n2, const float i[3], const float n[3]) { return schlick(n1, n2, ld3(i), ld3(n)); }
void wo_map_to_cube(const float dir[3], uint32_t face, int32_t uv[2]) { map_to_cube(ld3(dir), face, uv, NULL); }
float wo_xorshift32(uint32_t* state) { return xorshift32(state); }
int wo_euclidean_modulo(int a, int b) { return euclidean_modulo(a, b); }
void wo_plane_texture_pixel(const void* plane96, const float p[3], const uint8_t* tex, int w, int h,
                            int layers, float rgb[3]) {
    wo_scene sc;
    memset(&sc, 0, sizeof sc);
    sc.tex = tex; sc.tex_w = w; sc.tex_h = h; sc.tex_layers = layers;
    st3(rgb, plane_texture_pixel((const w_plane*)plane96, ld3(p), &sc, NULL));
}
float wo_shadow(const float to[3], const float from[3], const wo_scene* sc) {
    return shadow_path(ld3(to), ld3(from), sc, NULL);
}
int wo_find_light(const float o[3], const float d[3], const wo_scene* sc, float color[3]) {
    ray_t ray = {ld3(o), ld3(d)};
    v3 c = V(0, 0, 0);
    int hit = find_light(&ray, sc, &c, NULL);
    st3(color, hit ? c : V(0, 0, 0));
    return hit;
}
int wo_find_solid(const float o[3], const float d[3], const wo_scene* sc, float point[3],
                  float normal[3], void* material64) {
    ray_t ray = {ld3(o), ld3(d)};
    v3 p, n; w_material m;
    int hit = find_solid(&ray, sc, &p, &n, &m, NULL);
    if (hit) { st3(point, p); st3(normal, n); memcpy(material64, &m, 64); }
    return hit;
}
