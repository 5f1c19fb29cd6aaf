/*
 * ref_harness.cpp -- TEST INFRASTRUCTURE ONLY (oracle/). Not part of the product path.
 *
 * Builds the reference's OWN kernel sources for the host CPU, in place from
 * /root/reference (nothing is copied into the repo), through oracle/clc_host.h:
 *
 *   src/cl/raygen.cl      -> raygen()      (reference raygen.cl:5-25)
 *   src/cl/raytracing.cl  -> raytracer()   (reference raytracing.cl:14-195)
 *   src/cl/primitives.cl  -> helper functions, exported one by one below
 *
 * raytracing.cl hard-codes `#define MAX_DEPTH 15` (raytracing.cl:9); the build
 * recipe (oracle/Makefile) pipes that one line through sed into a temporary
 * file under /tmp so the same source can be instantiated at several depths
 * (SURVEY.md M3).  RAYTRACING_CL is the path of that temporary file.
 *
 * Output: oracle/_ref/libref_cl.so (git-ignored; travels to the GPU box as a
 * binary).  Used by tests (to pin oracle/whitted_oracle.c bit-for-bit), by
 * oracle/gen_golden.py (to generate tests/golden/) and optionally as a CPU
 * baseline.  Never linked or loaded by the product.
 */
#include "clc_host.h"
#include <vector>
#include <cstring>

thread_local size_t clc_global_id = 0;
thread_local unsigned long clc_oob_reads = 0;

/* global scope: types + primitives + the raygen kernel */
#include "src/cl/raygen.cl"

#ifndef RAYTRACING_CL
#error "RAYTRACING_CL must name the depth-parameterised temporary copy (see oracle/Makefile)"
#endif

/* one instantiation of the trace kernel per supported depth */
#define MAX_DEPTH 1
namespace ref_d1 {
#include RAYTRACING_CL
}
#undef MAX_DEPTH
#define MAX_DEPTH 2
namespace ref_d2 {
#include RAYTRACING_CL
}
#undef MAX_DEPTH
#define MAX_DEPTH 3
namespace ref_d3 {
#include RAYTRACING_CL
}
#undef MAX_DEPTH
#define MAX_DEPTH 4
namespace ref_d4 {
#include RAYTRACING_CL
}
#undef MAX_DEPTH
#define MAX_DEPTH 8
namespace ref_d8 {
#include RAYTRACING_CL
}
#undef MAX_DEPTH
#define MAX_DEPTH 15
namespace ref_d15 {
#include RAYTRACING_CL
}
#undef MAX_DEPTH

static_assert(sizeof(rmaterial) == 64, "rmaterial layout");
static_assert(sizeof(rsphere) == 96, "rsphere layout");
static_assert(sizeof(rplane) == 96, "rplane layout");
static_assert(sizeof(rlight) == 48, "rlight layout");
static_assert(sizeof(rray) == 64, "rray layout");

typedef void (*trace_fn)(rray*, rsphere*, rplane*, rlight*, uchar, uchar, uchar, uint,
                         image2d_array_t, image2d_array_t, uint*);

static trace_fn pick_depth(int depth) {
    switch (depth) {
        case 1: return ref_d1::raytracer;
        case 2: return ref_d2::raytracer;
        case 3: return ref_d3::raytracer;
        case 4: return ref_d4::raytracer;
        case 8: return ref_d8::raytracer;
        case 15: return ref_d15::raytracer;
        default: return nullptr;
    }
}

static inline float3 f3(const float* p) { return (float3){p[0], p[1], p[2]}; }

extern "C" {

/* camera = the six values rgen_perspective produces + width/height (raygen.cl:5-8).
 * Renders linear ids [id_begin, id_end) into out[id - id_begin].
 * Returns 0 on success, -1 for an unsupported depth. */
int ref_render(int depth, const float* im_corner, const float* cam_origin, const float* up,
               const float* right, float w_factor, float h_factor, uint pwidth, uint pheight,
               const void* spheres, uint ns, const void* planes, uint np, const void* lights,
               uint nl, const uint8_t* tex, int tex_w, int tex_h, int tex_layers,
               const uint8_t* sky, int sky_w, int sky_h, size_t id_begin, size_t id_end,
               uint* out, unsigned long* oob_reads) {
    trace_fn fn = pick_depth(depth);
    if (!fn) return -1;
    clc_image tex_im{tex, tex_w, tex_h, tex_layers};
    clc_image sky_im{sky, sky_w, sky_h, 1};
    const size_t chunk = 4096;
    std::vector<rray> rays(chunk);
    clc_oob_reads = 0;
    uint total = pwidth * pheight;
    for (size_t base = id_begin; base < id_end; base += chunk) {
        size_t end = base + chunk < id_end ? base + chunk : id_end;
        for (size_t id = base; id < end; id++) {
            clc_global_id = id;
            raygen(f3(im_corner), f3(cam_origin), f3(up), f3(right), w_factor, h_factor, pwidth,
                   pheight, rays.data() - base);
        }
        for (size_t id = base; id < end; id++) {
            clc_global_id = id;
            fn(rays.data() - base, (rsphere*)spheres, (rplane*)planes, (rlight*)lights, (uchar)ns,
               (uchar)np, (uchar)nl, total, &tex_im, &sky_im, out - id_begin);
        }
    }
    if (oob_reads) *oob_reads = clc_oob_reads;
    return 0;
}

/* raygen alone: writes 16 floats per id (the 64-byte rray record, raygen.cl:20-24) */
void ref_raygen(const float* im_corner, const float* cam_origin, const float* up,
                const float* right, float w_factor, float h_factor, uint pwidth, uint pheight,
                size_t id_begin, size_t id_end, void* rays_out) {
    rray* rays = (rray*)rays_out;
    memset(rays, 0, (id_end - id_begin) * sizeof(rray));
    for (size_t id = id_begin; id < id_end; id++) {
        clc_global_id = id;
        raygen(f3(im_corner), f3(cam_origin), f3(up), f3(right), w_factor, h_factor, pwidth,
               pheight, rays - id_begin);
    }
}

/* ---- per-function vectors (primitives.cl) ------------------------------- */
int ref_intersect_sphere(const float* o, const float* d, const float* c, float r, float* t) {
    rray ray; ray.origin = f3(o); ray.dir = f3(d);
    float3 cc = f3(c);
    float tt = 0.0f;
    bool hit = intersect_sphere(&ray, &cc, r, &tt);
    *t = hit ? tt : 0.0f;
    return hit;
}
int ref_intersect_plane(const float* o, const float* d, const float* n, const float* p0, float* t) {
    rray ray; ray.origin = f3(o); ray.dir = f3(d);
    float3 nn = f3(n), pp = f3(p0);
    float tt = 0.0f;
    bool hit = intersect_plane(&ray, &nn, &pp, &tt);
    *t = hit ? tt : 0.0f;
    return hit;
}
void ref_reflect(const float* i, const float* n, float* out) {
    float3 ii = f3(i), nn = f3(n);
    float3 r = reflect(&ii, &nn);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void ref_refract(float n1, float n2, const float* i, const float* n, float* out) {
    float3 ii = f3(i), nn = f3(n);
    float3 r = refract(n1, n2, &ii, &nn);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
float ref_schlick(float n1, float n2, const float* i, const float* n) {
    float3 ii = f3(i), nn = f3(n);
    return compute_schlick(n1, n2, &ii, &nn);
}
void ref_map_to_cube(const float* dir, uint face, int* uv) {
    float3 dd = f3(dir);
    int2 r = map_to_cube(&dd, face);
    uv[0] = r.x; uv[1] = r.y;
}
float ref_xorshift32(uint* state) {
    xorshift32_state s; s.x = *state;
    float r = xorshift32(&s);
    *state = s.x;
    return r;
}
int ref_euclidean_modulo(int a, int b) { return euclidean_modulo(a, b); }
void ref_plane_texture_pixel(const void* plane, const float* p, const uint8_t* tex, int w, int h,
                             int layers, float* rgb) {
    rplane pl = *(const rplane*)plane;
    clc_image im{tex, w, h, layers};
    float3 pp = f3(p);
    float3 c = plane_texture_pixel(&pl, &pp, &im);
    rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
}
float ref_shadow(const float* to, const float* from, const void* spheres, uint ns,
                 const void* planes, uint np) {
    float3 t = f3(to), f = f3(from);
    return testShadowPath(&t, &f, (rsphere*)spheres, (rplane*)planes, ns, np);
}
int ref_find_light(const float* o, const float* d, const void* lights, uint nl,
                   const void* spheres, uint ns, const void* planes, uint np, float* color) {
    rray ray; ray.origin = f3(o); ray.dir = f3(d);
    float3 c = (float3){0.0f, 0.0f, 0.0f};
    bool hit = findLightIntersection(&ray, (rlight*)lights, (rsphere*)spheres, (rplane*)planes,
                                     nl, ns, np, &c);
    color[0] = hit ? c.x : 0.0f; color[1] = hit ? c.y : 0.0f; color[2] = hit ? c.z : 0.0f;
    return hit;
}
/* out: point[3], normal[3], then the 16 words of the chosen rmaterial */
int ref_find_solid(const float* o, const float* d, const void* spheres, uint ns,
                   const void* planes, uint np, const uint8_t* tex, int w, int h, int layers,
                   float* point, float* normal, void* material64) {
    rray ray; ray.origin = f3(o); ray.dir = f3(d);
    clc_image im{tex, w, h, layers};
    float3 p, n; rmaterial m;
    memset(&m, 0, sizeof m);
    bool hit = findSolidIntersection(&ray, (rsphere*)spheres, (rplane*)planes, (uchar)ns,
                                     (uchar)np, &p, &n, &m, &im);
    if (hit) {
        point[0] = p.x; point[1] = p.y; point[2] = p.z;
        normal[0] = n.x; normal[1] = n.y; normal[2] = n.z;
        memcpy(material64, &m, 64);
    }
    return hit;
}

} /* extern "C" */
