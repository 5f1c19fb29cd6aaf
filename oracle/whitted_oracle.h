/*
 * whitted_oracle.h -- TEST INFRASTRUCTURE ONLY (oracle/).  Not part of the product path:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Plain-C restatement of the reference's per-pixel Whitted trace
 * (reference src/cl/raygen.cl, raytracing.cl, primitives.cl, types.cl and the
 * host-side rgen_perspective of src/cpu_ray.c).  See whitted_oracle.c for the
 * function-by-function citations and for how the restatement is pinned.
 */
#ifndef WHITTED_ORACLE_H
#define WHITTED_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The eight by-value raygen arguments (raygen.cl:5-8). */
typedef struct {
    float im_corner[3];
    float origin[3];
    float up[3];
    float right[3];
    float w_factor, h_factor;
    uint32_t width, height;
} wo_camera;

/* Raw scene arrays in the reference's wire layout (types.cl:4-59): 96-B spheres,
 * 96-B planes, 48-B lights; textures/skybox as RGBA8 layer stacks (opencl_wrap.c:212-332). */
typedef struct {
    const void* spheres; uint32_t ns;
    const void* planes;  uint32_t np;
    const void* lights;  uint32_t nl;
    const uint8_t* tex;  int32_t tex_w, tex_h, tex_layers;
    const uint8_t* sky;  int32_t sky_w, sky_h;
} wo_scene;

/* Work counters (rays = segments + shadow_rays). */
typedef struct {
    uint64_t segments;       /* findSolidIntersection calls (primary included)   */
    uint64_t light_probes;   /* findLightIntersection calls                       */
    uint64_t shadow_rays;    /* testShadowPath calls                              */
    uint64_t sky_fetches;    /* skybox texel reads                                */
    uint64_t texel_fetches;  /* plane texture texel reads                         */
    uint64_t sphere_tests;   /* intersect_sphere evaluations (lights included)    */
    uint64_t plane_tests;    /* intersect_plane evaluations                       */
    uint64_t shaded_hits;    /* Phong-shaded hits (each costs nl*2 shadow rays)   */
    uint64_t pushes;         /* refraction children pushed                        */
    uint64_t tir_drops;      /* children dropped by total internal reflection     */
    uint64_t int_cast_oor;   /* float->int conversions outside int range / NaN    */
    uint64_t oob_reads;      /* image reads outside the image (undefined in CL)   */
    uint64_t max_stack;      /* deepest DFS stack seen                            */
} wo_counters;

#define WO_MAX_DEPTH 64

/* Restatement of rgen_perspective (cpu_ray.c:42-106).  dir must already be
 * normalised (rinit_camera does that, cpu_ray.c:24-35; wo_normalize3 restates it).
 * Returns 0 for the inputs the reference rejects (cpu_ray.c:58-63), else 1. */
int wo_perspective(const float origin[3], const float dir[3], float fov, float focal,
                   uint32_t width, uint32_t height, wo_camera* out);
void wo_normalize3(const float v[3], float out[3]); /* cpu_ray.c:8-18 */

/* raygen kernel (raygen.cl:5-25): writes 16 floats per id (64-byte rray). */
void wo_raygen(const wo_camera* cam, uint64_t id_begin, uint64_t id_end, float* rays16);

/* raygen + raytracer (raytracing.cl:14-195) for linear ids [id_begin,id_end);
 * out[id-id_begin] = 0x00RRGGBB.  out_rgb (nullable) receives the un-clamped float
 * radiance, 3 floats per pixel.  threads<=0: all OpenMP threads.  depth in [1,64]. */
int wo_render(const wo_camera* cam, const wo_scene* sc, int depth, uint64_t id_begin,
              uint64_t id_end, uint32_t* out, float* out_rgb, wo_counters* counters,
              int threads);

/* raytracer alone from an explicit 64-byte-per-pixel ray buffer (unfused path). */
int wo_trace_rays(const float* rays16, const wo_scene* sc, int depth, uint64_t id_begin,
                  uint64_t id_end, uint32_t* out, float* out_rgb, wo_counters* counters,
                  int threads);

/* ---- per-function entry points (primitives.cl), for the vector tests ---- */
int   wo_intersect_sphere(const float o[3], const float d[3], const float c[3], float r, float* t);
int   wo_intersect_plane(const float o[3], const float d[3], const float n[3], const float p0[3], float* t);
void  wo_reflect(const float i[3], const float n[3], float out[3]);
void  wo_refract(float n1, float n2, const float i[3], const float n[3], float out[3]);
float wo_schlick(float n1, float n2, const float i[3], const float n[3]);
void  wo_map_to_cube(const float dir[3], uint32_t face, int32_t uv[2]);
float wo_xorshift32(uint32_t* state);
int   wo_euclidean_modulo(int a, int b);
void  wo_plane_texture_pixel(const void* plane96, const float p[3], const uint8_t* tex, int w,
                             int h, int layers, float rgb[3]);
float wo_shadow(const float to[3], const float from[3], const wo_scene* sc);
int   wo_find_light(const float o[3], const float d[3], const wo_scene* sc, float color[3]);
int   wo_find_solid(const float o[3], const float d[3], const wo_scene* sc, float point[3],
                    float normal[3], void* material64);
/* one pixel with its sinf / cosf / powf calls logged as rows {tag, a, b, result} and, optionally, their results taken from an
 * override table of the same rows (see whitted_oracle.c); returns the row count */
int   wo_trace_libm(const wo_camera* cam, const wo_scene* sc, int depth, uint64_t id, uint32_t* out, float* log, uint32_t cap,
                    const float* overrides, uint32_t n_overrides);
int   wo_num_threads(void);
/* factor a transparent sphere applies to a shadow ray (default 0.8f = primitives.cl:7); see whitted_oracle.c */
void  wo_set_transparent_through(float t);

#ifdef __cplusplus
}
#endif
#endif
