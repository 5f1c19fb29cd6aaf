/*
 * hip_wrap_ext.h -- entry points the reference API has no slot for.
 *
 * The reference bakes these into compile-time macros or has no notion of them:
 *   depth           #define MAX_DEPTH 15            (reference raytracing.cl:9)
 *   frame size      #define WIDTH/HEIGHT 800/600    (raypng.c:8-9, rayinteractive.c:13-14)
 *   counts          one byte each                   (raytracing.cl:17, cpu_obj.c:62-68)
 *   device / queue  platform 0, device 0, one queue (opencl_wrap.c:26-34, 118-119)
 * They are needed by the bench driver, the row-strip multi-GPU mode and the parity
 * tests.  Everything here is optional: a caller that only uses opencl_wrap.h gets the
 * reference's behaviour.  Plain C ABI: pointers and sizes only.
 *
 * Environment variables read once by cl_wrap_init (same meaning as the setters):
 *   CLWRAP_DEPTH=<1..32>   CLWRAP_STRICT=<0|1>   CLWRAP_FUSE=<0|1>   CLWRAP_DEVICE=<ordinal>   CLWRAP_PIPELINE=<0|1>   CLWRAP_THROUGH=<float>
 * Tuning / experiment knobs (defaults are the measured optima): CLWRAP_GRID_MIN, CLWRAP_GRID_DENSITY (uniform grid),
 *   CLWRAP_OCC_TILES_PER_DEPTH (deep launches of >= this x depth tiles take the high-occupancy kernel flavour),
 *   CLWRAP_TIMING_EVERY, CLWRAP_VARIANT (bit mask of clw_ext_set_variant).
 */
#ifndef HIP_WRAP_EXT_H
#define HIP_WRAP_EXT_H
#include "opencl_wrap.h"
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CLW_MAX_DEPTH 32

/* Trace depth (reference MAX_DEPTH).  Default 15.  Errors (print + exit(1)) outside [1, 32]. */
void clw_ext_set_depth(cl_wrap* wrap, int depth);
int  clw_ext_get_depth(const cl_wrap* wrap);

/* Arithmetic mode of the trace kernel: 0 = fast (FMA contraction, native rcp/rsq/sqrt --
 * the envelope an OpenCL device build of the reference is allowed), 1 = strict (no
 * contraction, IEEE divide/sqrt: tracks the un-contracted oracle). Default 0. */
void clw_ext_set_strict(cl_wrap* wrap, int strict);

/* 1 (default): a "raygen" launch latches its eight by-value arguments and the trace
 * kernel synthesises primary rays in registers; the 64 B/pixel ray buffer is only
 * materialised if somebody reads it back.  0: run the two kernels as the reference does. */
void clw_ext_set_fuse(cl_wrap* wrap, int fuse);

/* Row strips / sub-ranges: work-item i of the next launches has global id first_id + i
 * (used for id %% width, id / width and the RNG seed; reference raygen.cl:13-14,
 * raytracing.cl:33).  Buffers are indexed by i.  Default 0. */
void clw_ext_set_id_offset(cl_wrap* wrap, uint64_t first_id);

/* Interleaved row bands (balanced multi-GPU sharding): with stride S > 1 the next launches own
 * every S-th 8-row band of the frame, starting at band `phase`: local row y is global row
 * ((y / 8) * S + phase) * 8 + y % 8.  Needs height % (8 * S) == 0 and id offset 0.
 * stride 1 (default) = one contiguous range. */
void clw_ext_set_row_bands(cl_wrap* wrap, uint32_t stride, uint32_t phase);

/* 1: cl_wrap_output returns without waiting for the kernel (no host read-back may be
 * requested in that mode); clw_ext_sync waits.  Default 0 = the reference's behaviour. */
void clw_ext_set_async(cl_wrap* wrap, int async);
void clw_ext_sync(cl_wrap* wrap);

/* Launch on a caller-owned hipStream_t (e.g. torch's current stream) instead of the
 * wrapper's own stream.  NULL restores the wrapper's stream. */
void clw_ext_set_stream(cl_wrap* wrap, void* hip_stream);

/* Per-kernel device timing from hipEvents recorded around every launch on the launch
 * stream, from the first clw_ext_timing_reset on (a wrapper nobody asks for timings records no events).
 * reset clears the log; get waits for the logged launches and returns how many
 * there were and their summed duration in milliseconds. */
void clw_ext_timing_reset(cl_wrap* wrap);
void clw_ext_timing_get(cl_wrap* wrap, cl_uint kernel_id, uint32_t* launches, double* total_ms);
/* Pipelined read-back (default on; env CLWRAP_PIPELINE=0): a blocking cl_wrap_output that renders a frame of >= 16 MB
 * at depth <= 4 and reads that same framebuffer back is executed as four strips, strip c being copied to the host while
 * strip c+1 renders (3840x2160: 940 -> 1 340 frames/s).  Same pixels, same semantics -- the call returns with the whole
 * frame in host memory.  Smaller frames and deep launches keep the single launch + single copy (measured: no gain / a
 * loss there). */
void clw_ext_set_pipeline(cl_wrap* wrap, int on);

/* Record the events around every n-th launch only (default 1 = every launch; env CLWRAP_TIMING_EVERY; 0 = none until the next
 * clw_ext_timing_reset / clw_ext_set_timing_every).  An event record between two kernels of a stream keeps the second from being
 * dispatched while the first drains: at 1920x1080 depth 4 that is 5 us per 125-us frame, so bench.py times its throughput loops
 * with no events at all and measures the kernel in one extra pass with events around every launch. */
void clw_ext_set_timing_every(cl_wrap* wrap, uint32_t n);

/* Texture / skybox layer stack from memory instead of PNG files: `rgba` is
 * layers*h*w*4 bytes, layer-major, row-major (the image cl_wrap_load_images builds,
 * opencl_wrap.c:212-332). */
void clw_ext_load_images_raw(cl_wrap* wrap, cl_uint kernel_id, cl_uint arg_id, const uint8_t* rgba,
                             uint32_t width, uint32_t height, uint32_t layers);

/* Register caller-owned device memory as buffer argument `arg_id` (not freed by release). */
void clw_ext_bind_device_buffer(cl_wrap* wrap, cl_uint kernel_id, cl_uint arg_id, void* device_ptr,
                                size_t size);
/* The scene arrays (raytracer args 1-3) are SNAPSHOTTED when first traced: the shim derives its prepared geometry
 * (and, for big scenes, the uniform grid) from them once per (buffers, counts).  A caller that rewrites a bound or
 * uploaded scene buffer in place calls this to have the next launch prepare the scene again. */
void clw_ext_invalidate_scene(cl_wrap* wrap);
/* Device address of a buffer argument (allocates a lazily created buffer). */
void* clw_ext_device_ptr(cl_wrap* wrap, cl_uint kernel_id, cl_uint arg_id);

/* Optional float radiance output of the trace kernel: 3 floats per work-item, written
 * before the 8-bit pack (reference raytracing.cl:193).  NULL disables. */
void clw_ext_set_debug_rgb(cl_wrap* wrap, void* device_ptr_f32x3);

/* Work counters of the trace kernel.  enable=1 selects the counting build of the kernel
 * for subsequent launches (slower); read returns and clears
 *   out[0] path segments  out[1] shadow rays  out[2] light probes  out[3] skybox fetches
 *   out[4] texel fetches  out[5] refraction pushes  out[6] lane-iterations  out[7] wave-iterations*64
 * (rays = out[0] + out[1]; out[6]/out[7] = SIMD lane utilisation). */
void clw_ext_enable_counters(cl_wrap* wrap, int enable);
void clw_ext_read_counters(cl_wrap* wrap, uint64_t out[8]);
/* The same with the later words: out[8] = shadow rays really TRACED.  out[1] counts every shadow ray the reference
 * would cast; those of a surface whose specular and diffuse coefficients are both zero (glass)
 * contribute exactly +0, so the kernel draws their random numbers but does not trace them -- out[8] leaves them
 * out.  n <= 32 words are returned (the rest reads 0), and the device block is cleared.  Words 16.. are
 * only written by the diagnostic stamp build of the kernel (tools/stamp_phases.py). */
void clw_ext_read_counters_ex(cl_wrap* wrap, uint64_t* out, uint32_t n);

/* The factor a transparent sphere applies to a shadow ray that passes through it: 0.8f by default, the reference's
 * TRANSPERENT_THROUGH (primitives.cl:7, :419).  A knob because the reference's only committed output, out/scene.png, was rendered by
 * a version of its kernels without that attenuation: with 1.0 the unchanged raypng.c driver reproduces that image (tests/
 * test_gpu_reference_fixture.py); also CLWRAP_THROUGH=<float> in the environment of cl_wrap_init. */
void clw_ext_set_shadow_through(cl_wrap* wrap, float factor);

/* Uniform grid over the spheres (default on; built for scenes with more than 256 spheres): rays test only the
 * spheres registered in the cells they cross instead of all of them.  Same arithmetic per test, same nearest
 * hit and same shadow factor as the reference's linear scan; 0 forces the linear scan. */
void clw_ext_set_grid(cl_wrap* wrap, int on);

/* Cost-sorted tile dispatch (default on): every 8x8 tile reports its cost, and the next frame serves each
 * XCD's tiles heaviest-first, so the expensive refraction tiles no longer end up in the tail of the launch.
 * Pure scheduling: the image is bit-identical either way. */
void clw_ext_set_tile_sched(cl_wrap* wrap, int on);

/* The per-tile cost table of the last tiled launch (row-major 8x8 tiles; cost = loop iterations of the tile's most expensive
 * pixel, +3 per shaded hit; for scenes on the uniform grid: the lifetime of the tile's wavefront in units of 256 cycles).  Returns the number of tiles; copies them if `capacity` suffices. */
uint32_t clw_ext_read_tile_costs(cl_wrap* wrap, uint32_t* out, uint32_t capacity);

/* Runs ONE device helper of the trace kernel over `n` input rows (host arrays; rows of `stride_in` / `stride_out`
 * floats) in the current arithmetic mode -- function-level parity tests against the reference's own functions.
 * op: 0 intersect_sphere {o,d,c,r -> hit,t}  1 intersect_plane {o,d,n,p0 -> hit,t}  2 reflect {i,n -> r}
 *     3 refract {n1,n2,i,n -> ok,r}  4 compute_schlick {n1,n2,i,n -> f}  5 map_to_cube {dir -> u,v bits; aux = face}
 *     6 xorshift32 {state bits -> state bits, value}  7 euclidean_modulo {a,b bits -> m bits}
 *     8 sin/cos of the sampling angle {u -> s,c of fl32(2 pi u) (aux 1) or fl32(pi u) (aux 0)}  9 pow {x,y -> x^y}  10 normalize {v -> unit, length}. */
void clw_ext_unit(cl_wrap* wrap, int op, const float* in, uint32_t stride_in, float* out, uint32_t stride_out,
                  uint32_t n, uint32_t aux);

/* The same for the helpers that need the SCENE: runs on the scene bound to raytracer kernel `kernel_id` (its args 1-6,
 * 8, 9), through the code the trace kernel itself runs (its hit phase and its batched shadow traversal).
 * op: 0 hit phase {o,d -> lit, light rgb[3], solid hit, point[3], normal[3], material rgb[3], ambient, diffuse, specular,
 *       shininess, transparent, dielectric, n, reflectivity}  (findLightIntersection + findSolidIntersection,
 *       reference primitives.cl:262-318, 322-394; 22 floats out)
 *     1 testShadowPath {to, from -> factor}  (primitives.cl:396-442)
 *     2 plane_texture_pixel {b0[3], scale, b1[3], texture id bits, p[3] -> rgb}  (primitives.cl:217-259; the first eight
 *       floats are the plane's prepared basis row; stride_in must be a multiple of 4). */
void clw_ext_unit_scene(cl_wrap* wrap, cl_uint kernel_id, int op, const float* in, uint32_t stride_in, float* out,
                        uint32_t stride_out, uint32_t n);

/* Kernel build variant for A/B measurements and equivalence tests (same image in every variant); 0 = default.  Bits:
 * 1 geometry from global memory instead of LDS, 2 linear work-item ids instead of 8x8 tiles, 4 no cost-sorted tile
 * order, 8 no uniform grid, 16 no tree-parallel tail (deep launches run their per-lane loop to the end), 64 never the high-occupancy flavour of the deep build,
 * 128 no light / plane side table (every shadow ray tests every plane), 256 no visibility classes (every needed shadow ray is traced),
 * 4096 heavy tiles of deep launches are not split over several wavefronts,
 * 2048 deep launches always carry the full-depth (31-parent) scratch stack instead of one sized for their depth,
 * 1024 (with clw_ext_enable_counters) VERIFICATION of the visibility classes: lights are classified AND traced, counter word 9 =
 * lights classified, word 28 = lights whose traced factors differ from their class's (must read 0),
 * 512 DIAGNOSTIC builds only (-DWT_TIMELINE=1, tools/timeline.py): the tile-cost buffer receives when each tile's wave ran inside the launch
 * (CLWRAP_TIMELINE_SHIFT = tick of 10 ns << shift; CLWRAP_TIMELINE_EDGES = 1 / 2: its prologue and epilogue instead); no effect otherwise. */
void clw_ext_set_variant(cl_wrap* wrap, int variant);

/* The tree-parallel tail of deep launches (depth > 4; csrc/whitted_tpt.inc): once at most `max_lanes` lanes of a tile's wavefront are
 * alive and they hold at least `min_paths` pending paths (current segments + continuations on their stacks), the rest of the tile is
 * traced by the whole wavefront as one pool of segments kept in a slot of a device pool of `pool_mb` MiB (8 192 slots).
 * Same image whatever the settings (the per-lane loop and the tail do the same arithmetic and add in the same order); max_lanes 64 sends
 * everything through the tail, 0 switches it off; a negative argument keeps the current value.  Defaults 40 / 4 / 8192, or
 * CLWRAP_TPT_MAX / CLWRAP_TPT_MIN / CLWRAP_TPT_POOL_MB.  Counting build: counter words 29 / 30 / 31 = tiles the tail gave up on (slice
 * full: finished by the per-lane loop) / tiles it finished / nodes it traced. */
void clw_ext_set_tpt(cl_wrap* wrap, int max_lanes, int min_paths, int pool_mb);

/* Host helper: camera -> the eight by-value raygen arguments, with the reference's exact
 * mixed float/double arithmetic (rinit_camera + rgen_perspective, src/cpu_ray.c:8-35, 42-106).
 * `look` need not be normalised.  Returns 0 for the cameras the reference rejects
 * (fov ~ 180, fov <= eps, look == +Y; cpu_ray.c:58-63), else 1. */
typedef struct clw_camera {
    float im_corner[3], origin[3], up[3], right[3];
    float w_factor, h_factor;
    uint32_t width, height;
} clw_camera;
int clw_host_perspective(const float origin[3], const float look[3], float fov, float focal,
                         uint32_t width, uint32_t height, clw_camera* out);

/* Host helpers: PNG files without libpng (reference png_dump, src/cpu_ray.c:108-165, and the
 * decode step of cl_wrap_load_images).  Return 0 on success. */
int clw_host_write_png(const char* path, const uint32_t* xrgb, uint32_t width, uint32_t height);
int clw_host_write_png_rgba(const char* path, const uint8_t* rgba, uint32_t width, uint32_t height);
int clw_host_read_png(const char* path, uint32_t* width, uint32_t* height, uint8_t** rgba_malloced);
void clw_host_free(void* p);

/* Library build info, e.g. "opencl_wrap_hip gfx950 fast+strict". */
const char* clw_ext_version(void);

#ifdef __cplusplus
}
#endif
#endif /* HIP_WRAP_EXT_H */
