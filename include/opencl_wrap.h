/*
 * opencl_wrap.h -- MI355X drop-in for the reference's device wrapper.
 *
 * Replaces reference src/opencl_wrap.h + src/opencl_wrap.c.  The reference's
 * unchanged drivers (raypng.c, rayinteractive.c) compile against this header and
 * link libopencl_wrap_hip.so instead of opencl_wrap.c + libOpenCL; see
 * INTEGRATION.md.  <CL/opencl.h> is included for its TYPES ONLY (cl_uint,
 * cl_float3, cl_mem, CL_MEM_* ...): nothing from the OpenCL runtime is called and
 * libOpenCL is not linked.
 *
 * The six entry points keep the reference's names, argument order/meaning and
 * error convention (message on stdout as "ERROR:\t..." then exit(1)):
 *
 *   cl_wrap_init             replaces opencl_wrap.h:29     / opencl_wrap.c:11-127
 *   cl_wrap_load_global_data replaces opencl_wrap.h:32-33  / opencl_wrap.c:129-170
 *   cl_wrap_load_single_data replaces opencl_wrap.h:34-35  / opencl_wrap.c:172-187
 *   cl_wrap_load_images      replaces opencl_wrap.h:36-37  / opencl_wrap.c:189-349
 *   cl_wrap_output           replaces opencl_wrap.h:39-41  / opencl_wrap.c:351-398
 *   cl_wrap_release          replaces opencl_wrap.h:42     / opencl_wrap.c:400-416
 *
 * Differences a caller can observe (all documented in DESIGN.md):
 *   - kernels are precompiled HIP code selected by NAME ("raygen", "raytracer");
 *     the source-path arguments are accepted and not read;
 *   - `buffers[k][a]` holds an opaque handle, not a cl_mem of a CL runtime; passing
 *     its 8 bytes back through cl_wrap_load_single_data still means "that buffer"
 *     (raypng.c:61);
 *   - knobs the API has no slot for (trace depth, row strips, device) come from
 *     hip_wrap_ext.h or the CLWRAP_* environment variables; defaults reproduce the
 *     reference (depth 15, 2 soft-shadow samples, device 0).
 */
#ifndef HIP_OPENCL_WRAP_H
#define HIP_OPENCL_WRAP_H

#ifndef CL_TARGET_OPENCL_VERSION
#define CL_TARGET_OPENCL_VERSION 300
#endif
#include <CL/opencl.h>
#include <stddef.h>

#define __MAX_KERNELS 16 /* opencl_wrap.h:6 */
#define __MAX_BUFFERS 32 /* opencl_wrap.h:7 */

#ifdef __cplusplus
extern "C" {
#endif

/* Caller-allocated (stack in raypng.c:31, global in rayinteractive.c:28).  Only
 * `buffers` is read by the drivers (raypng.c:61, rayinteractive.c:158). */
typedef struct cl_wrap {
    void*   impl;                                        /* shim state (device, stream, kernels) */
    cl_uint kernels_num;
    cl_uint buffers_num[__MAX_KERNELS];                  /* registered buffer args per kernel   */
    cl_uint buffers_ids[__MAX_KERNELS][__MAX_BUFFERS];   /* their arg ids, in registration order */
    cl_mem  buffers[__MAX_KERNELS][__MAX_BUFFERS];       /* opaque handle per (kernel, arg)      */
} cl_wrap;

/* varargs: (const char* source_path, const char* kernel_name)*, NULL */
void cl_wrap_init(cl_wrap* wrap, cl_device_type type, ...);

/* Creates a device buffer of `size` bytes as argument `arg_id` of kernel `kernel_id`;
 * `data` != NULL is copied in (blocking). */
void cl_wrap_load_global_data(cl_wrap* wrap, cl_uint kernel_id, cl_uint arg_id, const void* data,
                              size_t size, cl_mem_flags mem_flags);

/* Sets a by-value argument (copied at call time; may be called again at any time). */
void cl_wrap_load_single_data(cl_wrap* wrap, cl_uint kernel_id, cl_uint arg_id, const void* data,
                              size_t obj_size);

/* varargs: image_num x const char* png_path.  8-bit RGB PNGs of equal size become one
 * RGBA8 (A = 255) layer stack bound as argument `arg_id`. */
void cl_wrap_load_images(cl_wrap* wrap, cl_uint kernel_id, cl_uint arg_id, cl_mem_flags mem_flags,
                         cl_uint image_num, ...);

/* Runs kernel `kernel_run_id` over `array_size` work-items, waits for it, and, if
 * `host_output` != NULL, copies `output_size` bytes of buffers[kernel_id][arg_id] to it. */
void cl_wrap_output(cl_wrap* wrap, size_t array_size, size_t output_size, cl_uint kernel_run_id,
                    cl_uint kernel_id, cl_int arg_id, void* host_output);

void cl_wrap_release(cl_wrap* wrap);

#ifdef __cplusplus
}
#endif
#endif /* HIP_OPENCL_WRAP_H */
