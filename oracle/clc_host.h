/*
 * clc_host.h -- TEST INFRASTRUCTURE ONLY (oracle/). Not part of the product path.
 *
 * A small OpenCL-C-on-host language layer: enough of the OpenCL C vector types
 * and built-ins for clang++ to compile the reference's kernel sources
 * (/root/reference/src/cl/{types,primitives,raygen,raytracing}.cl) for x86-64
 * *where they lie* -- nothing from the reference is copied into this repo.
 * It is used only by oracle/ref_harness.cpp to build oracle/_ref/libref_cl.so,
 * which pins the CPU restatement (oracle/whitted_oracle.c) bit-for-bit and
 * generates the fixtures under tests/golden/ (see SURVEY.md section 8(c)).
 *
 * Semantics chosen where OpenCL leaves latitude (and mirrored exactly by the
 * restatement so both are comparable bit-for-bit under -ffp-contract=off):
 *   dot(a,b)      = a.x*b.x + a.y*b.y + a.z*b.z   (left to right)
 *   normalize(v)  = v / sqrtf(dot(v,v))
 *   distance(a,b) = sqrtf(dot(a-b, a-b))
 *   sin/cos/pow/sqrt/fabs = glibc sinf/cosf/powf/sqrtf/fabsf
 *   read_imagei   = unfiltered, unnormalised fetch from a raw RGBA8 array;
 *                   out-of-range coordinates are counted and clamped (the
 *                   OpenCL result would be undefined).
 *   (int)float    = C++ cast (x86 semantics); out-of-int-range casts never
 *                   occur on the golden scenes (checked by the restatement's
 *                   counters).
 */
#pragma once
#include <cmath>
#include <cstdint>
#include <cstddef>

typedef float float3 __attribute__((ext_vector_type(3)));
typedef float float4 __attribute__((ext_vector_type(4)));
typedef int int2 __attribute__((ext_vector_type(2)));
typedef int int4 __attribute__((ext_vector_type(4)));
typedef unsigned int uint;
typedef unsigned char uchar;

#define __kernel
#define __global
#define read_only
#ifndef M_1_PI_F
#define M_1_PI_F 0.31830988618379067154f
#endif

/* ---- image objects: a raw RGBA8 layer stack ---------------------------- */
struct clc_image {
    const uint8_t* texels; /* layer-major, row-major, 4 B per texel */
    int width, height, layers;
};
typedef const clc_image* image2d_array_t;

extern thread_local size_t clc_global_id;
extern thread_local unsigned long clc_oob_reads;

static inline size_t get_global_id(int) { return clc_global_id; }

static inline int2 get_image_dim(image2d_array_t im) { return (int2){im->width, im->height}; }

static inline int4 read_imagei(image2d_array_t im, int4 c) {
    int x = c.x, y = c.y, l = c.z;
    if (x < 0 || y < 0 || l < 0 || x >= im->width || y >= im->height || l >= im->layers) {
        clc_oob_reads++;
        x = x < 0 ? 0 : (x >= im->width ? im->width - 1 : x);
        y = y < 0 ? 0 : (y >= im->height ? im->height - 1 : y);
        l = l < 0 ? 0 : (l >= im->layers ? im->layers - 1 : l);
    }
    const uint8_t* p = im->texels + 4 * ((size_t)l * im->width * im->height + (size_t)y * im->width + x);
    return (int4){p[0], p[1], p[2], p[3]};
}

/* ---- geometric / math built-ins ---------------------------------------- */
static inline float dot(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline float3 cross(float3 a, float3 b) {
    return (float3){a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
static inline float3 normalize(float3 v) { return v / sqrtf(dot(v, v)); }
static inline float distance(float3 a, float3 b) { float3 d = a - b; return sqrtf(dot(d, d)); }

static inline float max(float a, float b) { return a > b ? a : b; }
static inline float clamp1(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
static inline float3 clamp(float3 v, float lo, float hi) {
    return (float3){clamp1(v.x, lo, hi), clamp1(v.y, lo, hi), clamp1(v.z, lo, hi)};
}
static inline float sin(float x) { return sinf(x); }
static inline float cos(float x) { return cosf(x); }
static inline float sqrt(float x) { return sqrtf(x); }
static inline float fabs(float x) { return fabsf(x); }
static inline float pow(float a, float b) { return powf(a, b); }
static inline bool isnan(float x) { return x != x; }
