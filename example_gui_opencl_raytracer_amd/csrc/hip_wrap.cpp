/*
 * hip_wrap.cpp -- the C-ABI shim: the reference's six cl_wrap_* entry points
 * (reference src/opencl_wrap.h:29-42, src/opencl_wrap.c:11-416) implemented over the HIP
 * runtime and the precompiled gfx950 kernels of whitted_{fast,strict}.hip, plus the
 * extension entry points of include/hip_wrap_ext.h.
 *
 * Same conventions as the reference: every failure prints "ERROR:\t<message>" on stdout
 * and calls exit(1); every call returns with its work complete (unless the caller opted
 * into async mode); single host thread.
 *
 * There is NO CPU fallback: without a HIP device cl_wrap_init fails exactly as the
 * reference fails without an OpenCL GPU (opencl_wrap.c:31-34).
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/hip_wrap_ext.h"
#include "png_codec.h"
#include "scene_prep.h"
#include "whitted_params.h"

extern "C" hipError_t wt_fast_launch_trace(const whitted_params*, int, unsigned, size_t, hipStream_t);
extern "C" hipError_t wt_fast_launch_raygen(const raygen_params*, hipStream_t);
extern "C" hipError_t wt_strict_launch_trace(const whitted_params*, int, unsigned, size_t, hipStream_t);
extern "C" hipError_t wt_strict_launch_raygen(const raygen_params*, hipStream_t);
extern "C" hipError_t wt_fast_launch_unit(int, const float*, float*, unsigned, unsigned, unsigned, unsigned, hipStream_t);
extern "C" hipError_t wt_strict_launch_unit(int, const float*, float*, unsigned, unsigned, unsigned, unsigned, hipStream_t);
extern "C" hipError_t wt_fast_launch_unit_scene(const whitted_params*, int, int, const float*, float*, unsigned, unsigned, unsigned, size_t, hipStream_t);
extern "C" hipError_t wt_strict_launch_unit_scene(const whitted_params*, int, int, const float*, float*, unsigned, unsigned, unsigned, size_t, hipStream_t);
extern "C" hipError_t wt_fast_launch_sched(const unsigned*, unsigned*, unsigned, unsigned, unsigned, unsigned, unsigned, unsigned, unsigned, hipStream_t);

namespace {

enum { F_COUNT = 1, F_DEEP = 2, F_GEOM_LDS = 4, F_RAYS = 8, F_GRID = 16, F_OCC = 32, F_D8 = 64, F_D16 = 128 }; /* = WT_F_* of whitted_trace.inc */
/* deep launches of at least OCC_TILES_PER_DEPTH x depth wavefronts take the high-occupancy flavour: the serial tail of
 * the deepest refraction trees grows with the depth, the throughput part with the tile count (tools/occ_sweep.py on
 * render.map: wins 12-14 % at 2560x1440 depth 6 and 3840x2160 depth 6-8, loses 2-5 % at 1920x1080 and at depth 15) */
constexpr unsigned OCC_TILES_PER_DEPTH = 9000;
/* tree-parallel tail (whitted_tpt.inc): a deep launch enters it with at most TPT_MAX_LANES live lanes holding at least TPT_MIN_PATHS
 * pending paths; the pool holds TPT_SLOTS_PER_XCC slots per XCD (more than an XCD's CUs hold wavefronts) of at most TPT_SLICE_WORDS_MAX
 * words, TPT_POOL_MB in all (below TPT_MIN_CAP nodes per slot -- hundreds of lights -- the launch runs without the tail) */
constexpr unsigned TPT_MAX_LANES = 40, TPT_MIN_PATHS = 4, TPT_POOL_MB = 8192, TPT_MIN_CAP = 160, TPT_SLOTS_PER_XCC = 1024;
constexpr uint64_t TPT_SLICE_WORDS_MAX = 1u << 18;
/* heavy tiles of such a launch are served by up to 16 wavefronts each (wt_sched_build): up to SPLIT_EXTRA_PER_SHARE more dispatch entries
 * per XCD share; a tile is split while its parts stay above SPLIT_MIN_QUOTA cost units (a part enters the tail at once and pays its
 * fixed costs: 800x600 depth 15 0.267 ms at 3 000, 0.208 at 750-1 500); the quota is the share's total cost over SPLIT_SLOTS wavefronts (384 / 512 / 640: 1920x1080 depth 15 0.370 / 0.332 / 0.322 ms, glass field 2048^2 depth 8 3.19 / 2.95 / 3.33) */
constexpr unsigned SPLIT_EXTRA_PER_SHARE = 1024, SPLIT_MIN_QUOTA = 1500, SPLIT_SLOTS = 512;
constexpr size_t COUNTER_WORDS = CLW_NUM_COUNTERS + 16 * (size_t)CLW_STAMP_SHARDS;
constexpr unsigned BLOCK = 256;       /* the reference's launch rounding unit: CL_KERNEL_WORK_GROUP_SIZE on AMD (opencl_wrap.c:359-374) */
constexpr unsigned TRACE_BLOCK = 64;  /* = WT_BLOCK: one wavefront per workgroup */
constexpr size_t GEOM_LDS_MAX_F4 = 1024; /* <= 16 KiB of prepared geometry is staged in LDS */
constexpr uint32_t VIS_MAX_SPHERES = 16; /* = WT_VIS_MAX_SPHERES of whitted_trace.inc */
constexpr int SHALLOW_LEVELS = 3;   /* DFS levels the shallow builds provide (LDS + scratch): depth <= SHALLOW_LEVELS + 1 */
constexpr uint32_t GRID_MIN_SPHERES = 256;   /* scenes beyond the reference's one-byte counts get the uniform grid (at 64 spheres it only
                                                wins when the cells happen to align with the spheres: 9.2-15 ms vs 10.9 ms linear) */
constexpr size_t GRID_MAX_PAIRS = (size_t)1 << 25;   /* 20 B per cell-list entry */

[[noreturn]] void die(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    printf("ERROR:\t");
    vprintf(fmt, ap);
    printf("\n");
    va_end(ap);
    fflush(stdout);
    exit(1);
}

#define HIP_OK(call, what)                                                        \
    do {                                                                          \
        hipError_t e_ = (call);                                                   \
        if (e_ != hipSuccess) die("%s (%s)", what, hipGetErrorString(e_));        \
    } while (0)

enum KernelKind { K_RAYGEN = 0, K_RAYTRACER = 1 };

struct RaygenArgs { /* snapshot of the eight by-value raygen arguments + the launch range */
    float corner[3], origin[3], up[3], right[3];
    float w_factor, h_factor;
    uint32_t width, height;
    uint64_t id_offset;
    uint32_t n_items;
    uint32_t band_stride, band_phase;
};
/* field by field (the struct has tail padding, which a memcmp would also compare); floats by bit pattern */
bool same_raygen(const RaygenArgs& a, const RaygenArgs& b) {
    return !memcmp(a.corner, b.corner, 12) && !memcmp(a.origin, b.origin, 12) && !memcmp(a.up, b.up, 12) &&
           !memcmp(a.right, b.right, 12) && !memcmp(&a.w_factor, &b.w_factor, 4) && !memcmp(&a.h_factor, &b.h_factor, 4) &&
           a.width == b.width && a.height == b.height && a.id_offset == b.id_offset && a.n_items == b.n_items &&
           a.band_stride == b.band_stride && a.band_phase == b.band_phase;
}

struct Buffer {
    void* dptr = nullptr;
    size_t size = 0;
    bool owned = true;       /* freed by release */
    bool lazy = false;       /* created with data == NULL: allocated on first real use */
    bool image = false;
    uint32_t w = 0, h = 0, layers = 0;
    std::vector<uint8_t> shadow;   /* host copy of uploaded data (scene arrays are re-read by scene prep) */
    bool gen_valid = false;        /* holds rays "generated" by a fused raygen launch */
    bool exposed = false;          /* its device pointer was handed out (clw_ext_device_ptr) since the rays were generated: a caller may have
                                      rewritten them in place, so they are no longer known to be this library's own unit vectors */
    bool gen_materialised = false;
    RaygenArgs gen{};
};

struct ArgValue {
    bool set = false;
    size_t size = 0;
    uint8_t bytes[32] = {0};
};

struct Kernel {
    KernelKind kind;
    std::string name;
    ArgValue values[__MAX_BUFFERS];
};

struct TimingEntry { hipEvent_t start, stop; cl_uint kernel; };

struct Impl {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::vector<Kernel> kernels;
    std::vector<Buffer*> live;      /* every buffer handed out as a cl_mem handle */
    /* knobs */
    int depth = 15;
    int strict = 0;
    int fuse = 1;
    int async = 0;
    int variant = 0;
    int counting = 0;
    float through = 0.8f;  /* primitives.cl:7 TRANSPERENT_THROUGH; CLWRAP_THROUGH / clw_ext_set_shadow_through */
    int stamps = 0;        /* CLWRAP_STAMPS=1: hand the counter block to the (diagnostic) stamp build of the kernel */
    uint64_t id_offset = 0;
    uint32_t band_stride = 1, band_phase = 0;
    float* debug_rgb = nullptr;
    /* prepared scene cache */
    const Buffer *prep_s = nullptr, *prep_p = nullptr, *prep_l = nullptr;
    uint32_t prep_ns = 0, prep_np = 0, prep_nl = 0;
    float* d_geom = nullptr; size_t geom_f4 = 0;
    float* d_ptex = nullptr;
    bool have_lpt = false;    /* the light / plane side table follows the lights in d_geom */
    uint64_t scene_generation = 0;   /* bumped every time the prepared scene is rebuilt (device pointers can be reused) */
    uint32_t *d_grid_start = nullptr, *d_grid_items = nullptr, *d_grid_box = nullptr;
    float* d_grid_geom = nullptr;
    wprep_grid grid{}; bool grid_ok = false, grid_pair = false; int use_grid = 1;
    unsigned long long* d_counters = nullptr;
    /* cost-sorted tile dispatch: costs written by frame n order the tiles of frame n+1 */
    int sched = 1;
    /* Double-buffered: trace k writes its tile costs into cost[k & 1]; when the camera / depth / scene changed, the
     * order for those costs is built on a SIDE stream behind trace k (order[k & 1], 15 us, 8 workgroups) while trace
     * k+1 already runs with the order built two frames earlier -- so a moving camera (the interactive loop of
     * rayinteractive.c:183-197) never waits for the sort.  Any order built for the same tile grid is a valid
     * permutation, so a stale one only costs balance, never pixels. */
    struct Sched {
        unsigned* cost[2] = {nullptr, nullptr}; unsigned* order[2] = {nullptr, nullptr};
        hipEvent_t built[2] = {nullptr, nullptr};    /* order[i] complete (recorded on the sched stream) */
        bool have[2] = {false, false};                /* order[i] holds / will hold a schedule */
        int wr = 0;                                   /* cost buffer the next trace writes */
        hipEvent_t traced = nullptr;
        uint32_t w = 0, rows = 0, cap = 0;           /* frame the buffers were sized for; dispatch entries per XCD share */
        RaygenArgs sig{}; int sig_depth = 0; uint64_t sig_scene = 0; bool sig_valid = false; int sig_age = 0, newest = 0; uint64_t frame = 0, newest_frame = 0;   /* what the newest order was built for, frames since */
        void reset() { have[0] = have[1] = false; sig_valid = false; newest = 0; }
        void free_all() {
            for (int i = 0; i < 2; i++) {
                if (cost[i]) (void)hipFree(cost[i]);
                if (order[i]) (void)hipFree(order[i]);
                if (built[i]) (void)hipEventDestroy(built[i]);
                cost[i] = order[i] = nullptr; built[i] = nullptr;
            }
            if (traced) { (void)hipEventDestroy(traced); traced = nullptr; }
            reset();
        }
    };
    hipStream_t sched_stream = nullptr;
    static constexpr int MAX_CHUNKS = 4;
    Sched scheds[1 + MAX_CHUNKS];   /* [0]: whole-range launches; [1 + c]: strip c of a pipelined read-back */
    /* pipelined read-back (cl_wrap_output of a large frame): strips rendered back to back, each copied to the host
     * while the next one renders */
    int pipeline = 1;
    hipStream_t copy_stream = nullptr;
    hipEvent_t chunk_done[MAX_CHUNKS] = {nullptr, nullptr, nullptr, nullptr};
    /* timing log */
    std::vector<TimingEntry> timing;
    unsigned occ_tiles_per_depth = OCC_TILES_PER_DEPTH;   /* CLWRAP_OCC_TILES_PER_DEPTH: tuning knob */
    /* the tree-parallel tail of deep launches (whitted_tpt.inc): one slice of node storage per workgroup of the launch */
    uint32_t* d_tpt_pool = nullptr; uint32_t* d_tpt_flags = nullptr; size_t tpt_pool_bytes = 0;
    uint32_t* d_tpt_jump = nullptr;              /* 7 x 32 words: M^(2^i), M = one hit's worth of xorshift steps (xorshift_jump_matrices) */
    unsigned tpt_max = TPT_MAX_LANES, tpt_min = TPT_MIN_PATHS, tpt_pool_mb = TPT_POOL_MB;   /* CLWRAP_TPT_MAX / _MIN / _POOL_MB, clw_ext_set_tpt */
    int tpt_clock = 0;                                    /* CLWRAP_TPT_CLOCK=1: DIAGNOSTIC phase clock of the tail (counter words 10-25) */
    unsigned split_slots = SPLIT_SLOTS;                   /* CLWRAP_SPLIT_SLOTS: the quota is a share's total cost over this many wavefronts */
    unsigned split_min_quota = SPLIT_MIN_QUOTA;           /* CLWRAP_SPLIT_MIN_QUOTA; 0 = heavy tiles are not split */
    bool timing_on = false;                       /* switched on by the first clw_ext_timing_reset / set_timing_every */
    uint32_t timing_every = 1, timing_tick = 0;   /* events around every n-th launch only */
    std::vector<std::pair<hipEvent_t, hipEvent_t>> free_events;
};

Impl* impl_of(const cl_wrap* w) {
    if (!w || !w->impl) die("cl_wrap is not initialised");
    return (Impl*)w->impl;
}

void use_device(Impl* I) {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != I->device) HIP_OK(hipSetDevice(I->device), "Cannot select the HIP device");
}

bool is_registered(const cl_wrap* w, cl_uint kernel_id, cl_uint arg_id) {
    for (cl_uint i = 0; i < w->buffers_num[kernel_id]; i++)
        if (w->buffers_ids[kernel_id][i] == arg_id) return true;
    return false;
}

void check_kernel_id(const cl_wrap* w, cl_uint kernel_id) {
    if (kernel_id >= w->kernels_num) die("Wrong kernel ID given");
}

Buffer* new_buffer(Impl* I) {
    Buffer* b = new Buffer();
    I->live.push_back(b);
    return b;
}

Buffer* lookup_handle(Impl* I, const void* handle) {
    for (Buffer* b : I->live)
        if ((const void*)b == handle) return b;
    return nullptr;
}

void ensure_allocated(Impl* I, Buffer* b) {
    (void)I;
    if (b->dptr || b->size == 0) return;
    HIP_OK(hipMalloc(&b->dptr, b->size), "Couldn't allocate device memory");
    b->lazy = false;
}

void register_buffer(cl_wrap* w, cl_uint kernel_id, cl_uint arg_id, Buffer* b) {
    w->buffers[kernel_id][arg_id] = (cl_mem)b;
    w->buffers_ids[kernel_id][w->buffers_num[kernel_id]++] = arg_id;
}

void precheck_buffer_arg(cl_wrap* w, cl_uint kernel_id, cl_uint arg_id) {
    check_kernel_id(w, kernel_id);
    if (arg_id < __MAX_BUFFERS && is_registered(w, kernel_id, arg_id)) die("Given kernel argument already in use");
    if (arg_id >= __MAX_BUFFERS) die("Wrong kernel ID given"); /* the reference's wording (opencl_wrap.c:144) */
}

/* ---- launch helpers -------------------------------------------------------------------- */
std::pair<hipEvent_t, hipEvent_t> take_events(Impl* I) {
    if (!I->free_events.empty()) {
        auto p = I->free_events.back();
        I->free_events.pop_back();
        return p;
    }
    hipEvent_t a, b;
    HIP_OK(hipEventCreate(&a), "Couldn't create a timing event");
    HIP_OK(hipEventCreate(&b), "Couldn't create a timing event");
    return {a, b};
}

struct LaunchTimer {
    Impl* I; cl_uint kernel; bool on; std::pair<hipEvent_t, hipEvent_t> ev;
    LaunchTimer(Impl* I_, cl_uint k) : I(I_), kernel(k), on(I_->timing_on && I_->timing.size() < 16384 && (I_->timing_tick++ % I_->timing_every) == 0) {
        if (on) { ev = take_events(I); HIP_OK(hipEventRecord(ev.first, I->stream), "Couldn't run the kernel"); }
    }
    void done() {
        if (on) { HIP_OK(hipEventRecord(ev.second, I->stream), "Couldn't run the kernel"); I->timing.push_back({ev.first, ev.second, kernel}); }
    }
};

void finish(Impl* I) {
    hipError_t e = hipStreamSynchronize(I->stream);
    if (e != hipSuccess) { printf("%d\n", (int)e); die("The device kernel failed"); }
}

const ArgValue& need_value(const Kernel& k, cl_uint arg, size_t min_size, size_t max_size) {
    const ArgValue& v = k.values[arg];
    if (!v.set || v.size < min_size || v.size > max_size) die("Couldn't run the kernel");
    return v;
}

RaygenArgs snapshot_raygen(Impl* I, const Kernel& k, size_t array_size) {
    RaygenArgs g{};
    memcpy(g.corner, need_value(k, 0, 12, 16).bytes, 12);
    memcpy(g.origin, need_value(k, 1, 12, 16).bytes, 12);
    memcpy(g.up, need_value(k, 2, 12, 16).bytes, 12);
    memcpy(g.right, need_value(k, 3, 12, 16).bytes, 12);
    memcpy(&g.w_factor, need_value(k, 4, 4, 4).bytes, 4);
    memcpy(&g.h_factor, need_value(k, 5, 4, 4).bytes, 4);
    memcpy(&g.width, need_value(k, 6, 4, 4).bytes, 4);
    memcpy(&g.height, need_value(k, 7, 4, 4).bytes, 4);
    if (g.width == 0 || g.height == 0) die("Couldn't run the kernel");
    g.id_offset = I->id_offset;
    g.band_stride = I->band_stride; g.band_phase = I->band_phase;
    /* the reference launches ceil(array_size/local)*local items guarded by id < w*h (raygen.cl:11) */
    uint64_t rounded = ((uint64_t)array_size + BLOCK - 1) / BLOCK * BLOCK;
    uint64_t total = (uint64_t)g.width * g.height;
    uint64_t room = total > g.id_offset ? total - g.id_offset : 0;
    if (g.band_stride > 1) {   /* this launch owns every band_stride-th 8-row band of the frame */
        if (g.id_offset != 0 || g.height % (8 * g.band_stride) != 0) die("Row bands need height %% (8*stride) == 0 and no id offset");
        room = total / g.band_stride;
    }
    uint64_t n = rounded < room ? rounded : room;
    if (n > 0xFFFFFFFFull) die("Couldn't run the kernel");
    g.n_items = (uint32_t)n;
    return g;
}

void run_raygen_kernel(Impl* I, const RaygenArgs& g, Buffer* rays, cl_uint kernel_id) {
    ensure_allocated(I, rays);
    size_t cap = rays->size / 64;
    uint32_t n = g.n_items < cap ? g.n_items : (uint32_t)cap; /* never write past the buffer */
    if (n == 0) return;
    raygen_params P{};
    memcpy(P.corner, g.corner, 12); memcpy(P.origin, g.origin, 12);
    memcpy(P.up, g.up, 12); memcpy(P.right, g.right, 12);
    P.w_factor = g.w_factor; P.h_factor = g.h_factor; P.width = g.width; P.height = g.height;
    P.id_offset = g.id_offset; P.n_items = n; P.rays = (float*)rays->dptr;
    P.band_stride = g.band_stride; P.band_phase = g.band_phase;
    LaunchTimer t(I, kernel_id);
    hipError_t e = I->strict ? wt_strict_launch_raygen(&P, I->stream) : wt_fast_launch_raygen(&P, I->stream);
    if (e != hipSuccess) die("Couldn't run the kernel");
    t.done();
}

void materialise_rays(Impl* I, Buffer* b) {
    if (b->gen_valid && !b->gen_materialised) {
        run_raygen_kernel(I, b->gen, b, 0);
        b->gen_materialised = true;
    }
}

uint32_t read_count(const Kernel& k, cl_uint arg) {
    const ArgValue& v = k.values[arg];
    if (!v.set) die("Couldn't run the kernel");
    switch (v.size) { /* uchar in the reference (raytracing.cl:17); 2/4 bytes = wide-count extension */
        case 1: return v.bytes[0];
        case 2: { uint16_t x; memcpy(&x, v.bytes, 2); return x; }
        case 4: { uint32_t x; memcpy(&x, v.bytes, 4); return x; }
        default: die("Couldn't run the kernel");
    }
}

const uint8_t* host_view(Impl* I, Buffer* b, size_t need, std::vector<uint8_t>& tmp) {
    if (b->size < need) die("Couldn't run the kernel");
    if (b->shadow.size() >= need) return b->shadow.data();
    ensure_allocated(I, b);
    tmp.resize(need);
    if (need) HIP_OK(hipMemcpy(tmp.data(), b->dptr, need, hipMemcpyDeviceToHost), "Failed to transfer device memory to host");
    return tmp.data();
}

int env_int(const char* name, int dflt) {
    const char* s = getenv(name);
    return (s && *s) ? atoi(s) : dflt;
}

/* xorshift32 (primitives.cl:116-125) is linear over GF(2): a step is a 32x32 bit matrix T (column b = the step applied to 1 << b), the 4 nl numbers
 * a shaded hit draws are M = T^(4 nl), and the state k hits on is M^k times the state now.  out[i] = M^(2^i), i = 0..6, as columns: the tail's xorshift
 * pass (whitted_tpt.inc, pass C) jumps with them instead of stepping. */
void xorshift_jump_matrices(uint32_t steps, uint32_t out[7][32]) {
    auto apply = [](const uint32_t* A, uint32_t x) { uint32_t r = 0; for (int b = 0; b < 32; b++) if ((x >> b) & 1u) r ^= A[b]; return r; };
    auto mul = [&](const uint32_t* A, const uint32_t* B, uint32_t* C) { uint32_t t[32]; for (int b = 0; b < 32; b++) t[b] = apply(A, B[b]); memcpy(C, t, sizeof t); };
    uint32_t T[32], R[32], Bp[32];
    for (int b = 0; b < 32; b++) { uint32_t x = 1u << b; x ^= x << 13; x ^= x >> 17; x ^= x << 5; T[b] = x; R[b] = 1u << b; }
    memcpy(Bp, T, sizeof T);
    for (uint32_t e = steps; e; e >>= 1) { if (e & 1u) mul(Bp, R, R); mul(Bp, Bp, Bp); }      /* R = T^steps */
    memcpy(out[0], R, sizeof R);
    for (int i = 1; i < 7; i++) mul(out[i - 1], out[i - 1], out[i]);
}

void prepare_scene(Impl* I, Buffer* s, uint32_t ns, Buffer* p, uint32_t np, Buffer* l, uint32_t nl) {
    if (I->d_geom && I->prep_s == s && I->prep_p == p && I->prep_l == l && I->prep_ns == ns &&
        I->prep_np == np && I->prep_nl == nl)
        return;
    std::vector<uint8_t> ts, tp, tl;
    const uint8_t* hs = host_view(I, s, 96 * (size_t)ns, ts);
    const uint8_t* hp = host_view(I, p, 96 * (size_t)np, tp);
    const uint8_t* hl = host_view(I, l, 48 * (size_t)nl, tl);
    const size_t base_f4 = wprep_geom_f4(ns, np, nl), lpt_f4 = wprep_lpt_f4(np, nl);
    size_t f4 = base_f4 + lpt_f4;
    std::vector<float> geom(4 * (f4 ? f4 : 1)), ptex(8 * (size_t)(np ? np : 1));
    wprep_build(hs, ns, hp, np, hl, nl, geom.data(), ptex.data());
    wprep_build_lpt(hp, np, hl, nl, geom.data() + 4 * base_f4);
    I->have_lpt = lpt_f4 != 0;
    {   /* the tail's xorshift jumps for this light count */
        uint32_t jm[7][32];
        xorshift_jump_matrices(4u * nl, jm);
        if (!I->d_tpt_jump) HIP_OK(hipMalloc((void**)&I->d_tpt_jump, sizeof jm), "Couldn't allocate device memory");
        HIP_OK(hipMemcpy(I->d_tpt_jump, jm, sizeof jm, hipMemcpyHostToDevice), "Couldn't transfer the data from host to the device");
    }
    if (I->d_geom) { (void)hipFree(I->d_geom); I->d_geom = nullptr; }
    if (I->d_ptex) { (void)hipFree(I->d_ptex); I->d_ptex = nullptr; }
    HIP_OK(hipMalloc((void**)&I->d_geom, geom.size() * 4), "Couldn't allocate device memory");
    HIP_OK(hipMalloc((void**)&I->d_ptex, ptex.size() * 4), "Couldn't allocate device memory");
    HIP_OK(hipMemcpy(I->d_geom, geom.data(), geom.size() * 4, hipMemcpyHostToDevice), "Couldn't transfer the data from host to the device");
    HIP_OK(hipMemcpy(I->d_ptex, ptex.data(), ptex.size() * 4, hipMemcpyHostToDevice), "Couldn't transfer the data from host to the device");
    /* uniform grid over the spheres for big scenes */
    for (uint32_t** q : {&I->d_grid_start, &I->d_grid_items, &I->d_grid_box}) { if (*q) { (void)hipFree(*q); *q = nullptr; } }
    if (I->d_grid_geom) { (void)hipFree(I->d_grid_geom); I->d_grid_geom = nullptr; }
    I->grid_ok = false;
    if (ns > (uint32_t)env_int("CLWRAP_GRID_MIN", (int)GRID_MIN_SPHERES)) {   /* CLWRAP_GRID_MIN: tuning knob */
        const char* dens = getenv("CLWRAP_GRID_DENSITY");   /* tuning knob: average spheres per cell */
        /* the two soft-shadow samples of a light share one walk (wt_grid_shadow_pair) when spheres are listed a light radius beyond their
         * boxes -- worth it while that is a fraction of a cell (CLWRAP_GRID_PAIR=0: every shadow ray walks on its own) */
        float maxr = 0.0f;
        for (uint32_t l = 0; l < nl; l++) { float r; memcpy(&r, hl + 48 * (size_t)l + 16, 4); r = fabsf(r); if (!(r <= maxr)) maxr = r; }
        const float density = dens ? (float)atof(dens) : 1.6f;
        size_t pairs = 0;
        I->grid_pair = false;
        if (env_int("CLWRAP_GRID_PAIR", 1) && nl > 0 && maxr > 0.0f && std::isfinite(maxr)) {
            pairs = wprep_grid_plan(hs, ns, density, maxr, &I->grid);
            I->grid_pair = maxr <= 0.3f * std::min(I->grid.cell[0], std::min(I->grid.cell[1], I->grid.cell[2])) && pairs <= GRID_MAX_PAIRS;
        }
        if (!I->grid_pair) pairs = wprep_grid_plan(hs, ns, density, 0.0f, &I->grid);
        if (pairs <= GRID_MAX_PAIRS) {
            std::vector<uint32_t> st((size_t)I->grid.ncells + 1), it(pairs ? pairs : 1), bx(2 * (size_t)ns);
            wprep_grid_fill(hs, ns, &I->grid, st.data(), it.data(), bx.data());
            HIP_OK(hipMalloc((void**)&I->d_grid_start, st.size() * 4), "Couldn't allocate device memory");
            HIP_OK(hipMalloc((void**)&I->d_grid_items, it.size() * 4), "Couldn't allocate device memory");
            HIP_OK(hipMalloc((void**)&I->d_grid_box, bx.size() * 4), "Couldn't allocate device memory");
            HIP_OK(hipMemcpy(I->d_grid_start, st.data(), st.size() * 4, hipMemcpyHostToDevice), "Couldn't transfer the data from host to the device");
            HIP_OK(hipMemcpy(I->d_grid_items, it.data(), it.size() * 4, hipMemcpyHostToDevice), "Couldn't transfer the data from host to the device");
            HIP_OK(hipMemcpy(I->d_grid_box, bx.data(), bx.size() * 4, hipMemcpyHostToDevice), "Couldn't transfer the data from host to the device");
            /* the prepared sphere of every cell-list entry, next to its index */
            std::vector<float> cg(4 * it.size(), 0.0f);
            for (size_t k = 0; k < pairs; k++) memcpy(&cg[4 * k], &geom[4 * (size_t)it[k]], 16);
            HIP_OK(hipMalloc((void**)&I->d_grid_geom, cg.size() * 4), "Couldn't allocate device memory");
            HIP_OK(hipMemcpy(I->d_grid_geom, cg.data(), cg.size() * 4, hipMemcpyHostToDevice), "Couldn't transfer the data from host to the device");
            I->grid_ok = true;
        }
    }
    I->geom_f4 = f4;
    I->scene_generation++;
    I->prep_s = s; I->prep_p = p; I->prep_l = l; I->prep_ns = ns; I->prep_np = np; I->prep_nl = nl;
}

Buffer* buffer_arg(cl_wrap* w, cl_uint kernel_id, cl_uint arg) {
    if (!is_registered(w, kernel_id, arg)) die("Couldn't run the kernel");
    return (Buffer*)w->buffers[kernel_id][arg];
}

/* The scene side of a raytracer launch (args 1-6, 8, 9 of kernel `kid`): checks the bindings, prepares the geometry
 * (once per scene) and fills the scene fields of P; chooses between the uniform grid, LDS-staged geometry and
 * geometry read from global memory (flags F_GRID / F_GEOM_LDS, dynamic LDS bytes). */
void bind_scene(cl_wrap* w, Impl* I, cl_uint kid, whitted_params& P, int& flags, size_t& dyn_lds) {
    Kernel& k = I->kernels[kid];
    Buffer* bs = buffer_arg(w, kid, 1);
    Buffer* bp = buffer_arg(w, kid, 2);
    Buffer* bl = buffer_arg(w, kid, 3);
    uint32_t ns = read_count(k, 4), np = read_count(k, 5), nl = read_count(k, 6);
    Buffer* tex = buffer_arg(w, kid, 8);
    Buffer* sky = buffer_arg(w, kid, 9);
    if (!tex->image || !sky->image) die("Couldn't run the kernel");
    prepare_scene(I, bs, ns, bp, np, bl, nl);
    ensure_allocated(I, bs); ensure_allocated(I, bp);
    P.geom = I->d_geom; P.ptex = I->d_ptex; P.geom_f4 = (uint32_t)I->geom_f4;
    P.lpt = (I->have_lpt && !(I->variant & 128)) ? 1u : 0u;
    /* visibility classes of the lights (wt_light_vis; compiled into the strict build's LDS-geometry kernels): small scenes whose planes are
     * covered by the side table; variant 256 = off, variant 1024 = verification (counting build: classify AND trace, disagreements in
     * counter word 28) */
    P.vis = (I->strict && ns <= VIS_MAX_SPHERES && (np == 0 || P.lpt) && !(I->variant & 256)) ? ((I->variant & 1024) ? 2u : 1u) : 0u;
    P.spheres_raw = (const uint8_t*)bs->dptr; P.planes_raw = (const uint8_t*)bp->dptr;
    P.ns = ns; P.np = np; P.nl = nl;
    P.through = I->through;
    P.tex = (const uint32_t*)tex->dptr; P.tex_w = (int)tex->w; P.tex_h = (int)tex->h; P.tex_layers = (int)tex->layers;
    P.sky = (const uint32_t*)sky->dptr; P.sky_w = (int)sky->w; P.sky_h = (int)sky->h;
    if (I->grid_ok && I->use_grid && !(I->variant & 8)) {
        flags |= F_GRID;
        P.grid_pair = I->grid_pair ? 1u : 0u;
        P.grid_start = I->d_grid_start; P.grid_items = I->d_grid_items; P.grid_box = I->d_grid_box; P.grid_geom = I->d_grid_geom;
        for (int a = 0; a < 3; a++) {
            P.grid_min[a] = I->grid.gmin[a]; P.grid_inv[a] = I->grid.inv[a]; P.grid_cell[a] = I->grid.cell[a]; P.grid_res[a] = I->grid.res[a];
        }
    } else if (I->geom_f4 <= GEOM_LDS_MAX_F4 && ns <= GRID_MIN_SPHERES && !(I->variant & 1)) {   /* (<= 256 spheres: the LDS kernels take a = d.d = 1, see unit_dirs) */
        /* the prepared geometry is staged in LDS */
        flags |= F_GEOM_LDS;
        dyn_lds = I->geom_f4 * 16;
    }
}

/* one strip of a frame: rows [row0, row0 + rows) of the launch range, scheduling state in scheds[slot] */
struct Strip { uint32_t row0, rows; int slot; };

void run_raytracer(cl_wrap* w, Impl* I, cl_uint kid, size_t array_size, const Strip* strip = nullptr) {
    Kernel& k = I->kernels[kid];
    /* arg 0: the ray buffer, passed by value as the 8 bytes of a handle (raypng.c:61) or bound as a buffer */
    Buffer* rays = nullptr;
    if (is_registered(w, kid, 0)) rays = (Buffer*)w->buffers[kid][0];
    else {
        const ArgValue& v = need_value(k, 0, sizeof(cl_mem), sizeof(cl_mem));
        void* h; memcpy(&h, v.bytes, sizeof h);
        rays = lookup_handle(I, h);
        if (!rays) die("Couldn't run the kernel");
    }
    uint32_t total; memcpy(&total, need_value(k, 7, 4, 4).bytes, 4);
    Buffer* out = buffer_arg(w, kid, 10);

    uint64_t rounded = ((uint64_t)array_size + BLOCK - 1) / BLOCK * BLOCK;
    uint64_t n64 = rounded < total ? rounded : total;      /* guard `id >= total_size` (raytracing.cl:24) */
    ensure_allocated(I, out);
    if (n64 > out->size / 4) n64 = out->size / 4;          /* never write past the framebuffer */
    if (n64 == 0) return;
    if (n64 > 0xFFFFFFFFull) die("Couldn't run the kernel");

    whitted_params P{};
    int flags = 0;
    size_t dyn_lds = 0;
    bind_scene(w, I, kid, P, flags, dyn_lds);
    P.n_items = (uint32_t)n64;
    P.depth = I->depth;
    P.out = (uint32_t*)out->dptr;
    P.out_rgb = I->debug_rgb;
    P.diag = (I->variant & 512) ? (env_int("CLWRAP_TIMELINE_EDGES", 0) ? 255u + (uint32_t)env_int("CLWRAP_TIMELINE_EDGES", 0) : 1u + (uint32_t)env_int("CLWRAP_TIMELINE_SHIFT", 0)) : 0u;

    const bool fused = I->fuse && rays->gen_valid && !rays->exposed;   /* (a buffer whose pointer was handed out is read, not regenerated) */
    RaygenArgs g{};
    if (fused) {
        g = rays->gen;
        if (strip) {   /* the caller checked: whole rows, no bands */
            g.id_offset += (uint64_t)strip->row0 * g.width;
            g.n_items = strip->rows * g.width;
            P.n_items = g.n_items;
            P.out = (uint32_t*)out->dptr + (size_t)strip->row0 * g.width;
        }
        memcpy(P.corner, g.corner, 12); memcpy(P.origin, g.origin, 12);
        memcpy(P.up, g.up, 12); memcpy(P.right, g.right, 12);
        P.w_factor = g.w_factor; P.h_factor = g.h_factor; P.width = g.width; P.height = g.height;
        P.id_offset = g.id_offset;
        P.band_stride = g.band_stride; P.band_phase = g.band_phase;
        if (P.n_items > g.n_items) P.n_items = g.n_items; /* rays past the generated range are undefined in the reference */
        if (P.n_items == 0) return;
        P.tiled = (g.id_offset % g.width == 0 || g.band_stride > 1) && (P.n_items % g.width == 0) && !(I->variant & 2);
        P.rows = P.n_items / g.width;
        P.row_offset = (uint32_t)(g.id_offset / g.width);
        P.unit_dirs = P.ns <= GRID_MIN_SPHERES ? 1u : 0u;      /* primary rays are generated (and normalised) in the kernel */
    } else {
        materialise_rays(I, rays);
        ensure_allocated(I, rays);
        if ((uint64_t)P.n_items * 64 > rays->size) die("Couldn't run the kernel");
        P.rays = (const float*)rays->dptr;
        P.id_offset = I->id_offset;
        P.width = 1; P.height = 1;
        P.unit_dirs = (rays->gen_valid && !rays->exposed && P.ns <= GRID_MIN_SPHERES) ? 1u : 0u;
        if (rays->gen_valid) {   /* ids (RNG seeds) follow the launch that generated the rays */
            P.id_offset = rays->gen.id_offset; P.width = rays->gen.width; P.height = rays->gen.height;
            P.band_stride = rays->gen.band_stride; P.band_phase = rays->gen.band_phase;
        } else if (I->band_stride > 1) die("Row bands need a raygen launch");
        flags |= F_RAYS;
    }
    if (I->depth > SHALLOW_LEVELS + 1) flags |= F_DEEP;
    if (!(flags & F_GEOM_LDS)) P.vis = 0;
    if (I->counting || I->stamps || I->tpt_clock) {
        if (I->counting) flags |= F_COUNT;
        P.tpt_clock = I->tpt_clock ? 1u : 0u;
        if (!I->d_counters) {
            HIP_OK(hipMalloc((void**)&I->d_counters, COUNTER_WORDS * sizeof(unsigned long long)), "Couldn't allocate device memory");
            HIP_OK(hipMemsetAsync(I->d_counters, 0, COUNTER_WORDS * sizeof(unsigned long long), I->stream), "Couldn't allocate device memory");
        }
        P.counters = I->d_counters;
    }
    unsigned grid;
    Impl::Sched& S = I->scheds[strip ? strip->slot : 0];
    bool sched_rebuild = false;
    unsigned trows = 0, tpr = 0, per_share = 0, per_share_cap = 0, base_grid = 0;
    /* deep launches with the tree-parallel tail: a tile's cost is its total work, and heavy tiles are split over several wavefronts */
    if (P.tiled) {
        trows = (P.rows + 7) / 8; tpr = (P.width + 7) / 8;
        per_share = ((trows + 7) / 8) * tpr;
        per_share_cap = per_share;
        grid = base_grid = 8 * per_share;
    } else {
        grid = base_grid = (P.n_items + TRACE_BLOCK - 1) / TRACE_BLOCK;
    }
    /* big deep launches are throughput-bound: the high-occupancy flavour (96 VGPRs, five waves per SIMD) -- which does not carry the tail:
     * under that register cap the tail's passes spill, and its code cost the per-lane loop 6 % (4096^2 depth 8: 7.4 -> 7.9 ms) */
    if ((flags & F_DEEP) && !I->strict && (uint64_t)base_grid >= (uint64_t)I->occ_tiles_per_depth * (unsigned)I->depth && !(I->variant & 64)) flags |= F_OCC;
    /* the tree-parallel tail (whitted_tpt.inc): (27 or 29 + weights per light x lights) words per node; the parked lane state and LDS stack levels
     * and the replay's saved sums besides; TPT_SLOTS_PER_XCC slots per XCD, taken and handed back by the waves themselves */
    const uint64_t tpt_per_node = (I->strict ? 29u : 27u) + (I->strict ? 4u : 1u) * (uint64_t)P.nl;
    const uint64_t tpt_fixed = (25u + 36u) * 64u + (uint64_t)I->depth * 3u * 64u;
    const uint64_t tpt_nslots = 8u * (uint64_t)TPT_SLOTS_PER_XCC;
    const uint64_t tpt_slice = std::min<uint64_t>(TPT_SLICE_WORDS_MAX, ((uint64_t)I->tpt_pool_mb << 18) / tpt_nslots) & ~(uint64_t)63;
    const uint64_t tpt_cap = std::min<uint64_t>(tpt_slice > tpt_fixed ? ((tpt_slice - tpt_fixed) / tpt_per_node) & ~(uint64_t)63 : 0, 64960u);   /* whole blocks of 64 nodes; node ids are 16 bits */
    const bool tail_wanted = (flags & F_DEEP) && !(flags & F_OCC) && !(I->variant & 16) && I->tpt_max != 0u && tpt_cap >= TPT_MIN_CAP;
    bool split = false;
    if (P.tiled) {
        if (I->sched && !(I->variant & 4) && trows <= 0xFFFu && tpr <= 0xFFFu) {   /* the order packs (tile column | row << 12 | parts) */
            split = tail_wanted && !(flags & F_GRID) && I->split_min_quota != 0u && !(I->variant & 4096);
            if (split) { per_share_cap = per_share + SPLIT_EXTRA_PER_SHARE; grid = 8 * per_share_cap; }
            P.cost_sum = (tail_wanted && !(flags & F_GRID)) ? 1u : 0u;
            if (S.w != P.width || S.rows != P.rows || S.cap != per_share_cap || !S.cost[0]) {
                S.free_all();
                for (int i = 0; i < 2; i++) {
                    HIP_OK(hipMalloc((void**)&S.cost[i], (size_t)trows * tpr * 4), "Couldn't allocate device memory");
                    HIP_OK(hipMalloc((void**)&S.order[i], (size_t)grid * 4), "Couldn't allocate device memory");
                    HIP_OK(hipEventCreateWithFlags(&S.built[i], hipEventDisableTiming), "Couldn't create a timing event");
                }
                HIP_OK(hipEventCreateWithFlags(&S.traced, hipEventDisableTiming), "Couldn't create a timing event");
                S.w = P.width; S.rows = P.rows; S.cap = per_share_cap; S.wr = 0;
            }
            const int wr = S.wr;
            P.tile_cost = S.cost[wr];
            /* the order built two frames ago is complete by now; the one built behind the previous frame may still be
             * running (waiting for it costs that frame ~15 us, once) */
            int rd = S.have[S.newest] ? S.newest : -1;
            if (rd >= 0 && S.frame - S.newest_frame <= 1 && S.have[rd ^ 1]) rd ^= 1;   /* a still camera ends up on the newest */
            S.frame++;
            /* build(k-2) read cost[wr] and wrote order[wr]: it must have finished before this trace touches either */
            if (S.have[wr]) HIP_OK(hipStreamWaitEvent(I->stream, S.built[wr], 0), "Couldn't run the kernel");
            if (rd >= 0 && rd != wr) HIP_OK(hipStreamWaitEvent(I->stream, S.built[rd], 0), "Couldn't run the kernel");
            P.tile_order = rd >= 0 ? S.order[rd] : nullptr;
            /* the costs can only change when the camera, the depth or the scene did */
            sched_rebuild = !S.sig_valid || !same_raygen(g, S.sig) || S.sig_depth != I->depth || S.sig_scene != I->scene_generation;
            /* (costs that add up: the buffer starts at zero; the build that read it two frames ago has been waited for above) */
            if (P.cost_sum) HIP_OK(hipMemsetAsync(S.cost[wr], 0, (size_t)trows * tpr * 4, I->stream), "Couldn't run the kernel");
            if (sched_rebuild) { S.sig = g; S.sig_depth = I->depth; S.sig_scene = I->scene_generation; S.sig_valid = true; S.sig_age = 0; }
            /* Grid builds measure a tile by its wave's lifetime, which depends on the company it ran in: the order built
             * from the first (unsorted, often cold) frame is refined once from the first sorted one. */
            else if ((flags & F_GRID) && S.sig_age < 2) sched_rebuild = true;
            if (S.sig_age < 255) S.sig_age++;
        }
    }
    /* deep launches: the scratch part of the DFS stack is sized for the launch's depth (7 / 15 / 31 parents) */
    if ((flags & F_DEEP) && !(flags & F_COUNT) && !(I->variant & 2048)) flags |= I->depth <= 8 ? F_D8 : (I->depth <= 16 ? F_D16 : 0);
    if (tail_wanted) {
        const uint64_t nslots = tpt_nslots, slice = tpt_slice, cap = tpt_cap;
        if (cap >= TPT_MIN_CAP) {
            const size_t need = (size_t)slice * 4u * nslots;
            if (I->tpt_pool_bytes != need) {
                finish(I);
                if (I->d_tpt_pool) (void)hipFree(I->d_tpt_pool);
                if (I->d_tpt_flags) (void)hipFree(I->d_tpt_flags);
                I->d_tpt_pool = nullptr; I->d_tpt_flags = nullptr; I->tpt_pool_bytes = 0;
                if (hipMalloc((void**)&I->d_tpt_pool, need) == hipSuccess && hipMalloc((void**)&I->d_tpt_flags, nslots * 4u) == hipSuccess) {
                    HIP_OK(hipMemset(I->d_tpt_flags, 0, nslots * 4u), "Couldn't allocate device memory");
                    I->tpt_pool_bytes = need;
                } else {                                /* no room: the launch runs without the tail */
                    (void)hipGetLastError();
                    if (I->d_tpt_pool) (void)hipFree(I->d_tpt_pool);
                    I->d_tpt_pool = nullptr;
                }
            }
            if (I->tpt_pool_bytes) {
                P.tpt_pool = I->d_tpt_pool; P.tpt_flags = I->d_tpt_flags; P.tpt_slice_words = (uint32_t)slice; P.tpt_cap = (uint32_t)cap;
                P.tpt_slots = TPT_SLOTS_PER_XCC; P.tpt_jump = I->d_tpt_jump;
                P.tpt_max = std::min(I->tpt_max, 64u); P.tpt_min = I->tpt_min;
            }
        }
    }
    LaunchTimer t(I, kid);
    hipError_t e = I->strict ? wt_strict_launch_trace(&P, flags, grid, dyn_lds, I->stream)
                             : wt_fast_launch_trace(&P, flags, grid, dyn_lds, I->stream);
    if (e != hipSuccess) die("Couldn't run the kernel");
    t.done();
    if (P.tile_cost) {
        const int wr = S.wr;
        if (sched_rebuild) {   /* sort this frame's costs behind it, on the side stream */
            if (!I->sched_stream) HIP_OK(hipStreamCreateWithFlags(&I->sched_stream, hipStreamNonBlocking), "Couldn't create a command queue for the given device");
            HIP_OK(hipEventRecord(S.traced, I->stream), "Couldn't run the kernel");
            HIP_OK(hipStreamWaitEvent(I->sched_stream, S.traced, 0), "Couldn't run the kernel");
            if (wt_fast_launch_sched(S.cost[wr], S.order[wr], tpr, trows, per_share, (flags & F_GRID) ? 1u : 0u, per_share_cap,
                                     split ? I->split_slots : 0u, I->split_min_quota, I->sched_stream) != hipSuccess)
                die("Couldn't run the kernel");
            HIP_OK(hipEventRecord(S.built[wr], I->sched_stream), "Couldn't run the kernel");
            S.have[wr] = true; S.newest = wr; S.newest_frame = S.frame - 1;
        }
        S.wr = wr ^ 1;
    }
}

/* cl_wrap_output of a big frame with read-back (what rayinteractive.c does every frame, :183-191): the launch is cut
 * into strips of whole tile rows, and strip c travels to the host while strip c+1 renders.  Same pixels (strips
 * keep their global ids), same blocking semantics -- the call returns when the whole frame is in `host_output`.
 * Returns false when the call is not of that shape; the caller then takes the plain path. */
bool pipelined_output(cl_wrap* w, Impl* I, size_t array_size, size_t output_size, cl_uint kid, cl_uint out_kernel,
                      cl_int out_arg, void* host_output) {
    if (!I->pipeline || I->async || !host_output || !I->fuse || I->counting || I->debug_rgb || (I->variant & 2)) return false;
    if (out_kernel != kid || out_arg != 10 || !is_registered(w, kid, 10)) return false;
    Kernel& k = I->kernels[kid];
    Buffer* rays = nullptr;
    if (is_registered(w, kid, 0)) rays = (Buffer*)w->buffers[kid][0];
    else if (k.values[0].set && k.values[0].size == sizeof(cl_mem)) { void* h; memcpy(&h, k.values[0].bytes, sizeof h); rays = lookup_handle(I, h); }
    if (!rays || !rays->gen_valid || rays->exposed || !k.values[7].set || k.values[7].size != 4) return false;
    const RaygenArgs& g = rays->gen;
    uint32_t total; memcpy(&total, k.values[7].bytes, 4);
    Buffer* out = (Buffer*)w->buffers[kid][10];
    const uint64_t n = g.n_items;
    if (g.band_stride > 1 || g.id_offset % g.width != 0 || n % g.width != 0) return false;
    if (array_size < n || total < n || out->size < n * 4 || output_size != n * 4) return false;   /* the launch covers exactly the generated rays */
    const uint32_t rows = (uint32_t)(n / g.width);
    /* worth it only when the copy is long and the launch is throughput-bound: measured 1.06 -> 0.745 ms at 3840x2160
     * depth 4, +3 % at 1920x1080 (four 2 MB copies are no faster than one of 8 MB), and a LOSS for deep launches,
     * where every strip would pay its own serial tail of refraction trees (0.76 -> 1.30 ms at 1280x1024 depth 15) */
    if (n * 4 < (16u << 20) || rows < 4 * 64 || I->depth > SHALLOW_LEVELS + 1) return false;
    ensure_allocated(I, out);

    const int nch = Impl::MAX_CHUNKS;
    const uint32_t base = rows / nch / 64 * 64;      /* strips of whole 64-row blocks: 8 tile rows, one per XCD */
    if (!I->copy_stream) {
        HIP_OK(hipStreamCreateWithFlags(&I->copy_stream, hipStreamNonBlocking), "Couldn't create a command queue for the given device");
        for (hipEvent_t& e : I->chunk_done) HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming), "Couldn't create a timing event");
    }
    for (int c = 0; c < nch; c++) {
        Strip st{(uint32_t)c * base, c == nch - 1 ? rows - (uint32_t)c * base : base, 1 + c};
        run_raytracer(w, I, kid, array_size, &st);
        HIP_OK(hipEventRecord(I->chunk_done[c], I->stream), "Couldn't run the kernel");
    }
    for (int c = 0; c < nch; c++) {
        const size_t off = (size_t)c * base * g.width * 4;
        const size_t bytes = (c == nch - 1 ? (size_t)(rows - (uint32_t)c * base) : (size_t)base) * g.width * 4;
        HIP_OK(hipStreamWaitEvent(I->copy_stream, I->chunk_done[c], 0), "Failed to transfer device memory to host");
        if (hipMemcpyAsync((uint8_t*)host_output + off, (const uint8_t*)out->dptr + off, bytes, hipMemcpyDeviceToHost, I->copy_stream) != hipSuccess)
            die("Failed to transfer device memory to host");
    }
    hipError_t e = hipStreamSynchronize(I->copy_stream);
    if (e != hipSuccess) { printf("%d\n", (int)e); die("The device kernel failed"); }
    finish(I);
    return true;
}

void run_raygen(cl_wrap* w, Impl* I, cl_uint kid, size_t array_size) {
    Kernel& k = I->kernels[kid];
    Buffer* rays = buffer_arg(w, kid, 8);
    RaygenArgs g = snapshot_raygen(I, k, array_size);
    rays->gen = g;
    rays->gen_valid = true;
    rays->gen_materialised = false;
    rays->exposed = false;
    if (!I->fuse) {
        run_raygen_kernel(I, g, rays, kid);
        rays->gen_materialised = true;
    }
}


} /* namespace */

/* ======================================================================================== */
extern "C" {

void cl_wrap_init(cl_wrap* wrap, cl_device_type type, ...) {
    memset(wrap, 0, sizeof *wrap);
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) die("Cannot find a HIP platform");
    if (!(type & (CL_DEVICE_TYPE_GPU | CL_DEVICE_TYPE_DEFAULT)) || count <= 0)
        die("Cannot find a device of the given type");

    Impl* I = new Impl();
    int cur = 0;
    if (hipGetDevice(&cur) != hipSuccess) cur = 0;
    I->device = env_int("CLWRAP_DEVICE", cur);
    if (I->device < 0 || I->device >= count) die("Cannot find a device of the given type");
    HIP_OK(hipSetDevice(I->device), "Could not create a HIP context from device");

    va_list vars;
    va_start(vars, type);
    const char* path = va_arg(vars, const char*);
    while (path) {
        const char* name = va_arg(vars, const char*);
        if (!name) { va_end(vars); die("Source file was not followed by kernel name"); }
        if (I->kernels.size() >= __MAX_KERNELS) { va_end(vars); die("Too many kernels"); }
        Kernel k;
        k.name = name;
        /* kernels are precompiled: identity comes from the NAME, the source path is not read */
        if (k.name == "raygen") k.kind = K_RAYGEN;
        else if (k.name == "raytracer") k.kind = K_RAYTRACER;
        else { va_end(vars); die("Couldn't create the CL kernel from: %s", name); }
        I->kernels.push_back(k);
        path = va_arg(vars, const char*);
    }
    va_end(vars);

    HIP_OK(hipStreamCreateWithFlags(&I->own_stream, hipStreamNonBlocking), "Couldn't create a command queue for the given device");
    I->stream = I->own_stream;
    I->depth = env_int("CLWRAP_DEPTH", 15);
    if (I->depth < 1 || I->depth > CLW_MAX_DEPTH) die("CLWRAP_DEPTH must be in [1, %d]", CLW_MAX_DEPTH);
    I->strict = env_int("CLWRAP_STRICT", 0) ? 1 : 0;
    I->fuse = env_int("CLWRAP_FUSE", 1) ? 1 : 0;
    I->variant = env_int("CLWRAP_VARIANT", 0);
    I->timing_every = (uint32_t)env_int("CLWRAP_TIMING_EVERY", 1);
    if (I->timing_every == 0) I->timing_every = 1;
    I->pipeline = env_int("CLWRAP_PIPELINE", 1) ? 1 : 0;
    I->stamps = env_int("CLWRAP_STAMPS", 0) ? 1 : 0;
    if (const char* th = getenv("CLWRAP_THROUGH")) { if (*th) I->through = (float)atof(th); }
    I->occ_tiles_per_depth = (unsigned)env_int("CLWRAP_OCC_TILES_PER_DEPTH", (int)OCC_TILES_PER_DEPTH);
    I->tpt_max = (unsigned)env_int("CLWRAP_TPT_MAX", (int)TPT_MAX_LANES);
    I->tpt_min = (unsigned)env_int("CLWRAP_TPT_MIN", (int)TPT_MIN_PATHS);
    I->tpt_pool_mb = (unsigned)env_int("CLWRAP_TPT_POOL_MB", (int)TPT_POOL_MB);
    I->split_min_quota = (unsigned)env_int("CLWRAP_SPLIT_MIN_QUOTA", (int)SPLIT_MIN_QUOTA);
    I->tpt_clock = env_int("CLWRAP_TPT_CLOCK", 0) ? 1 : 0;
    I->split_slots = (unsigned)std::max(1, env_int("CLWRAP_SPLIT_SLOTS", (int)SPLIT_SLOTS));

    wrap->impl = I;
    wrap->kernels_num = (cl_uint)I->kernels.size();
}

void cl_wrap_load_global_data(cl_wrap* wrap, cl_uint kernel_id, cl_uint arg_id, const void* data,
                              size_t size, cl_mem_flags mem_flags) {
    (void)mem_flags;
    Impl* I = impl_of(wrap);
    use_device(I);
    precheck_buffer_arg(wrap, kernel_id, arg_id);
    Buffer* b = new_buffer(I);
    b->size = size;
    if (data) {
        ensure_allocated(I, b);
        if (size) {
            if (hipMemcpy(b->dptr, data, size, hipMemcpyHostToDevice) != hipSuccess)
                die("Couldn't transfer the data from host to the device");
            if (size <= (64u << 20)) b->shadow.assign((const uint8_t*)data, (const uint8_t*)data + size);
        }
    } else {
        b->lazy = true; /* e.g. the 64 B/pixel ray buffer: stays virtual unless somebody needs the bytes */
    }
    register_buffer(wrap, kernel_id, arg_id, b);
}

void cl_wrap_load_single_data(cl_wrap* wrap, cl_uint kernel_id, cl_uint arg_id, const void* data, size_t obj_size) {
    Impl* I = impl_of(wrap);
    check_kernel_id(wrap, kernel_id);
    if (arg_id < __MAX_BUFFERS && is_registered(wrap, kernel_id, arg_id)) die("Given kernel argument already in use");
    if (arg_id >= __MAX_BUFFERS || !data || obj_size == 0 || obj_size > sizeof(ArgValue::bytes))
        die("Couldn't pass the data argument to the kernel");
    ArgValue& v = I->kernels[kernel_id].values[arg_id];
    memcpy(v.bytes, data, obj_size); /* copied at call time, like clSetKernelArg */
    v.size = obj_size;
    v.set = true;
}

static void install_images(cl_wrap* wrap, Impl* I, cl_uint kernel_id, cl_uint arg_id, const uint8_t* rgba,
                           uint32_t w, uint32_t h, uint32_t layers) {
    Buffer* b = new_buffer(I);
    b->size = (size_t)w * h * layers * 4;
    b->image = true; b->w = w; b->h = h; b->layers = layers;
    if (b->size == 0) die("Couldn't create an image array %d", -40);
    ensure_allocated(I, b);
    if (hipMemcpy(b->dptr, rgba, b->size, hipMemcpyHostToDevice) != hipSuccess)
        die("Couldn't create an image array %d", -1);
    register_buffer(wrap, kernel_id, arg_id, b);
}

void cl_wrap_load_images(cl_wrap* wrap, cl_uint kernel_id, cl_uint arg_id, cl_mem_flags mem_flags,
                         cl_uint image_num, ...) {
    (void)mem_flags;
    Impl* I = impl_of(wrap);
    use_device(I);
    precheck_buffer_arg(wrap, kernel_id, arg_id);
    std::vector<uint8_t> all;
    uint32_t W = 0, H = 0;
    va_list vars;
    va_start(vars, image_num);
    for (cl_uint i = 0; i < image_num; i++) {
        const char* filename = va_arg(vars, const char*);
        uint32_t w = 0, h = 0;
        uint8_t* px = nullptr;
        int rc = wpng_read_rgba(filename, &w, &h, &px);
        switch (rc) { /* the reference's messages (opencl_wrap.c:233-307) */
            case WPNG_OK: break;
            case WPNG_ERR_OPEN: va_end(vars); die("Cannot open file \"%s\"", filename);
            case WPNG_ERR_NOT_PNG: va_end(vars); die("\"%s\" is not a PNG file", filename);
            case WPNG_ERR_FORMAT: va_end(vars); die("\"%s\" must have a depth of 8 bits and be RGB", filename);
            default: va_end(vars); die("Could not decode PNG file \"%s\" (code %d)", filename, rc);
        }
        if (i == 0) { W = w; H = h; all.reserve((size_t)w * h * 4 * image_num); }
        if (w != W || h != H) { free(px); va_end(vars); die("All images must have same dimensions"); }
        all.insert(all.end(), px, px + (size_t)w * h * 4);
        free(px);
    }
    va_end(vars);
    install_images(wrap, I, kernel_id, arg_id, all.data(), W, H, image_num);
}

void cl_wrap_output(cl_wrap* wrap, size_t array_size, size_t output_size, cl_uint kernel_run_id,
                    cl_uint kernel_id, cl_int arg_id, void* host_output) {
    Impl* I = impl_of(wrap);
    use_device(I);
    check_kernel_id(wrap, kernel_run_id);
    if (I->kernels[kernel_run_id].kind == K_RAYGEN) run_raygen(wrap, I, kernel_run_id, array_size);
    else if (pipelined_output(wrap, I, array_size, output_size, kernel_run_id, kernel_id, arg_id, host_output)) return;
    else run_raytracer(wrap, I, kernel_run_id, array_size);

    if (!host_output) {
        if (!I->async) finish(I);
        return;
    }
    /* blocking device -> host copy of buffers[kernel_id][arg_id] (opencl_wrap.c:390-397) */
    check_kernel_id(wrap, kernel_id);
    if (arg_id < 0 || arg_id >= __MAX_BUFFERS || !is_registered(wrap, kernel_id, (cl_uint)arg_id))
        die("Failed to transfer device memory to host");
    Buffer* b = (Buffer*)wrap->buffers[kernel_id][arg_id];
    materialise_rays(I, b);
    ensure_allocated(I, b);
    if (output_size > b->size) die("Failed to transfer device memory to host");
    if (output_size &&
        hipMemcpyAsync(host_output, b->dptr, output_size, hipMemcpyDeviceToHost, I->stream) != hipSuccess)
        die("Failed to transfer device memory to host");
    finish(I);
}

void cl_wrap_release(cl_wrap* wrap) {
    if (!wrap || !wrap->impl) return;
    Impl* I = (Impl*)wrap->impl;
    use_device(I);
    (void)hipStreamSynchronize(I->stream);
    for (Buffer* b : I->live) {
        if (b->owned && b->dptr) (void)hipFree(b->dptr);
        delete b;
    }
    if (I->d_geom) (void)hipFree(I->d_geom);
    if (I->d_ptex) (void)hipFree(I->d_ptex);
    if (I->d_counters) (void)hipFree(I->d_counters);
    if (I->d_tpt_pool) (void)hipFree(I->d_tpt_pool);
    if (I->d_tpt_flags) (void)hipFree(I->d_tpt_flags);
    if (I->d_tpt_jump) (void)hipFree(I->d_tpt_jump);
    for (uint32_t* q : {I->d_grid_start, I->d_grid_items, I->d_grid_box}) if (q) (void)hipFree(q);
    if (I->d_grid_geom) (void)hipFree(I->d_grid_geom);
    if (I->sched_stream) (void)hipStreamSynchronize(I->sched_stream);
    for (auto& sc : I->scheds) sc.free_all();
    if (I->sched_stream) (void)hipStreamDestroy(I->sched_stream);
    if (I->copy_stream) (void)hipStreamDestroy(I->copy_stream);
    for (hipEvent_t e : I->chunk_done) if (e) (void)hipEventDestroy(e);
    for (auto& t : I->timing) { (void)hipEventDestroy(t.start); (void)hipEventDestroy(t.stop); }
    for (auto& p : I->free_events) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    if (I->own_stream) (void)hipStreamDestroy(I->own_stream);
    delete I;
    memset(wrap, 0, sizeof *wrap);
}

/* ---------------------------------- extensions ------------------------------------------ */
void clw_ext_set_depth(cl_wrap* wrap, int depth) {
    if (depth < 1 || depth > CLW_MAX_DEPTH) die("Trace depth must be in [1, %d]", CLW_MAX_DEPTH);
    impl_of(wrap)->depth = depth;
}
int clw_ext_get_depth(const cl_wrap* wrap) { return impl_of(wrap)->depth; }
void clw_ext_set_strict(cl_wrap* wrap, int strict) { impl_of(wrap)->strict = strict ? 1 : 0; }
void clw_ext_set_fuse(cl_wrap* wrap, int fuse) { impl_of(wrap)->fuse = fuse ? 1 : 0; }
void clw_ext_set_id_offset(cl_wrap* wrap, uint64_t first_id) { impl_of(wrap)->id_offset = first_id; }
void clw_ext_set_row_bands(cl_wrap* wrap, uint32_t stride, uint32_t phase) {
    if (stride == 0 || phase >= stride) die("Row bands need 0 <= phase < stride");
    Impl* I = impl_of(wrap);
    I->band_stride = stride; I->band_phase = phase;
}
void clw_ext_set_async(cl_wrap* wrap, int async) { impl_of(wrap)->async = async ? 1 : 0; }
void clw_ext_sync(cl_wrap* wrap) { Impl* I = impl_of(wrap); use_device(I); finish(I); }
void clw_ext_set_stream(cl_wrap* wrap, void* hip_stream) {
    Impl* I = impl_of(wrap);
    I->stream = hip_stream ? (hipStream_t)hip_stream : I->own_stream;
}
void clw_ext_unit(cl_wrap* wrap, int op, const float* in, uint32_t stride_in, float* out, uint32_t stride_out, uint32_t n,
                  uint32_t aux) {
    Impl* I = impl_of(wrap);
    use_device(I);
    if (n == 0) return;
    float *din = nullptr, *dout = nullptr;
    HIP_OK(hipMalloc((void**)&din, (size_t)n * stride_in * 4), "Couldn't allocate device memory");
    HIP_OK(hipMalloc((void**)&dout, (size_t)n * stride_out * 4), "Couldn't allocate device memory");
    HIP_OK(hipMemcpy(din, in, (size_t)n * stride_in * 4, hipMemcpyHostToDevice), "Couldn't transfer the data from host to the device");
    HIP_OK(hipMemset(dout, 0, (size_t)n * stride_out * 4), "Couldn't allocate device memory");
    HIP_OK(hipDeviceSynchronize(), "The device kernel failed");
    hipError_t e = I->strict ? wt_strict_launch_unit(op, din, dout, n, stride_in, stride_out, aux, I->stream)
                             : wt_fast_launch_unit(op, din, dout, n, stride_in, stride_out, aux, I->stream);
    if (e != hipSuccess) die("Couldn't run the kernel");
    finish(I);
    HIP_OK(hipMemcpy(out, dout, (size_t)n * stride_out * 4, hipMemcpyDeviceToHost), "Failed to transfer device memory to host");
    (void)hipFree(din); (void)hipFree(dout);
}
void clw_ext_unit_scene(cl_wrap* wrap, cl_uint kernel_id, int op, const float* in, uint32_t stride_in, float* out,
                        uint32_t stride_out, uint32_t n) {
    Impl* I = impl_of(wrap);
    use_device(I);
    check_kernel_id(wrap, kernel_id);
    if (I->kernels[kernel_id].kind != K_RAYTRACER) die("Wrong kernel ID given");
    if (n == 0) return;
    whitted_params P{};
    int flags = 0;
    size_t dyn_lds = 0;
    bind_scene(wrap, I, kernel_id, P, flags, dyn_lds);
    P.depth = I->depth;
    float *din = nullptr, *dout = nullptr;
    HIP_OK(hipMalloc((void**)&din, (size_t)n * stride_in * 4), "Couldn't allocate device memory");
    HIP_OK(hipMalloc((void**)&dout, (size_t)n * stride_out * 4), "Couldn't allocate device memory");
    HIP_OK(hipMemcpy(din, in, (size_t)n * stride_in * 4, hipMemcpyHostToDevice), "Couldn't transfer the data from host to the device");
    HIP_OK(hipMemset(dout, 0, (size_t)n * stride_out * 4), "Couldn't allocate device memory");
    HIP_OK(hipDeviceSynchronize(), "The device kernel failed");
    hipError_t e = I->strict ? wt_strict_launch_unit_scene(&P, flags, op, din, dout, n, stride_in, stride_out, dyn_lds, I->stream)
                             : wt_fast_launch_unit_scene(&P, flags, op, din, dout, n, stride_in, stride_out, dyn_lds, I->stream);
    if (e != hipSuccess) die("Couldn't run the kernel");
    finish(I);
    HIP_OK(hipMemcpy(out, dout, (size_t)n * stride_out * 4, hipMemcpyDeviceToHost), "Failed to transfer device memory to host");
    (void)hipFree(din); (void)hipFree(dout);
}
uint32_t clw_ext_read_tile_costs(cl_wrap* wrap, uint32_t* out, uint32_t capacity) {
    Impl* I = impl_of(wrap);
    use_device(I);
    finish(I);
    const Impl::Sched& S = I->scheds[0];
    if (!S.cost[0]) return 0;
    uint32_t n = ((S.rows + 7) / 8) * ((S.w + 7) / 8);
    if (out && capacity >= n) HIP_OK(hipMemcpy(out, S.cost[S.wr ^ 1], (size_t)n * 4, hipMemcpyDeviceToHost), "Failed to transfer device memory to host");
    return n;
}
void clw_ext_set_shadow_through(cl_wrap* wrap, float factor) { impl_of(wrap)->through = factor; }
void clw_ext_set_grid(cl_wrap* wrap, int on) { impl_of(wrap)->use_grid = on ? 1 : 0; }
void clw_ext_set_tile_sched(cl_wrap* wrap, int on) { Impl* I = impl_of(wrap); I->sched = on ? 1 : 0; for (auto& sc : I->scheds) sc.reset(); }
void clw_ext_set_variant(cl_wrap* wrap, int variant) { impl_of(wrap)->variant = variant; }
void clw_ext_set_tpt(cl_wrap* wrap, int max_lanes, int min_paths, int pool_mb) {
    Impl* I = impl_of(wrap);
    if (max_lanes >= 0) I->tpt_max = (unsigned)std::min(max_lanes, 64);
    if (min_paths >= 0) I->tpt_min = (unsigned)min_paths;
    if (pool_mb >= 0) I->tpt_pool_mb = (unsigned)pool_mb;
}
void clw_ext_set_debug_rgb(cl_wrap* wrap, void* p) { impl_of(wrap)->debug_rgb = (float*)p; }

void clw_ext_set_pipeline(cl_wrap* wrap, int on) { impl_of(wrap)->pipeline = on ? 1 : 0; }
void clw_ext_set_timing_every(cl_wrap* wrap, uint32_t n) { Impl* I = impl_of(wrap); I->timing_every = n ? n : 1; I->timing_tick = 0; I->timing_on = n != 0; }

void clw_ext_timing_reset(cl_wrap* wrap) {
    Impl* I = impl_of(wrap);
    use_device(I);
    finish(I);
    for (auto& t : I->timing) I->free_events.push_back({t.start, t.stop});
    I->timing.clear();
    I->timing_on = true;
}

void clw_ext_timing_get(cl_wrap* wrap, cl_uint kernel_id, uint32_t* launches, double* total_ms) {
    Impl* I = impl_of(wrap);
    use_device(I);
    finish(I);
    uint32_t n = 0;
    double sum = 0.0;
    for (auto& t : I->timing) {
        if (t.kernel != kernel_id) continue;
        float ms = 0.0f;
        HIP_OK(hipEventElapsedTime(&ms, t.start, t.stop), "Couldn't read a timing event");
        sum += ms;
        n++;
    }
    if (launches) *launches = n;
    if (total_ms) *total_ms = sum;
}

void clw_ext_load_images_raw(cl_wrap* wrap, cl_uint kernel_id, cl_uint arg_id, const uint8_t* rgba,
                             uint32_t width, uint32_t height, uint32_t layers) {
    Impl* I = impl_of(wrap);
    use_device(I);
    precheck_buffer_arg(wrap, kernel_id, arg_id);
    if (!rgba) die("Couldn't create an image array %d", -37);
    install_images(wrap, I, kernel_id, arg_id, rgba, width, height, layers);
}

void clw_ext_bind_device_buffer(cl_wrap* wrap, cl_uint kernel_id, cl_uint arg_id, void* device_ptr, size_t size) {
    Impl* I = impl_of(wrap);
    precheck_buffer_arg(wrap, kernel_id, arg_id);
    if (!device_ptr) die("Couldn't pass the data argument to the kernel");
    Buffer* b = new_buffer(I);
    b->dptr = device_ptr; b->size = size; b->owned = false;
    register_buffer(wrap, kernel_id, arg_id, b);
}

void clw_ext_invalidate_scene(cl_wrap* wrap) {
    Impl* I = impl_of(wrap);
    I->prep_s = I->prep_p = I->prep_l = nullptr;
    /* a scene buffer rewritten in place on the device no longer matches its host copy */
    for (Buffer* b : I->live) if (!b->image) b->shadow.clear();
}

void* clw_ext_device_ptr(cl_wrap* wrap, cl_uint kernel_id, cl_uint arg_id) {
    Impl* I = impl_of(wrap);
    use_device(I);
    check_kernel_id(wrap, kernel_id);
    if (arg_id >= __MAX_BUFFERS || !is_registered(wrap, kernel_id, arg_id)) die("Wrong kernel ID given");
    Buffer* b = (Buffer*)wrap->buffers[kernel_id][arg_id];
    materialise_rays(I, b);
    ensure_allocated(I, b);
    if (b->gen_valid) b->exposed = true;
    return b->dptr;
}

void clw_ext_enable_counters(cl_wrap* wrap, int enable) { impl_of(wrap)->counting = enable ? 1 : 0; }

void clw_ext_read_counters_ex(cl_wrap* wrap, uint64_t* out, uint32_t n) {
    Impl* I = impl_of(wrap);
    use_device(I);
    finish(I);
    for (uint32_t k = 0; k < n; k++) out[k] = 0;
    if (!I->d_counters) return;
    std::vector<unsigned long long> h(COUNTER_WORDS);
    HIP_OK(hipMemcpy(h.data(), I->d_counters, COUNTER_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToHost), "Failed to transfer device memory to host");
    HIP_OK(hipMemsetAsync(I->d_counters, 0, COUNTER_WORDS * sizeof(unsigned long long), I->stream), "Couldn't allocate device memory");
    finish(I);
    for (size_t sh = 0; sh < CLW_STAMP_SHARDS; sh++)        /* stamp shards of the diagnostic build -> words 16.. */
        for (int k = 0; k < 16; k++) h[16 + k] += h[CLW_NUM_COUNTERS + 16 * sh + k];
    for (uint32_t k = 0; k < n && k < CLW_NUM_COUNTERS; k++) out[k] = h[k];
}
void clw_ext_read_counters(cl_wrap* wrap, uint64_t out[8]) { clw_ext_read_counters_ex(wrap, out, 8); }

int clw_host_write_png(const char* path, const uint32_t* xrgb, uint32_t width, uint32_t height) {
    return wpng_write_xrgb(path, xrgb, width, height, 1);
}
int clw_host_write_png_rgba(const char* path, const uint8_t* rgba, uint32_t width, uint32_t height) {
    return wpng_write_rgba_as_rgb(path, rgba, width, height, 1);
}
int clw_host_read_png(const char* path, uint32_t* width, uint32_t* height, uint8_t** rgba_malloced) {
    return wpng_read_rgba(path, width, height, rgba_malloced);
}
void clw_host_free(void* p) { free(p); }

#ifndef WT_SOURCE_HASH
#define WT_SOURCE_HASH "unknown"
#endif
const char* clw_ext_version(void) { return "opencl_wrap_hip 0.2 gfx950 fast+strict kernels:" WT_SOURCE_HASH; }

} /* extern "C" */
