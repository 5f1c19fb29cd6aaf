/*
 * raybench.c -- pure-C bench driver: the reference's raypng.c call sequence (raypng.c:31-100) with the frame
 * size, depth, scene and frame count taken from the command line instead of #defines (the
 * reference drivers hard-code 800x600 and a cl_uint byte count that overflows at 8192x8192).
 * Host code is C99 and talks to the GPU only through include/opencl_wrap.h + include/hip_wrap_ext.h.
 *
 *   raybench [-w W] [-h H] [-d depth] [-n frames] [-s scene.map] [-a assets_dir] [-o out.png] [-r] [-S]
 *     -s  scene archive: the reference's render.map format or the extended wide-count format (scene.py)
 *     -a  directory holding cobblestone.png sand.png check.png grass.png bg/stormydays.png (raypng.c:74-81)
 *     -r  include the blocking framebuffer read-back in every frame (what rayinteractive's loop does)
 *     -S  strict arithmetic build
 * Prints one JSON line with kernel-only and per-frame wall times.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include "hip_wrap_ext.h"

static double now_ms(void) { struct timeval tv; gettimeofday(&tv, NULL); return tv.tv_sec * 1e3 + tv.tv_usec * 1e-3; }

/* scene archive reader: legacy = u8 count + structs, three times (reference src/cpu_obj.c:76-101);
 * extended = "\xffRMAPv2\0" + 3 x u32 counts + the same struct arrays */
static int load_scene(const char* path, void** sph, uint32_t* ns, void** pln, uint32_t* np, void** lgt, uint32_t* nl) {
    FILE* fp = fopen(path, "rb");
    if (!fp) return 0;
    unsigned char magic[8];
    static const unsigned char ext[8] = {0xff, 'R', 'M', 'A', 'P', 'v', '2', 0};
    size_t got = fread(magic, 1, 8, fp);
    int is_ext = got == 8 && memcmp(magic, ext, 8) == 0;
    uint32_t cnt[3];
    if (is_ext) { if (fread(cnt, 4, 3, fp) != 3) { fclose(fp); return 0; } }
    else rewind(fp);
    void** dst[3] = {sph, pln, lgt};
    uint32_t* n[3] = {ns, np, nl};
    const size_t sz[3] = {96, 96, 48};
    for (int k = 0; k < 3; k++) {
        uint32_t c;
        if (is_ext) c = cnt[k];
        else { unsigned char b; if (fread(&b, 1, 1, fp) != 1) { fclose(fp); return 0; } c = b; }
        *dst[k] = malloc(sz[k] * (c ? c : 1));
        if (fread(*dst[k], sz[k], c, fp) != c) { fclose(fp); return 0; }
        *n[k] = c;
    }
    fclose(fp);
    return 1;
}

int main(int argc, char** argv) {
    uint32_t W = 1920, H = 1080;
    int depth = 4, frames = 100, readback = 0, strict = 0;
    const char *scene = "scenes/render.map", *assets = "assets", *out = NULL;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "-w") && i + 1 < argc) W = (uint32_t)atoi(argv[++i]);
        else if (!strcmp(argv[i], "-h") && i + 1 < argc) H = (uint32_t)atoi(argv[++i]);
        else if (!strcmp(argv[i], "-d") && i + 1 < argc) depth = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-n") && i + 1 < argc) frames = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-s") && i + 1 < argc) scene = argv[++i];
        else if (!strcmp(argv[i], "-a") && i + 1 < argc) assets = argv[++i];
        else if (!strcmp(argv[i], "-o") && i + 1 < argc) out = argv[++i];
        else if (!strcmp(argv[i], "-r")) readback = 1;
        else if (!strcmp(argv[i], "-S")) strict = 1;
        else { fprintf(stderr, "usage: raybench [-w W] [-h H] [-d depth] [-n frames] [-s scene] [-a assets] [-o out.png] [-r] [-S]\n"); return 2; }
    }
    void *sph, *pln, *lgt;
    uint32_t ns, np, nl;
    if (!load_scene(scene, &sph, &ns, &pln, &np, &lgt, &nl)) { fprintf(stderr, "cannot read scene %s\n", scene); return 1; }

    const float origin[3] = {0.8f, 2.5f, -8.0f}, look[3] = {0.2f, 0.0f, 1.0f};     /* raypng.c:17-21 */
    clw_camera cam;
    if (!clw_host_perspective(origin, look, 90.0f, 1.0f, W, H, &cam)) { fprintf(stderr, "bad camera\n"); return 1; }

    cl_wrap w;
    cl_wrap_init(&w, CL_DEVICE_TYPE_GPU, "src/cl/raygen.cl", "raygen", "src/cl/raytracing.cl", "raytracer", NULL);
    clw_ext_set_depth(&w, depth);
    clw_ext_set_strict(&w, strict);

    const size_t pixels = (size_t)W * H;                 /* size_t: 64 B x 8192^2 overflows cl_uint (raypng.c:44) */
    const size_t buffer_size = pixels * sizeof(cl_uint);
    cl_uint* buffer = (cl_uint*)malloc(buffer_size);
    cl_float3 v;
    memset(&v, 0, sizeof v);
    memcpy(&v, cam.im_corner, 12); cl_wrap_load_single_data(&w, 0, 0, &v, sizeof v);
    memcpy(&v, cam.origin, 12);    cl_wrap_load_single_data(&w, 0, 1, &v, sizeof v);
    memcpy(&v, cam.up, 12);        cl_wrap_load_single_data(&w, 0, 2, &v, sizeof v);
    memcpy(&v, cam.right, 12);     cl_wrap_load_single_data(&w, 0, 3, &v, sizeof v);
    cl_wrap_load_single_data(&w, 0, 4, &cam.w_factor, sizeof(cl_float));
    cl_wrap_load_single_data(&w, 0, 5, &cam.h_factor, sizeof(cl_float));
    cl_wrap_load_single_data(&w, 0, 6, &W, sizeof(cl_uint));
    cl_wrap_load_single_data(&w, 0, 7, &H, sizeof(cl_uint));
    cl_wrap_load_global_data(&w, 0, 8, NULL, 64 * pixels, CL_MEM_READ_WRITE);

    cl_wrap_load_single_data(&w, 1, 0, &w.buffers[0][8], sizeof(cl_mem));
    cl_wrap_load_global_data(&w, 1, 1, sph, 96 * (size_t)ns, CL_MEM_READ_ONLY);
    cl_wrap_load_global_data(&w, 1, 2, pln, 96 * (size_t)np, CL_MEM_READ_ONLY);
    cl_wrap_load_global_data(&w, 1, 3, lgt, 48 * (size_t)nl, CL_MEM_READ_ONLY);
    if (ns < 256 && np < 256 && nl < 256) {              /* one-byte counts like the reference */
        cl_uchar a = (cl_uchar)ns, b = (cl_uchar)np, c = (cl_uchar)nl;
        cl_wrap_load_single_data(&w, 1, 4, &a, 1); cl_wrap_load_single_data(&w, 1, 5, &b, 1); cl_wrap_load_single_data(&w, 1, 6, &c, 1);
    } else {                                             /* wide-count extension */
        cl_wrap_load_single_data(&w, 1, 4, &ns, 4); cl_wrap_load_single_data(&w, 1, 5, &np, 4); cl_wrap_load_single_data(&w, 1, 6, &nl, 4);
    }
    cl_uint total = (cl_uint)pixels;
    cl_wrap_load_single_data(&w, 1, 7, &total, sizeof(cl_uint));
    char p[5][1024];
    const char* names[5] = {"cobblestone.png", "sand.png", "check.png", "grass.png", "bg/stormydays.png"};
    for (int k = 0; k < 5; k++) snprintf(p[k], sizeof p[k], "%s/%s", assets, names[k]);
    cl_wrap_load_images(&w, 1, 8, CL_MEM_COPY_HOST_PTR, 4, p[0], p[1], p[2], p[3]);
    cl_wrap_load_images(&w, 1, 9, CL_MEM_COPY_HOST_PTR, 1, p[4]);
    cl_wrap_load_global_data(&w, 1, 10, NULL, buffer_size, CL_MEM_WRITE_ONLY);

    /* warm-up (first frame also builds the prepared scene and the tile order) */
    for (int k = 0; k < 3; k++) { cl_wrap_output(&w, pixels, 0, 0, 0, 0, NULL); cl_wrap_output(&w, pixels, 0, 1, 1, 10, NULL); }
    clw_ext_timing_reset(&w);
    if (!readback) clw_ext_set_async(&w, 1);
    double t0 = now_ms();
    for (int k = 0; k < frames; k++) {
        cl_wrap_output(&w, pixels, 0, 0, 0, 0, NULL);
        if (readback) cl_wrap_output(&w, pixels, buffer_size, 1, 1, 10, buffer);
        else cl_wrap_output(&w, pixels, 0, 1, 1, 10, NULL);
    }
    clw_ext_sync(&w);
    double wall = (now_ms() - t0) / frames;
    uint32_t launches; double kms;
    clw_ext_timing_get(&w, 1, &launches, &kms);
    clw_ext_set_async(&w, 0);
    cl_wrap_output(&w, pixels, 0, 0, 0, 0, NULL);
    cl_wrap_output(&w, pixels, buffer_size, 1, 1, 10, buffer);
    printf("{\"driver\": \"raybench.c\", \"frame\": \"%ux%u\", \"depth\": %d, \"spheres\": %u, \"planes\": %u, \"lights\": %u, "
           "\"frames\": %d, \"readback\": %d, \"strict\": %d, \"trace_kernel_ms\": %.4f, \"wall_ms_per_frame\": %.4f, \"frames_per_s\": %.1f}\n",
           W, H, depth, ns, np, nl, frames, readback, strict, kms / (launches ? launches : 1), wall, 1e3 / wall);
    if (out && clw_host_write_png(out, buffer, W, H) != 0) fprintf(stderr, "cannot write %s\n", out);
    cl_wrap_release(&w);
    free(sph); free(pln); free(lgt); free(buffer);
    return 0;
}
