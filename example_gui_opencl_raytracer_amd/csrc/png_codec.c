/*
 * png_codec.c -- minimal PNG reader / writer over zlib (no libpng: its development
 * headers are not guaranteed on the GPU box).
 *
 * Reader: exactly the subset the reference's cl_wrap_load_images accepts
 * (reference src/opencl_wrap.c:189-349): 8-bit, colour type RGB; the decoded rows get
 * an A = 255 filler like png_set_filler(.., 255, PNG_FILLER_AFTER) (opencl_wrap.c:309).
 * Anything else is reported to the caller, who prints the reference's message.
 * Writer: what the reference's png_dump produces (src/cpu_ray.c:108-165): 8-bit RGB from
 * a 0x00RRGGBB framebuffer.
 */
#include "png_codec.h"
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

static const unsigned char k_sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};

static uint32_t be32(const unsigned char* p) {
    return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
}

static int paeth(int a, int b, int c) {
    int p = a + b - c;
    int pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    if (pa <= pb && pa <= pc) return a;
    return pb <= pc ? b : c;
}

int wpng_read_rgba(const char* path, uint32_t* width, uint32_t* height, uint8_t** rgba_out) {
    FILE* fp = fopen(path, "rb");
    if (!fp) return WPNG_ERR_OPEN;
    unsigned char hdr[8];
    if (fread(hdr, 1, 8, fp) != 8 || memcmp(hdr, k_sig, 8) != 0) { fclose(fp); return WPNG_ERR_NOT_PNG; }

    uint32_t w = 0, h = 0;
    int have_ihdr = 0, done = 0, rc = WPNG_OK;
    unsigned char* idat = NULL;
    size_t idat_len = 0, idat_cap = 0;
    while (!done) {
        unsigned char ch[8];
        if (fread(ch, 1, 8, fp) != 8) { rc = WPNG_ERR_CORRUPT; break; }
        uint32_t len = be32(ch);
        if (!memcmp(ch + 4, "IHDR", 4)) {
            unsigned char ih[13];
            if (len != 13 || fread(ih, 1, 13, fp) != 13) { rc = WPNG_ERR_CORRUPT; break; }
            w = be32(ih); h = be32(ih + 4);
            if (ih[8] != 8 || ih[9] != 2) { rc = WPNG_ERR_FORMAT; break; }      /* depth 8, colour type RGB */
            if (ih[10] != 0 || ih[11] != 0 || ih[12] != 0) { rc = WPNG_ERR_UNSUPPORTED; break; } /* interlace */
            /* the PNG limit is 2^31 - 1 per side; the products below must also fit size_t and zlib's uLong */
            if (w == 0 || h == 0 || w > 0x7FFFFFFFu || h > 0x7FFFFFFFu) { rc = WPNG_ERR_CORRUPT; break; }
            if ((size_t)h > SIZE_MAX / ((size_t)w * 4 + 1)) { rc = WPNG_ERR_CORRUPT; break; }
            have_ihdr = 1;
            fseek(fp, 4, SEEK_CUR);
        } else if (!memcmp(ch + 4, "IDAT", 4)) {
            if (!have_ihdr || len > 0x7FFFFFFFu) { rc = WPNG_ERR_CORRUPT; break; }
            if (idat_len + len > idat_cap) {
                idat_cap = (idat_len + len) * 2 + 4096;
                unsigned char* n = (unsigned char*)realloc(idat, idat_cap);
                if (!n) { rc = WPNG_ERR_NOMEM; break; }
                idat = n;
            }
            if (fread(idat + idat_len, 1, len, fp) != len) { rc = WPNG_ERR_CORRUPT; break; }
            idat_len += len;
            fseek(fp, 4, SEEK_CUR);
        } else if (!memcmp(ch + 4, "IEND", 4)) {
            done = 1;
        } else {
            if (len > 0x7FFFFFFFu || fseek(fp, (long)len + 4, SEEK_CUR) != 0) { rc = WPNG_ERR_CORRUPT; break; }
        }
    }
    fclose(fp);
    if (rc == WPNG_OK && (!have_ihdr || idat_len == 0)) rc = WPNG_ERR_CORRUPT;
    if (rc != WPNG_OK) { free(idat); return rc; }

    const size_t stride = (size_t)w * 3;
    const size_t raw_len = (stride + 1) * (size_t)h;
    unsigned char* raw = (unsigned char*)malloc(raw_len);
    uint8_t* rgba = (uint8_t*)malloc((size_t)w * h * 4);
    if (!raw || !rgba) { free(raw); free(rgba); free(idat); return WPNG_ERR_NOMEM; }
    uLongf got = (uLongf)raw_len;
    int z = uncompress(raw, &got, idat, (uLong)idat_len);
    free(idat);
    if (z != Z_OK || got != raw_len) { free(raw); free(rgba); return WPNG_ERR_CORRUPT; }

    /* undo the per-row filters in place (bpp = 3) */
    for (uint32_t y = 0; y < h; y++) {
        unsigned char* row = raw + (stride + 1) * y;
        unsigned char ft = row[0];
        unsigned char* cur = row + 1;
        const unsigned char* up = y ? row - stride : NULL; /* previous row's pixel bytes */
        for (size_t x = 0; x < stride; x++) {
            int a = x >= 3 ? cur[x - 3] : 0;
            int b = up ? up[x] : 0;
            int c = (up && x >= 3) ? up[x - 3] : 0;
            int v = cur[x];
            switch (ft) {
                case 0: break;
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) >> 1; break;
                case 4: v += paeth(a, b, c); break;
                default: free(raw); free(rgba); return WPNG_ERR_CORRUPT;
            }
            cur[x] = (unsigned char)v;
        }
        uint8_t* dst = rgba + (size_t)y * w * 4;
        for (uint32_t x = 0; x < w; x++) {
            dst[4 * x + 0] = cur[3 * x + 0];
            dst[4 * x + 1] = cur[3 * x + 1];
            dst[4 * x + 2] = cur[3 * x + 2];
            dst[4 * x + 3] = 255;
        }
    }
    free(raw);
    *width = w; *height = h; *rgba_out = rgba;
    return WPNG_OK;
}

static void put_be32(unsigned char* p, uint32_t v) {
    p[0] = (unsigned char)(v >> 24); p[1] = (unsigned char)(v >> 16); p[2] = (unsigned char)(v >> 8); p[3] = (unsigned char)v;
}

static int write_chunk(FILE* fp, const char* type, const unsigned char* data, uint32_t len) {
    unsigned char hd[8];
    put_be32(hd, len);
    memcpy(hd + 4, type, 4);
    uLong crc = crc32(0L, hd + 4, 4);
    if (len) crc = crc32(crc, data, len);
    unsigned char tail[4];
    put_be32(tail, (uint32_t)crc);
    if (fwrite(hd, 1, 8, fp) != 8) return 0;
    if (len && fwrite(data, 1, len, fp) != len) return 0;
    return fwrite(tail, 1, 4, fp) == 4;
}

static void fill_rows(unsigned char* rows, size_t stride, const uint32_t* xrgb, uint32_t width, uint32_t y0, uint32_t n) {
    for (uint32_t r = 0; r < n; r++) {
        unsigned char* row = rows + stride * r;
        const uint32_t* src = xrgb + (size_t)(y0 + r) * width;
        row[0] = 0; /* filter: none */
        for (uint32_t x = 0; x < width; x++) {
            uint32_t v = src[x];
            row[1 + 3 * x] = (unsigned char)(v >> 16);
            row[2 + 3 * x] = (unsigned char)(v >> 8);
            row[3 + 3 * x] = (unsigned char)v;
        }
    }
}

/* ---- big frames: the bands are deflated in PARALLEL ---------------------------------------------------------------
 * A PNG's IDAT data is ONE zlib stream, but a deflate stream may be cut at byte boundaries: every band of WPNG_BAND rows
 * is compressed on its own as RAW deflate ended by Z_SYNC_FLUSH (an empty stored block, BFINAL = 0; the last band ends
 * with Z_FINISH), and the pieces are concatenated behind a two-byte zlib header with the Adler-32 of the whole image
 * (adler32_combine of the bands') at the end -- the pigz construction.  A band cannot refer back into the previous one,
 * which costs a fraction of a per cent of size.  Measured for config C5's 8192 x 8192 frame (192 MB of RGB, an
 * 11 MB file) on the GPU box's host cores: 0.037 s (bench.py --config c5, `png.write_s`). */
#include <pthread.h>
#include <unistd.h>
#define WPNG_BAND 256u
typedef struct {
    const uint32_t* xrgb; uint32_t width, height, nbands; int level;
    unsigned char** out; size_t* out_len; uLong* adler; volatile int failed;
    uint32_t next; pthread_mutex_t lock;
} wpng_job;

static void* wpng_worker(void* arg) {
    wpng_job* J = (wpng_job*)arg;
    const size_t stride = (size_t)J->width * 3 + 1;
    unsigned char* rows = (unsigned char*)malloc(stride * WPNG_BAND);
    if (!rows) { J->failed = 1; return NULL; }
    for (;;) {
        pthread_mutex_lock(&J->lock);
        const uint32_t b = J->next++;
        pthread_mutex_unlock(&J->lock);
        if (b >= J->nbands || J->failed) break;
        const uint32_t y0 = b * WPNG_BAND, n = J->height - y0 < WPNG_BAND ? J->height - y0 : WPNG_BAND;
        fill_rows(rows, stride, J->xrgb, J->width, y0, n);
        const size_t in_len = stride * n;
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (deflateInit2(&zs, J->level < 0 ? Z_DEFAULT_COMPRESSION : J->level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { J->failed = 1; break; }
        const size_t cap = deflateBound(&zs, (uLong)in_len) + 64;
        unsigned char* z = (unsigned char*)malloc(cap);
        if (!z) { deflateEnd(&zs); J->failed = 1; break; }
        zs.next_in = rows; zs.avail_in = (uInt)in_len; zs.next_out = z; zs.avail_out = (uInt)cap;
        const int last = b + 1 == J->nbands;
        const int zr = deflate(&zs, last ? Z_FINISH : Z_SYNC_FLUSH);
        if ((last && zr != Z_STREAM_END) || (!last && (zr != Z_OK || zs.avail_in != 0 || zs.avail_out == 0))) { free(z); deflateEnd(&zs); J->failed = 1; break; }
        J->out[b] = z; J->out_len[b] = cap - zs.avail_out;
        J->adler[b] = adler32(adler32(0L, Z_NULL, 0), rows, (uInt)in_len);
        deflateEnd(&zs);
    }
    free(rows);
    return NULL;
}

static int wpng_write_parallel(FILE* fp, const uint32_t* xrgb, uint32_t width, uint32_t height, int level) {
    wpng_job J;
    memset(&J, 0, sizeof J);
    J.xrgb = xrgb; J.width = width; J.height = height; J.level = level;
    J.nbands = (height + WPNG_BAND - 1) / WPNG_BAND;
    J.out = (unsigned char**)calloc(J.nbands, sizeof *J.out);
    J.out_len = (size_t*)calloc(J.nbands, sizeof *J.out_len);
    J.adler = (uLong*)calloc(J.nbands, sizeof *J.adler);
    if (!J.out || !J.out_len || !J.adler) { free(J.out); free(J.out_len); free(J.adler); return WPNG_ERR_NOMEM; }
    pthread_mutex_init(&J.lock, NULL);
    long nc = sysconf(_SC_NPROCESSORS_ONLN);
    int nt = (int)(nc < 1 ? 1 : (nc > 32 ? 32 : nc));
    if ((uint32_t)nt > J.nbands) nt = (int)J.nbands;
    pthread_t th[32];
    int started = 0;
    for (int t = 0; t < nt; t++) if (pthread_create(&th[started], NULL, wpng_worker, &J) == 0) started++;
    if (started == 0) wpng_worker(&J);                 /* no threads to be had: this one does it all */
    for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
    pthread_mutex_destroy(&J.lock);
    int rc = J.failed ? WPNG_ERR_NOMEM : WPNG_OK;
    const size_t stride = (size_t)width * 3 + 1;
    uLong ad = adler32(0L, Z_NULL, 0);
    static const unsigned char zhdr[2] = {0x78, 0x01};  /* deflate, 32 KiB window, no dictionary, fastest-level hint; 0x7801 % 31 == 0 */
    if (rc == WPNG_OK && !write_chunk(fp, "IDAT", zhdr, 2)) rc = WPNG_ERR_IO;
    for (uint32_t b = 0; b < J.nbands; b++) {
        if (rc == WPNG_OK && !J.out[b]) rc = WPNG_ERR_NOMEM;
        if (rc == WPNG_OK) {
            const uint32_t n = height - b * WPNG_BAND < WPNG_BAND ? height - b * WPNG_BAND : WPNG_BAND;
            ad = b ? adler32_combine(ad, J.adler[b], (z_off_t)(stride * n)) : J.adler[b];
            for (size_t off = 0; off < J.out_len[b] && rc == WPNG_OK; off += 0x7FFF0000u) {   /* a chunk length is < 2^31 */
                const size_t len = J.out_len[b] - off < 0x7FFF0000u ? J.out_len[b] - off : 0x7FFF0000u;
                if (!write_chunk(fp, "IDAT", J.out[b] + off, (uint32_t)len)) rc = WPNG_ERR_IO;
            }
        }
        free(J.out[b]);
    }
    unsigned char tail[4];
    put_be32(tail, (uint32_t)ad);
    if (rc == WPNG_OK && !write_chunk(fp, "IDAT", tail, 4)) rc = WPNG_ERR_IO;
    free(J.out); free(J.out_len); free(J.adler);
    return rc;
}

int wpng_write_xrgb(const char* path, const uint32_t* xrgb, uint32_t width, uint32_t height, int level) {
    FILE* fp = fopen(path, "wb");
    if (!fp) return WPNG_ERR_OPEN;
    int rc = WPNG_OK;
    unsigned char ih[13];
    put_be32(ih, width); put_be32(ih + 4, height);
    ih[8] = 8; ih[9] = 2; ih[10] = 0; ih[11] = 0; ih[12] = 0;
    if (fwrite(k_sig, 1, 8, fp) != 8 || !write_chunk(fp, "IHDR", ih, 13)) rc = WPNG_ERR_IO;
    if (rc == WPNG_OK && (uint64_t)width * height >= (1u << 22) && height >= 4 * WPNG_BAND) {
        rc = wpng_write_parallel(fp, xrgb, width, height, level);     /* >= 4 Mpixel: bands deflated by all cores */
        if (rc == WPNG_OK && !write_chunk(fp, "IEND", NULL, 0)) rc = WPNG_ERR_IO;
        if (fclose(fp) != 0 && rc == WPNG_OK) rc = WPNG_ERR_IO;
        return rc;
    }
    const size_t stride = (size_t)width * 3 + 1;
    const uint32_t band = WPNG_BAND; /* rows deflated per IDAT chunk: bounded memory */
    unsigned char* rows = (unsigned char*)malloc(stride * band);
    const size_t zcap = compressBound((uLong)(stride * band)) + 64;
    unsigned char* zbuf = (unsigned char*)malloc(zcap);
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (!rows || !zbuf || deflateInit(&zs, level < 0 ? Z_DEFAULT_COMPRESSION : level) != Z_OK) {
        free(rows); free(zbuf); fclose(fp); return WPNG_ERR_NOMEM;
    }
    for (uint32_t y0 = 0; y0 < height && rc == WPNG_OK; y0 += band) {
        uint32_t n = height - y0 < band ? height - y0 : band;
        fill_rows(rows, stride, xrgb, width, y0, n);
        zs.next_in = rows;
        zs.avail_in = (uInt)(stride * n);
        const int flush = (y0 + n >= height) ? Z_FINISH : Z_NO_FLUSH;
        do { /* canonical zlib loop: run deflate until it stops filling the output buffer */
            zs.next_out = zbuf;
            zs.avail_out = (uInt)zcap;
            if (deflate(&zs, flush) == Z_STREAM_ERROR) { rc = WPNG_ERR_IO; break; }
            uint32_t have = (uint32_t)(zcap - zs.avail_out);
            if (have && !write_chunk(fp, "IDAT", zbuf, have)) { rc = WPNG_ERR_IO; break; }
        } while (zs.avail_out == 0);
    }
    deflateEnd(&zs);
    if (rc == WPNG_OK && !write_chunk(fp, "IEND", NULL, 0)) rc = WPNG_ERR_IO;
    free(rows); free(zbuf);
    if (fclose(fp) != 0 && rc == WPNG_OK) rc = WPNG_ERR_IO;
    return rc;
}

int wpng_write_rgba_as_rgb(const char* path, const uint8_t* rgba, uint32_t width, uint32_t height, int level) {
    uint32_t* tmp = (uint32_t*)malloc((size_t)width * height * 4);
    if (!tmp) return WPNG_ERR_NOMEM;
    for (size_t i = 0; i < (size_t)width * height; i++)
        tmp[i] = ((uint32_t)rgba[4 * i] << 16) | ((uint32_t)rgba[4 * i + 1] << 8) | rgba[4 * i + 2];
    int rc = wpng_write_xrgb(path, tmp, width, height, level);
    free(tmp);
    return rc;
}
