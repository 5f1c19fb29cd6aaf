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

int wpng_write_xrgb(const char* path, const uint32_t* xrgb, uint32_t width, uint32_t height, int level) {
    FILE* fp = fopen(path, "wb");
    if (!fp) return WPNG_ERR_OPEN;
    int rc = WPNG_OK;
    const size_t stride = (size_t)width * 3 + 1;
    const uint32_t band = 256; /* rows deflated per IDAT chunk: bounded memory for 8192^2 frames */
    unsigned char* rows = (unsigned char*)malloc(stride * band);
    const size_t zcap = compressBound((uLong)(stride * band)) + 64;
    unsigned char* zbuf = (unsigned char*)malloc(zcap);
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (!rows || !zbuf || deflateInit(&zs, level < 0 ? Z_DEFAULT_COMPRESSION : level) != Z_OK) {
        free(rows); free(zbuf); fclose(fp); return WPNG_ERR_NOMEM;
    }
    unsigned char ih[13];
    put_be32(ih, width); put_be32(ih + 4, height);
    ih[8] = 8; ih[9] = 2; ih[10] = 0; ih[11] = 0; ih[12] = 0;
    if (fwrite(k_sig, 1, 8, fp) != 8 || !write_chunk(fp, "IHDR", ih, 13)) rc = WPNG_ERR_IO;

    for (uint32_t y0 = 0; y0 < height && rc == WPNG_OK; y0 += band) {
        uint32_t n = height - y0 < band ? height - y0 : band;
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
