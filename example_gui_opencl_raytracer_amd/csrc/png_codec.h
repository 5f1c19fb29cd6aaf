/* png_codec.h -- see png_codec.c */
#ifndef WPNG_CODEC_H
#define WPNG_CODEC_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
enum {
    WPNG_OK = 0,
    WPNG_ERR_OPEN = 1,        /* cannot open the file                                  */
    WPNG_ERR_NOT_PNG = 2,     /* bad signature                                         */
    WPNG_ERR_FORMAT = 3,      /* not 8-bit RGB (the reference rejects these too)        */
    WPNG_ERR_UNSUPPORTED = 4, /* interlaced / unknown compression or filter method      */
    WPNG_ERR_CORRUPT = 5,
    WPNG_ERR_NOMEM = 6,
    WPNG_ERR_IO = 7
};
/* Decodes an 8-bit RGB PNG into malloc'ed RGBA8 (A = 255), row-major.  Caller frees. */
int wpng_read_rgba(const char* path, uint32_t* width, uint32_t* height, uint8_t** rgba_out);
/* Writes a 0x00RRGGBB framebuffer as an 8-bit RGB PNG.  level: zlib level, <0 = default. */
int wpng_write_xrgb(const char* path, const uint32_t* xrgb, uint32_t width, uint32_t height, int level);
/* Writes the RGB channels of an RGBA8 image (used to emit procedural textures as files). */
int wpng_write_rgba_as_rgb(const char* path, const uint8_t* rgba, uint32_t width, uint32_t height, int level);
#ifdef __cplusplus
}
#endif
#endif
