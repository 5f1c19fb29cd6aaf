/*
 * whitted_params.h -- launch parameters shared by the host shim and the HIP kernels.
 *
 * "Prepared geometry": one float4 stream per scene, built once on the host from the raw
 * wire structs (scene_prep.c) so the intersection loops read 16 B per sphere instead of
 * the 96-B rsphere the reference copies per test (reference primitives.cl:295,338,409):
 *
 *   geom[0 .. ns)                 sphere i : { cx, cy, cz, +-r*r }   sign bit set = transparent
 *   geom[ns + 2p], geom[ns+2p+1]  plane p  : { nx, ny, nz, has_texture ? 1 : 0 }, { px, py, pz, 0 }
 *   geom[ns + 2np + 2l], [..+1]   light l  : { ox, oy, oz, r*r }, { lr, lg, lb, radius }
 *                                            with l* = (rgb * intensity) * (1/pi)
 *   geom[ns + 2np + 2nl + c*np + p]  (only if lpt != 0) light chunk c, plane p: { mu0, mu1, mu2, 1/|n| } -- the signed
 *                                            distance of the three lights of the chunk from the plane, less radius and
 *                                            slack (scene_prep.c): lets a wave skip plane tests no shadow ray can fail
 *   ptex[2p], ptex[2p+1]          plane p  : { b0x, b0y, b0z, texture_scale }, { b1x, b1y, b1z, bits(texture_id) }
 *                                            (tangent basis of reference primitives.cl:226-236)
 * Materials stay in the raw arrays and are gathered for the winning primitive only.
 */
#ifndef WHITTED_PARAMS_H
#define WHITTED_PARAMS_H
#include <stdint.h>

#define CLW_MAX_DEPTH 32 /* deepest supported trace depth (hip_wrap_ext.h) */
#define CLW_NUM_COUNTERS 32
#define CLW_STAMP_SHARDS 1024 /* diagnostic stamp build: 16-word shards behind the counter block, summed into words 16.. on read */ /* words of the device counter block (hip_wrap_ext.h: clw_ext_read_counters_ex); 16.. = phase stamps of the diagnostic build */

typedef struct {
    /* camera: the eight by-value raygen arguments (reference raygen.cl:5-8) */
    float corner[3], origin[3], up[3], right[3];
    float w_factor, h_factor;
    uint32_t width, height;
    /* work range: work-item i <-> global id id_offset + i, i in [0, n_items) */
    uint64_t id_offset;
    uint32_t n_items;
    uint32_t tiled;        /* 1: range is whole rows -> 8x8 pixel tile per wavefront */
    uint32_t rows;         /* tiled: number of rows in the range                      */
    uint32_t row_offset;   /* tiled: global row of the range's first row (= id_offset / width) */
    /* interleaved 8-row bands (multi-GPU load balance): local row y is global row
     * ((y/8)*band_stride + band_phase)*8 + y%8; band_stride <= 1 = contiguous range        */
    uint32_t band_stride, band_phase;
    /* cost-sorted tile dispatch (tiled mode; both nullable): tile_order[b] = tile served by workgroup b
     * (packed: tile column | tile row << 12 | log2(parts) << 24 | part << 27; 0xFFFFFFFF = none), heaviest tiles of the previous frame first, one list per
     * XCD interleaved as order[8*j + k]; tile_cost[tile] receives this frame's cost of every tile.  */
    const uint32_t* tile_order;
    uint32_t* tile_cost;
    /* the tree-parallel tail of deep launches (whitted_tpt.inc): once at most tpt_max lanes of a tile's wave are alive and they hold at
     * least tpt_min pending paths, the rest of the tile is traced by the whole wave as one pool of segments whose nodes live in a slot
     * of tpt_pool: 8 x tpt_slots slots (tpt_slots per XCD) of tpt_slice_words words, room for tpt_cap nodes each; tpt_flags[8 * tpt_slots]:
     * 1 = slot taken (tpt_cap 0 = no tail: the per-lane loop runs to the end) */
    uint32_t* tpt_pool;
    uint32_t* tpt_flags;
    const uint32_t* tpt_jump;   /* [7][32]: the GF(2) matrices M^(2^i), M = the xorshift steps of one shaded hit (4 x lights), as columns */
    uint32_t tpt_slice_words, tpt_cap, tpt_slots, tpt_max, tpt_min;
    uint32_t tpt_clock;    /* DIAGNOSTIC: 1 = the tail books its phases' durations into counter words 10-25 (`counters` must be set) */
    uint32_t cost_sum;     /* 1: a tile's cost is the SUM over its lanes (and over the wavefronts that share the tile), added atomically to a zeroed tile_cost; 0: the maximum over its lanes, stored */
    int32_t depth;         /* reference MAX_DEPTH                                     */
    /* scene */
    const float* geom;     /* float4 stream, layout above                             */
    const float* ptex;     /* float4 x 2 per plane                                    */
    const uint8_t* spheres_raw; /* 96-B rsphere array (materials)                     */
    const uint8_t* planes_raw;  /* 96-B rplane array                                  */
    uint32_t ns, np, nl;
    float through;         /* factor a transparent sphere applies to a shadow ray: 0.8f = reference primitives.cl:7 (clw_ext_set_shadow_through) */
    uint32_t geom_f4;      /* number of float4 in geom                                */
    uint32_t unit_dirs;    /* 1: ray directions are unit (this library's own rays) and the scene is small: the fast build takes a = d.d = 1 */
    uint32_t lpt;          /* 1: the light / plane side table follows the lights in geom */
    uint32_t vis;          /* strict build, small scenes: 1: lights are sorted into visibility classes (wt_light_vis) and only the samples of
                              undecided ones are traced; 2 (counting build): classify AND trace, count disagreements in counter word 28 */
    uint32_t diag;         /* DIAGNOSTIC builds (-DWT_TIMELINE=1) only: 1 + s = tile_cost receives (start << 16 | end) in ticks of 10 ns << s, not costs */
    /* uniform grid over the spheres (big scenes only; see scene_prep.c wprep_grid_*): cell c holds
     * grid_items[grid_start[c] .. grid_start[c+1]) = sphere indices in ascending order; grid_box[2i], [2i+1] =
     * sphere i's inclusive cell box, 10 bits per axis: lo = x0 | y0<<10 | z0<<20, hi likewise               */
    const uint32_t* grid_start;
    const uint32_t* grid_items;
    const uint32_t* grid_box;
    const float* grid_geom;   /* float4 per cell-list entry: geom[grid_items[k]], so a test costs one load, not two dependent ones */
    float grid_min[3], grid_inv[3], grid_cell[3];
    uint32_t grid_pair;      /* 1: spheres are listed a light radius beyond their boxes: the two samples of a light share one walk (wt_grid_shadow_pair) */
    int32_t grid_res[3];
    /* images: RGBA8 layer stacks */
    const uint32_t* tex; int32_t tex_w, tex_h, tex_layers;
    const uint32_t* sky; int32_t sky_w, sky_h;
    /* I/O */
    const float* rays;     /* unfused path: 64-B rray records, else NULL              */
    uint32_t* out;         /* packed 0x00RRGGBB per work-item                         */
    float* out_rgb;        /* optional float radiance, 3 per work-item                */
    unsigned long long* counters; /* counting build only: CLW_NUM_COUNTERS words          */
} whitted_params;

typedef struct {
    float corner[3], origin[3], up[3], right[3];
    float w_factor, h_factor;
    uint32_t width, height;
    uint64_t id_offset;
    uint32_t n_items;
    uint32_t band_stride, band_phase;
    float* rays;           /* 16 floats per work-item                                 */
} raygen_params;

#endif
