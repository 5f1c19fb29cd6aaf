/* scene_prep.h -- see scene_prep.c */
#ifndef WHITTED_SCENE_PREP_H
#define WHITTED_SCENE_PREP_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
/* number of float4 in the prepared geometry stream */
size_t wprep_geom_f4(uint32_t ns, uint32_t np, uint32_t nl);
/* geom: 4*wprep_geom_f4 floats; ptex: 8*np floats */
void wprep_build(const uint8_t* spheres, uint32_t ns, const uint8_t* planes, uint32_t np,
                 const uint8_t* lights, uint32_t nl, float* geom, float* ptex);

/* light / plane side table (see scene_prep.c): wprep_lpt_f4 float4 entries, appended behind the geometry stream */
size_t wprep_lpt_f4(uint32_t np, uint32_t nl);
void wprep_build_lpt(const uint8_t* planes, uint32_t np, const uint8_t* lights, uint32_t nl, float* out);

/* Uniform grid over the spheres (an acceleration structure for big scenes; it changes how many tests a ray
 * makes, never their results).  wprep_grid_plan fills `g` (bounds, resolution) and returns the number of
 * (cell, sphere) pairs; wprep_grid_fill then writes start[ncells+1], items[pairs], box[2*ns]. */
typedef struct {
    float gmin[3], inv[3], cell[3];
    int32_t res[3];
    uint32_t ncells;
    float reg_pad;      /* every sphere is listed in all cells its box, widened by this much, touches (and the bounds reach that far) */
} wprep_grid;
/* reg_pad > 0 (the largest light radius): the two soft-shadow samples of a light can then share ONE walk along the ray to the light's
 * centre -- a sphere either sample meets lies within reg_pad of that ray, so a cell the walk visits lists it (whitted_trace.inc
 * wt_grid_shadow_pair) */
size_t wprep_grid_plan(const uint8_t* spheres, uint32_t ns, float density, float reg_pad, wprep_grid* g);
void wprep_grid_fill(const uint8_t* spheres, uint32_t ns, const wprep_grid* g, uint32_t* start, uint32_t* items,
                     uint32_t* box);
#ifdef __cplusplus
}
#endif
#endif
