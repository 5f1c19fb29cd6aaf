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
#ifdef __cplusplus
}
#endif
#endif
