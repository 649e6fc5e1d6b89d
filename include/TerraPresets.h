/*
 * TerraPresets.h -- built-in BSDF presets (drop-in boundary, part 3 of 3).
 *
 * Same names and attribute-slot numbers as reference include/TerraPresets.h:11-33.
 * terra_bsdf_*_init() stores the library's own sample/pdf/eval entry points in
 * the TerraBSDF; terra_scene_commit() recognises those pointers and selects the
 * matching device BSDF ("Function-pointer plug-ins").
 *
 * The sample/pdf/eval entry points themselves are exported (as in the
 * reference, src/TerraPresets.c:34,47,52,84,108,125) as *markers*: calling
 * them on the host is not supported -- they log an error and return zero --
 * because this library has no CPU shading path.
 */
#ifndef TERRA_AMD_TERRA_PRESETS_H
#define TERRA_AMD_TERRA_PRESETS_H

#include "Terra.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Lambertian, cosine-weighted sampling. attributes[0] = albedo. */
#define TERRA_DIFFUSE_ALBEDO 0
#define TERRA_DIFFUSE_END    1
void terra_bsdf_diffuse_init ( TerraBSDF* bsdf );

typedef struct {
    TerraFloat3 albedo;
} TerraMaterialDiffuse;

/* Modified Phong. Slot 3 is scratch written by sample() and read by pdf(). */
#define TERRA_PHONG_SPECULAR_COLOR     0
#define TERRA_PHONG_ALBEDO             1
#define TERRA_PHONG_SPECULAR_INTENSITY 2
#define TERRA_PHONG_SAMPLE_PICK        3
#define TERRA_PHONG_END                4
void terra_bsdf_phong_init ( TerraBSDF* bsdf );

typedef struct {
    TerraFloat3 albedo;
    TerraFloat3 specular_color;
    float       specular_intensity;
    bool        sample_diffuse;
} TerraMaterialPhong;

/* ---- presets the reference does not have in runnable form ------------------------------
 * BASELINE.json config 4 asks for a GGX metal and a dielectric glass. The reference only
 * contains dead code for them (src/TerraPresets.c:298-465, inside #if 0, written against a
 * removed API), so there is NO reference behaviour to match: these two presets are DEFINED
 * here, in the live sample/pdf/eval form, from that code's building blocks (GGX D :316-320,
 * Smith G1 :307-314, half-vector sampling :333-343, Snell/TIR/Schlick :399-449). Parity for
 * them is device vs this repo's oracle only ("parity unpinned" vs the reference).
 *
 * GGX conductor:  attributes[0] = F0 (specular colour), attributes[1].x = roughness alpha.
 * Glass:          attributes[0] = tint, material.ior = index of refraction; attributes[2]
 *                 and attributes[3].x are scratch written by sample() (the chosen direction and
 *                 its probability), like Phong's slot 3. */
#define TERRA_GGX_F0        0
#define TERRA_GGX_ROUGHNESS 1
#define TERRA_GGX_END       2
void terra_bsdf_ggx_init ( TerraBSDF* bsdf );

#define TERRA_GLASS_TINT        0
#define TERRA_GLASS_UNUSED      1
#define TERRA_GLASS_SAMPLE_DIR  2
#define TERRA_GLASS_SAMPLE_PROB 3
#define TERRA_GLASS_END         4
void terra_bsdf_glass_init ( TerraBSDF* bsdf );

#define TERRA_DISNEY_BASE_COLOR 0
#define TERRA_DISNEY_SHEEN      1

typedef struct {
    TerraFloat3 base_color;
    float specular, specular_tint;
    float sheen, sheen_tint;
    float clearcoat, clearcoat_gloss;
    float metalness, roughness;
    float anisotropic, subsurface;
} TerraMaterialDisney;

/* Profiler target ids kept for source compatibility; profiling is done with
   rocprofv3 and terra_amd_get_stats() instead (reference include/TerraPresets.h:53-60). */
#define TERRA_PROFILE_SESSION_DEFAULT 0
#define TERRA_PROFILE_TARGET_RENDER   0
#define TERRA_PROFILE_TARGET_TRACE    1
#define TERRA_PROFILE_TARGET_RAY      2
#define TERRA_PROFILE_TARGET_RAY_TRIANGLE_INTERSECTION 3
#define TERRA_PROFILE_TARGET_COUNT    4

#ifdef __cplusplus
}
#endif
#endif /* TERRA_AMD_TERRA_PRESETS_H */
