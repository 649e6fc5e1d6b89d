/*
 * cornell_demo.c -- a C client written against Terra.h / TerraPresets.h only.
 *
 * The same source links against the reference's objects or against
 * libterra_amd.so: it builds the 32-triangle Cornell
 * box of SURVEY.md section 8d through terra_scene_add_object, commits, renders
 * the frame in 128-pixel tiles (the reference client's default tile size,
 * satellite/include/Config.hpp:25) and writes a binary PPM.
 *
 *   cc -Iinclude examples/cornell_demo.c -Lterra_amd -lterra_amd -Wl,-rpath,$PWD/terra_amd -lm -o cornell_demo
 *   ./cornell_demo out.ppm 512 512 64
 */
#include <stdio.h>
#include <string.h>
#include "Terra.h"
#include "TerraPresets.h"

#ifdef TERRA_AMD_H_AVAILABLE
#include "terra_amd.h"
#endif
const char* terra_amd_last_error ( void ) __attribute__ ( ( weak ) );   /* absent when linked against the reference */

static void quad ( TerraObject* o, size_t* k, const float p[4][3], const float n[3] ) {
    const int idx[2][3] = { { 0, 1, 2 }, { 0, 2, 3 } };
    for ( int t = 0; t < 2; ++t, ++*k ) {
        TerraTriangle* tri = &o->triangles[*k];
        TerraTriangleProperties* pr = &o->properties[*k];
        tri->a = terra_f3_setv ( p[idx[t][0]] ); tri->b = terra_f3_setv ( p[idx[t][1]] ); tri->c = terra_f3_setv ( p[idx[t][2]] );
        pr->normal_a = pr->normal_b = pr->normal_c = terra_f3_setv ( n );
        pr->texcoord_a = pr->texcoord_b = pr->texcoord_c = terra_f2_set ( 0.f, 0.f );
    }
}

static TerraObject* diffuse_object ( HTerraScene s, size_t tris, float r, float g, float b, float emit ) {
    TerraObject* o = terra_scene_add_object ( s, tris );
    TerraFloat3 albedo = terra_f3_set ( r, g, b ), e = terra_f3_set1 ( emit );
    terra_bsdf_diffuse_init ( &o->material.bsdf );
    terra_attribute_init_constant ( &o->material.attributes[TERRA_DIFFUSE_ALBEDO], &albedo );
    terra_attribute_init_constant ( &o->material.emissive, &e );
    o->material.attributes_count = TERRA_DIFFUSE_END;
    o->material.ior = 1.5f;
    return o;
}

static void box ( TerraObject* o, float x0, float x1, float y1, float z0, float z1 ) {
    size_t k = 0;
    const float top[4][3] = { { x0, y1, z0 }, { x1, y1, z0 }, { x1, y1, z1 }, { x0, y1, z1 } }, nt[3] = { 0, 1, 0 };
    const float fr[4][3] = { { x0, 0, z0 }, { x1, 0, z0 }, { x1, y1, z0 }, { x0, y1, z0 } }, nf[3] = { 0, 0, -1 };
    const float bk[4][3] = { { x0, 0, z1 }, { x1, 0, z1 }, { x1, y1, z1 }, { x0, y1, z1 } }, nb[3] = { 0, 0, 1 };
    const float lf[4][3] = { { x0, 0, z0 }, { x0, 0, z1 }, { x0, y1, z1 }, { x0, y1, z0 } }, nl[3] = { -1, 0, 0 };
    const float rt[4][3] = { { x1, 0, z0 }, { x1, 0, z1 }, { x1, y1, z1 }, { x1, y1, z0 } }, nr[3] = { 1, 0, 0 };
    quad ( o, &k, top, nt ); quad ( o, &k, fr, nf ); quad ( o, &k, bk, nb ); quad ( o, &k, lf, nl ); quad ( o, &k, rt, nr );
}

int main ( int argc, char** argv ) {
    const char* out = argc > 1 ? argv[1] : "cornell.ppm";
    size_t W = argc > 2 ? ( size_t ) atoi ( argv[2] ) : 256, H = argc > 3 ? ( size_t ) atoi ( argv[3] ) : 256, spp = argc > 4 ? ( size_t ) atoi ( argv[4] ) : 16;
    HTerraScene scene = terra_scene_create();
    size_t k;
    TerraObject* white = diffuse_object ( scene, 6, 0.73f, 0.73f, 0.73f, 0.f ); k = 0;
    { const float fl[4][3] = { { -1, 0, -1 }, { 1, 0, -1 }, { 1, 0, 1 }, { -1, 0, 1 } }, n0[3] = { 0, 1, 0 };
      const float ce[4][3] = { { -1, 2, -1 }, { 1, 2, -1 }, { 1, 2, 1 }, { -1, 2, 1 } }, n1[3] = { 0, -1, 0 };
      const float bw[4][3] = { { -1, 0, 1 }, { 1, 0, 1 }, { 1, 2, 1 }, { -1, 2, 1 } }, n2[3] = { 0, 0, -1 };
      quad ( white, &k, fl, n0 ); quad ( white, &k, ce, n1 ); quad ( white, &k, bw, n2 ); }
    TerraObject* red = diffuse_object ( scene, 2, 0.65f, 0.05f, 0.05f, 0.f ); k = 0;
    { const float p[4][3] = { { -1, 0, -1 }, { -1, 0, 1 }, { -1, 2, 1 }, { -1, 2, -1 } }, n[3] = { 1, 0, 0 }; quad ( red, &k, p, n ); }
    TerraObject* green = diffuse_object ( scene, 2, 0.12f, 0.45f, 0.15f, 0.f ); k = 0;
    { const float p[4][3] = { { 1, 0, -1 }, { 1, 0, 1 }, { 1, 2, 1 }, { 1, 2, -1 } }, n[3] = { -1, 0, 0 }; quad ( green, &k, p, n ); }
    TerraObject* light = diffuse_object ( scene, 2, 0.78f, 0.78f, 0.78f, 15.f ); k = 0;
    { const float p[4][3] = { { -0.25f, 1.99f, -0.25f }, { 0.25f, 1.99f, -0.25f }, { 0.25f, 1.99f, 0.25f }, { -0.25f, 1.99f, 0.25f } }, n[3] = { 0, -1, 0 }; quad ( light, &k, p, n ); }
    box ( diffuse_object ( scene, 10, 0.73f, 0.73f, 0.73f, 0.f ), 0.15f, 0.75f, 0.6f, -0.65f, -0.05f );
    box ( diffuse_object ( scene, 10, 0.73f, 0.73f, 0.73f, 0.f ), -0.75f, -0.15f, 1.2f, 0.05f, 0.65f );

    TerraSceneOptions* o = terra_scene_get_options ( scene );
    TerraFloat3 black = terra_f3_zero;
    terra_attribute_init_constant ( &o->environment_map, &black );
    o->tonemapping_operator = kTerraTonemappingOperatorReinhard; o->accelerator = kTerraAcceleratorBVH;
    o->sampling_method = kTerraSamplingMethodRandom; o->integrator = kTerraIntegratorDirect;
    o->subpixel_jitter = 0.5f; o->samples_per_pixel = spp; o->bounces = 8; o->strata = 4; o->manual_exposure = 1.f; o->gamma = 2.2f;
    terra_scene_commit ( scene );

    TerraCamera cam;
    cam.position = terra_f3_set ( 0.f, 1.f, -3.4f ); cam.direction = terra_f3_set ( 0.f, 0.f, 1.f ); cam.up = terra_f3_set ( 0.f, 1.f, 0.f ); cam.fov = 45.f;
    TerraFramebuffer fb;
    if ( !terra_framebuffer_create ( &fb, W, H ) ) { fprintf ( stderr, "framebuffer\n" ); return 1; }
    for ( size_t y = 0; y < H; y += 128 ) for ( size_t x = 0; x < W; x += 128 )
        terra_render ( &cam, scene, &fb, x, y, W - x < 128 ? W - x : 128, H - y < 128 ? H - y : 128 );
    if ( terra_amd_last_error && *terra_amd_last_error() ) { fprintf ( stderr, "terra_amd: %s\n", terra_amd_last_error() ); return 2; }

    FILE* f = fopen ( out, "wb" );
    if ( !f ) return 3;
    fprintf ( f, "P6\n%zu %zu\n255\n", W, H );
    for ( size_t i = 0; i < W * H; ++i ) {
        float c[3] = { fb.pixels[i].x, fb.pixels[i].y, fb.pixels[i].z };
        for ( int j = 0; j < 3; ++j ) { float v = c[j] < 0 ? 0 : ( c[j] > 1 ? 1 : c[j] ); fputc ( ( int ) ( v * 255.f ), f ); }     /* clamp x 255, as satellite/src/Visualization.cpp:286-357 */
    }
    fclose ( f );
    printf ( "wrote %s (%zux%zu, %zu spp)\n", out, W, H, spp );
    terra_framebuffer_destroy ( &fb );
    terra_scene_destroy ( scene );
    return 0;
}
