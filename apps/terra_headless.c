/*
 * terra_headless -- OBJ/MTL in, PNG/PPM/PFM/HDR out, through the Terra.h API only.
 *
 * the headless counterpart of the reference's GUI client for this
 * path -- mesh -> TerraObject fill and material mapping (satellite/src/Scene.cpp:133-245:
 * one object per mesh/material, specular materials -> Phong preset, everything else ->
 * diffuse; right-handed OBJ -> left-handed Terra by flipping z and the winding,
 * Scene.cpp:90-93) and image export (satellite/src/Visualization.cpp:286-357: clamp x 255
 * for PNG, raw floats for HDR). It links against libterra_amd.so or, unchanged, against the
 * reference's objects (only Terra.h / TerraPresets.h symbols are used; terra_amd_* are weak).
 *
 * Import policy (--normals apollo, the default): the reference's client imports OBJ with its Apollo importer
 * (satellite/include/Apollo.h:964-1700) under the options of satellite/src/Scene.cpp:83-93 -- recompute_vertex_normals,
 * remove_vertex_duplicates, flip_z, flip_faces_winding_order. Apollo.h does not compile with this image's toolchains, so its
 * policy is RESTATED here and pinned by hand-derived fixtures (tests/test_headless_tool.py), not by running it:
 *   - one object per `g` / `o` group (a group is opened implicitly by the first `usemtl`, `s` or `f`); the group's material is
 *     the LAST `usemtl` inside it; the material's class is the word of Apollo's own `illum <word>` dialect (Apollo.h:877-897) and only
 *     `illum specular` selects Phong: diffuse, mirror, pbr, disney and materials without such a word (every standard MTL file, whose
 *     `illum 2` Apollo rejects) become diffuse, the last four with a warning, as satellite/src/Scene.cpp:193-230 does; Ke is the emissive;
 *   - z is negated at parse time and every triangle (a, b, c) becomes (c, b, a);
 *   - the file's `vn` are ignored. Face normal = normalize(cross(v1 - v0, v2 - v0)) of the flipped triangle. A group is smooth
 *     iff its last `s` token starts with '1'; in smooth groups corners with bit-equal positions share one vertex (the first one
 *     created anywhere in the file) and its first texcoord. A vertex normal is the normalised, unweighted sum of the face
 *     normals of the triangles the vertex was PARSED for: corners 0-2 of a polygon count for its first triangle only, corner
 *     j >= 3 for triangle j - 2 (Apollo records adjacency per parsed corner, Apollo.h:1373-1381); `s off` / `s 0` groups get
 *     the face normal per triangle; groups without any `s` take the smooth path without sharing, i.e. face normals again.
 * --normals file keeps the file's `vn` (area-weighted smooth normals per position where absent), one object per material, and reads
 * standard MTL files the usual way (Ks > 0 -> Phong when there is no Apollo `illum` word): this repo's own policy, not the reference's.
 *
 *   terra_headless scene.obj out.png [--width W] [--height H] [--spp N] [--bounces N]
 *       [--integrator simple|direct|mis|normals|depth] [--tonemap none|linear|reinhard|filmic|uncharted2]
 *       [--camera px py pz dx dy dz] [--fov deg] [--exposure e] [--gamma g] [--jitter j]
 *       [--no-flip-z] [--normals apollo|file] [--dump-scene file] [--no-render] [--fast-tree | --replica-tree] [--sample-split n] [--seed n] [--tile n] [--gpus n]
 */
#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "Terra.h"
#include "TerraPresets.h"

const char* terra_amd_last_error ( void ) __attribute__ ( ( weak ) );
int         terra_amd_set_tree_mode ( HTerraScene, int ) __attribute__ ( ( weak ) );
int         terra_amd_set_sample_split ( HTerraScene, int ) __attribute__ ( ( weak ) );
void        terra_amd_set_frame_seed ( HTerraScene, uint64_t ) __attribute__ ( ( weak ) );
int         terra_amd_init ( void ) __attribute__ ( ( weak ) );
int         terra_amd_set_devices ( const int*, int ) __attribute__ ( ( weak ) );
int         terra_amd_device_count ( void ) __attribute__ ( ( weak ) );
int         terra_amd_render_multi ( const TerraCamera*, HTerraScene, const TerraFramebuffer*, size_t, size_t, size_t, size_t, size_t ) __attribute__ ( ( weak ) );

/* ---- growable arrays ------------------------------------------------------------------------ */
#define VEC(T) struct { T* d; size_t n, cap; }
#define PUSH(v, x) do { if ( ( v ).n == ( v ).cap ) { ( v ).cap = ( v ).cap ? ( v ).cap * 2 : 256; ( v ).d = realloc ( ( v ).d, ( v ).cap * sizeof *( v ).d ); } ( v ).d[( v ).n++] = ( x ); } while ( 0 )

/* illum: Apollo's BSDF class of the material (satellite/include/Apollo.h:80-87, set ONLY by its own `illum <word>` dialect, :877-897):
   0 diffuse, 1 specular, 2 mirror, 3 pbr, 4 disney, 5 invalid = no `illum` line or a word Apollo does not know (the numeric models of
   standard MTL files included: "Unsupported material bsdf", the class stays invalid) */
enum { ILLUM_DIFFUSE = 0, ILLUM_SPECULAR, ILLUM_MIRROR, ILLUM_PBR, ILLUM_DISNEY, ILLUM_INVALID };
typedef struct { char name[128]; float kd[3], ks[3], ke[3], ns; int illum; } Mtl;
typedef struct { int v[3], t[3], n[3]; int mtl; } Face;
typedef struct { int v, t, n; } Corner;
typedef struct { size_t first, count; int group; } Poly;              /* corners [first, first + count) in file order */
typedef struct { int smooth, mtl; } Group;                            /* smooth: -1 no `s` seen, else 1 / 0 */

typedef struct {
    VEC ( TerraFloat3 ) pos, nrm;
    VEC ( TerraFloat2 ) uv;
    VEC ( Face ) faces;
    VEC ( Corner ) corners;
    VEC ( Poly ) polys;
    VEC ( Group ) groups;
    VEC ( Mtl ) mtls;
} Model;

static int find_mtl ( Model* m, const char* name ) {
    for ( size_t i = 0; i < m->mtls.n; ++i ) if ( strcmp ( m->mtls.d[i].name, name ) == 0 ) return ( int ) i;
    return -1;
}

static void load_mtl ( Model* m, const char* dir, const char* file ) {
    char path[2048];
    snprintf ( path, sizeof path, "%s%s", dir, file );
    FILE* f = fopen ( path, "r" );
    if ( !f ) { fprintf ( stderr, "terra_headless: cannot open material library %s\n", path ); return; }
    char line[1024]; Mtl* cur = NULL;
    while ( fgets ( line, sizeof line, f ) ) {
        char key[64]; int off = 0;
        if ( sscanf ( line, "%63s%n", key, &off ) != 1 || key[0] == '#' ) continue;
        const char* rest = line + off;
        if ( strcmp ( key, "newmtl" ) == 0 ) {
            Mtl x; memset ( &x, 0, sizeof x );
            sscanf ( rest, "%127s", x.name );
            x.kd[0] = x.kd[1] = x.kd[2] = 0.7f; x.ns = 1.f; x.illum = ILLUM_INVALID;
            PUSH ( m->mtls, x ); cur = &m->mtls.d[m->mtls.n - 1];
        } else if ( cur ) {
            if ( strcmp ( key, "Kd" ) == 0 ) sscanf ( rest, "%f %f %f", &cur->kd[0], &cur->kd[1], &cur->kd[2] );
            else if ( strcmp ( key, "Ks" ) == 0 ) sscanf ( rest, "%f %f %f", &cur->ks[0], &cur->ks[1], &cur->ks[2] );
            else if ( strcmp ( key, "Ke" ) == 0 ) sscanf ( rest, "%f %f %f", &cur->ke[0], &cur->ke[1], &cur->ke[2] );
            else if ( strcmp ( key, "Ns" ) == 0 ) sscanf ( rest, "%f", &cur->ns );
            else if ( strcmp ( key, "illum" ) == 0 ) {
                static const char* words[] = { "diffuse", "specular", "mirror", "pbr", "disney" };
                char v[64] = ""; sscanf ( rest, "%63s", v );
                for ( int k = 0; k < 5; ++k ) if ( !strcmp ( v, words[k] ) ) cur->illum = k;      /* any other word leaves the class as it was (Apollo.h:895-897) */
            }
        }
    }
    fclose ( f );
}

static int fix_index ( int i, size_t n ) { return i > 0 ? i - 1 : ( i < 0 ? ( int ) n + i : -1 ); }

static int load_obj ( Model* m, const char* path ) {
    FILE* f = fopen ( path, "r" );
    if ( !f ) { fprintf ( stderr, "terra_headless: cannot open %s\n", path ); return 0; }
    char dir[1024] = "";
    const char* slash = strrchr ( path, '/' );
    if ( slash ) { size_t k = ( size_t ) ( slash - path ) + 1; if ( k < sizeof dir ) { memcpy ( dir, path, k ); dir[k] = 0; } }
    char line[4096]; int cur_mtl = -1;
#define NEED_GROUP() do { if ( m->groups.n == 0 ) { Group g0 = { -1, -1 }; PUSH ( m->groups, g0 ); } } while ( 0 )
    while ( fgets ( line, sizeof line, f ) ) {
        char* p = line;
        while ( isspace ( ( unsigned char ) *p ) ) ++p;
        if ( p[0] == 'v' && p[1] == ' ' ) { TerraFloat3 v = { 0, 0, 0 }; sscanf ( p + 2, "%f %f %f", &v.x, &v.y, &v.z ); PUSH ( m->pos, v ); }
        else if ( p[0] == 'v' && p[1] == 'n' ) { TerraFloat3 v = { 0, 0, 0 }; sscanf ( p + 3, "%f %f %f", &v.x, &v.y, &v.z ); PUSH ( m->nrm, v ); }
        else if ( p[0] == 'v' && p[1] == 't' ) { TerraFloat2 v = { 0, 0 }; sscanf ( p + 3, "%f %f", &v.x, &v.y ); PUSH ( m->uv, v ); }
        else if ( p[0] == 'f' && isspace ( ( unsigned char ) p[1] ) ) {
            int vi[64], ti[64], ni[64], cnt = 0;
            char* tok = strtok ( p + 2, " \t\r\n" );
            while ( tok && cnt < 64 ) {
                int a = 0, b = 0, c = 0;
                if ( sscanf ( tok, "%d/%d/%d", &a, &b, &c ) == 3 ) {}
                else if ( sscanf ( tok, "%d//%d", &a, &c ) == 2 ) { b = 0; }
                else if ( sscanf ( tok, "%d/%d", &a, &b ) == 2 ) { c = 0; }
                else { sscanf ( tok, "%d", &a ); b = c = 0; }
                vi[cnt] = fix_index ( a, m->pos.n ); ti[cnt] = fix_index ( b, m->uv.n ); ni[cnt] = fix_index ( c, m->nrm.n ); ++cnt;
                tok = strtok ( NULL, " \t\r\n" );
            }
            if ( cnt >= 3 ) {
                NEED_GROUP();
                Poly pl = { m->corners.n, ( size_t ) cnt, ( int ) m->groups.n - 1 };
                int ok = 1;
                for ( int k = 0; k < cnt; ++k ) if ( vi[k] < 0 ) ok = 0;
                if ( ok ) { for ( int k = 0; k < cnt; ++k ) { Corner c = { vi[k], ti[k], ni[k] }; PUSH ( m->corners, c ); } PUSH ( m->polys, pl ); }
            }
            for ( int k = 1; k + 1 < cnt; ++k ) {       /* fan triangulation */
                Face fc = { { vi[0], vi[k], vi[k + 1] }, { ti[0], ti[k], ti[k + 1] }, { ni[0], ni[k], ni[k + 1] }, cur_mtl };
                if ( fc.v[0] >= 0 && fc.v[1] >= 0 && fc.v[2] >= 0 ) PUSH ( m->faces, fc );
            }
        } else if ( strncmp ( p, "usemtl", 6 ) == 0 ) { char name[128] = ""; sscanf ( p + 6, "%127s", name ); cur_mtl = find_mtl ( m, name ); NEED_GROUP(); m->groups.d[m->groups.n - 1].mtl = cur_mtl; }
        else if ( ( p[0] == 'g' || p[0] == 'o' ) && p[1] == ' ' ) { Group g = { -1, -1 }; PUSH ( m->groups, g ); }       /* Apollo.h:1205-1218: groups and objects alike open a mesh */
        else if ( p[0] == 's' && p[1] == ' ' ) { NEED_GROUP(); m->groups.d[m->groups.n - 1].smooth = p[2] == '1' ? 1 : 0; }           /* Apollo.h:1225-1240: smooth iff the character after "s " is '1' */
        else if ( strncmp ( p, "mtllib", 6 ) == 0 ) { char name[512] = ""; sscanf ( p + 6, "%511s", name ); load_mtl ( m, dir, name ); }
    }
    fclose ( f );
    return m->faces.n > 0;
}

/* ---- scene construction ---------------------------------------------------------------------- */
static TerraFloat3 flipz ( TerraFloat3 v, int flip ) { if ( flip ) v.z = -v.z; return v; }

static void set_material ( TerraObject* o, const Mtl* mt, int apollo );
static void dump_object ( FILE* f, size_t j, const TerraObject* o );

static HTerraScene build_scene ( Model* m, int flip, FILE* dump ) {
    /* smooth normals per position for faces without vn */
    TerraFloat3* smooth = calloc ( m->pos.n ? m->pos.n : 1, sizeof ( TerraFloat3 ) );
    for ( size_t i = 0; i < m->faces.n; ++i ) {
        Face* f = &m->faces.d[i];
        TerraFloat3 a = m->pos.d[f->v[0]], b = m->pos.d[f->v[1]], c = m->pos.d[f->v[2]];
        TerraFloat3 e1 = terra_subf3 ( &b, &a ), e2 = terra_subf3 ( &c, &a ), n = terra_crossf3 ( &e1, &e2 );   /* area weighted */
        for ( int k = 0; k < 3; ++k ) smooth[f->v[k]] = terra_addf3 ( &smooth[f->v[k]], &n );
    }
    HTerraScene scene = terra_scene_create();
    int groups = ( int ) m->mtls.n + 1;                      /* one object per material; the last holds faces without one */
    for ( int g = 0; g < groups; ++g ) {
        int want = g < ( int ) m->mtls.n ? g : -1;
        size_t cnt = 0;
        for ( size_t i = 0; i < m->faces.n; ++i ) if ( m->faces.d[i].mtl == want ) ++cnt;
        if ( !cnt ) continue;
        TerraObject* o = terra_scene_add_object ( scene, cnt );
        size_t k = 0;
        for ( size_t i = 0; i < m->faces.n; ++i ) {
            Face* f = &m->faces.d[i];
            if ( f->mtl != want ) continue;
            /* flipping z mirrors the mesh; swapping b and c restores the winding (Scene.cpp:90-93) */
            int order[3] = { 0, flip ? 2 : 1, flip ? 1 : 2 };
            TerraFloat3 P[3], N[3]; TerraFloat2 T[3];
            for ( int c = 0; c < 3; ++c ) {
                int s = order[c];
                P[c] = flipz ( m->pos.d[f->v[s]], flip );
                TerraFloat3 n = f->n[s] >= 0 ? m->nrm.d[f->n[s]] : smooth[f->v[s]];
                float len = terra_lenf3 ( &n );
                N[c] = len > 0.f ? terra_normf3 ( &n ) : terra_f3_set ( 0.f, 1.f, 0.f );
                N[c] = flipz ( N[c], flip );
                T[c] = f->t[s] >= 0 ? m->uv.d[f->t[s]] : terra_f2_set ( 0.f, 0.f );
            }
            o->triangles[k].a = P[0]; o->triangles[k].b = P[1]; o->triangles[k].c = P[2];
            o->properties[k].normal_a = N[0]; o->properties[k].normal_b = N[1]; o->properties[k].normal_c = N[2];
            o->properties[k].texcoord_a = T[0]; o->properties[k].texcoord_b = T[1]; o->properties[k].texcoord_c = T[2];
            ++k;
        }
        Mtl def; memset ( &def, 0, sizeof def ); def.kd[0] = def.kd[1] = def.kd[2] = 0.7f; def.ns = 1.f; def.illum = ILLUM_INVALID;
        set_material ( o, want >= 0 ? &m->mtls.d[want] : &def, 0 );
        if ( dump ) dump_object ( dump, terra_scene_count_objects ( scene ) - 1, o );
    }
    free ( smooth );
    return scene;
}

/* ---- the reference importer's policy (see the header comment) ------------------------------------ */
typedef struct { TerraFloat3 pos; TerraFloat2 tex; TerraFloat3 norm; VEC ( unsigned ) adj; } AVert;
typedef struct { unsigned bits[3]; unsigned value; int used; } VSlot;
static unsigned pos_hash ( const unsigned* b ) { unsigned h = 2166136261u; for ( int i = 0; i < 3; ++i ) { h ^= b[i]; h *= 16777619u; h ^= h >> 13; } return h; }

/* Material class -> preset. apollo = 1: the reference client's switch (satellite/src/Scene.cpp:193-230): APOLLO_SPECULAR -> Phong; mirror, pbr and
   everything else (disney, and the invalid class of a material without an Apollo `illum` word) fall through its warnings to diffuse.
   apollo = 0 (--normals file, this repo's policy for standard MTL files): `illum specular`, or no Apollo word and Ks > 0, -> Phong. */
static void set_material ( TerraObject* o, const Mtl* mt, int apollo ) {
    TerraFloat3 kd = terra_f3_setv ( mt->kd ), ks = terra_f3_setv ( mt->ks ), ke = terra_f3_setv ( mt->ke ), zero = terra_f3_zero;
    o->material.ior = 1.5f;                                /* Scene.cpp:187 */
    terra_attribute_init_constant ( &o->material.emissive, &ke );
    int phong = mt->illum == ILLUM_SPECULAR;
    if ( !apollo && mt->illum == ILLUM_INVALID ) phong = ks.x + ks.y + ks.z > 0.f;
    if ( apollo && mt->illum != ILLUM_SPECULAR && mt->illum != ILLUM_DIFFUSE ) {      /* the reference warns and goes on (Scene.cpp:215-220) */
        static const char* what[] = { "", "", "mirror", "pbr", "disney", "unclassified (no Apollo `illum` word)" };
        fprintf ( stderr, "terra_headless: unsupported %s material(%s). Defaulting to diffuse\n", what[mt->illum], mt->name[0] ? mt->name : "<none>" );
    }
    if ( phong ) {                                         /* specular -> Phong (Scene.cpp:193-213) */
        TerraFloat3 ns = terra_f3_set1 ( mt->ns );
        terra_attribute_init_constant ( &o->material.attributes[TERRA_PHONG_ALBEDO], &kd );
        terra_attribute_init_constant ( &o->material.attributes[TERRA_PHONG_SPECULAR_COLOR], &ks );
        terra_attribute_init_constant ( &o->material.attributes[TERRA_PHONG_SPECULAR_INTENSITY], &ns );
        terra_attribute_init_constant ( &o->material.attributes[TERRA_PHONG_SAMPLE_PICK], &zero );
        o->material.attributes_count = TERRA_PHONG_END;
        terra_bsdf_phong_init ( &o->material.bsdf );
    } else {                                               /* everything else -> diffuse (Scene.cpp:215-230) */
        terra_attribute_init_constant ( &o->material.attributes[TERRA_DIFFUSE_ALBEDO], &kd );
        o->material.attributes_count = TERRA_DIFFUSE_END;
        terra_bsdf_diffuse_init ( &o->material.bsdf );
    }
}

static HTerraScene build_scene_apollo ( Model* m, int flip, FILE* dump ) {
    VEC ( AVert ) verts; memset ( &verts, 0, sizeof verts );
    VEC ( unsigned ) idx; memset ( &idx, 0, sizeof idx );
    VEC ( int ) tri_group; memset ( &tri_group, 0, sizeof tri_group );
    size_t cap = 64; while ( cap < 4 * ( m->corners.n + 1 ) ) cap *= 2;
    VSlot* table = calloc ( cap, sizeof ( VSlot ) );
    int zero_used = 0; unsigned zero_value = 0;
    unsigned face_count = 0;
    for ( size_t pi = 0; pi < m->polys.n; ++pi ) {
        const Poly* pl = &m->polys.d[pi];
        const int smooth = m->groups.d[pl->group].smooth;
        unsigned first = 0, last = 0;
        for ( size_t j = 0; j < pl->count; ++j ) {
            const Corner* c = &m->corners.d[pl->first + j];
            TerraFloat3 pos = flipz ( m->pos.d[c->v], flip );                     /* z is negated when the `v` line is read (Apollo.h:1167) */
            unsigned bits[3]; memcpy ( bits, &pos, 12 );
            for ( int k = 0; k < 3; ++k ) if ( bits[k] == 0x80000000u ) bits[k] = 0;      /* float equality: -0 == +0 */
            const int is_zero = !bits[0] && !bits[1] && !bits[2];
            unsigned index = 0; int dup = 0;
            size_t slot = pos_hash ( bits ) & ( cap - 1 );
            if ( smooth == 1 ) {                                                  /* lookup only inside smooth groups (Apollo.h:1349-1356) */
                if ( is_zero ) { if ( zero_used ) { index = zero_value; dup = 1; } }
                else for ( size_t s2 = slot; table[s2].used; s2 = ( s2 + 1 ) & ( cap - 1 ) ) if ( !memcmp ( table[s2].bits, bits, 12 ) ) { index = table[s2].value; dup = 1; break; }      /* the first one inserted */
            }
            if ( !dup ) {
                AVert v; memset ( &v, 0, sizeof v );
                v.pos = pos; v.tex = c->t >= 0 ? m->uv.d[c->t] : terra_f2_set ( 0.f, 0.f );
                PUSH ( verts, v ); index = ( unsigned ) verts.n - 1;
                if ( is_zero ) { zero_used = 1; zero_value = index; }              /* the all-zero key keeps the LAST vertex (Apollo.h:2180-2184) */
                else { size_t s2 = slot; while ( table[s2].used ) s2 = ( s2 + 1 ) & ( cap - 1 ); memcpy ( table[s2].bits, bits, 12 ); table[s2].value = index; table[s2].used = 1; }
            }
            {   /* adjacency: the triangle being formed while this corner is parsed (Apollo.h:1373-1381) */
                AVert* v = &verts.d[index]; int seen = 0;
                for ( size_t a = 0; a < v->adj.n; ++a ) if ( v->adj.d[a] == face_count ) seen = 1;
                if ( !seen ) PUSH ( v->adj, face_count );
            }
            if ( j == 0 ) first = index;
            if ( j >= 3 ) { PUSH ( idx, first ); PUSH ( idx, last ); }             /* fan: (first, previous, current) */
            PUSH ( idx, index );
            if ( j >= 2 ) { last = index; PUSH ( tri_group, pl->group ); ++face_count; }
        }
    }
    if ( flip ) for ( size_t i = 0; i + 2 < idx.n; i += 3 ) { unsigned t = idx.d[i]; idx.d[i] = idx.d[i + 2]; idx.d[i + 2] = t; }      /* Apollo.h:1412-1418 */
    TerraFloat3* fn = malloc ( ( face_count ? face_count : 1 ) * sizeof ( TerraFloat3 ) );
    for ( unsigned i = 0; i < face_count; ++i ) {                                  /* Apollo.h:1438-1462 */
        TerraFloat3 v0 = verts.d[idx.d[3 * i]].pos, v1 = verts.d[idx.d[3 * i + 1]].pos, v2 = verts.d[idx.d[3 * i + 2]].pos;
        TerraFloat3 e01 = { v1.x - v0.x, v1.y - v0.y, v1.z - v0.z }, e02 = { v2.x - v0.x, v2.y - v0.y, v2.z - v0.z };
        TerraFloat3 n = { e01.y * e02.z - e01.z * e02.y, e01.z * e02.x - e01.x * e02.z, e01.x * e02.y - e01.y * e02.x };
        float len = sqrtf ( n.x * n.x + n.y * n.y + n.z * n.z );
        n.x /= len; n.y /= len; n.z /= len;
        fn[i] = n;
    }
    unsigned char* done = calloc ( verts.n ? verts.n : 1, 1 );
    for ( size_t g = 0; g < m->groups.n; ++g ) {                                   /* Apollo.h:1465-1541, group after group */
        memset ( done, 0, verts.n );
        for ( unsigned i = 0; i < face_count; ++i ) {
            if ( tri_group.d[i] != ( int ) g ) continue;
            for ( int k = 0; k < 3; ++k ) {
                AVert* v = &verts.d[idx.d[3 * i + k]];
                if ( m->groups.d[g].smooth == 0 ) { v->norm = fn[i]; continue; }
                if ( done[idx.d[3 * i + k]] ) continue;
                done[idx.d[3 * i + k]] = 1;
                TerraFloat3 sum = { 0.f, 0.f, 0.f };
                for ( size_t a = 0; a < v->adj.n; ++a ) { sum.x += fn[v->adj.d[a]].x; sum.y += fn[v->adj.d[a]].y; sum.z += fn[v->adj.d[a]].z; }
                float len = sqrtf ( sum.x * sum.x + sum.y * sum.y + sum.z * sum.z );
                sum.x /= len; sum.y /= len; sum.z /= len;
                v->norm = sum;
            }
        }
    }
    HTerraScene scene = terra_scene_create();
    for ( size_t g = 0; g < m->groups.n; ++g ) {
        size_t cnt = 0;
        for ( unsigned i = 0; i < face_count; ++i ) if ( tri_group.d[i] == ( int ) g ) ++cnt;
        if ( !cnt ) continue;
        TerraObject* o = terra_scene_add_object ( scene, cnt );
        size_t k = 0;
        for ( unsigned i = 0; i < face_count; ++i ) {
            if ( tri_group.d[i] != ( int ) g ) continue;
            const AVert* a = &verts.d[idx.d[3 * i]]; const AVert* b = &verts.d[idx.d[3 * i + 1]]; const AVert* c = &verts.d[idx.d[3 * i + 2]];
            o->triangles[k].a = a->pos; o->triangles[k].b = b->pos; o->triangles[k].c = c->pos;
            o->properties[k].normal_a = a->norm; o->properties[k].normal_b = b->norm; o->properties[k].normal_c = c->norm;
            o->properties[k].texcoord_a = a->tex; o->properties[k].texcoord_b = b->tex; o->properties[k].texcoord_c = c->tex;
            ++k;
        }
        Mtl def; memset ( &def, 0, sizeof def ); def.kd[0] = def.kd[1] = def.kd[2] = 0.7f; def.ns = 1.f; def.illum = ILLUM_INVALID;
        set_material ( o, m->groups.d[g].mtl >= 0 ? &m->mtls.d[m->groups.d[g].mtl] : &def, 1 );
        if ( dump ) dump_object ( dump, terra_scene_count_objects ( scene ) - 1, o );
    }
    for ( size_t i = 0; i < verts.n; ++i ) free ( verts.d[i].adj.d );
    free ( verts.d ); free ( idx.d ); free ( tri_group.d ); free ( table ); free ( fn ); free ( done );
    return scene;
}

/* --dump-scene: every object exactly as it was filled through terra_scene_add_object (text, %.9g round-trips a float): for the loader's tests */
static void dump_object ( FILE* f, size_t j, const TerraObject* o ) {
    fprintf ( f, "object %zu triangles %zu attributes %zu\n", j, o->triangles_count, o->material.attributes_count );
    {   /* the material as it was filled: preset (by its attribute count), constant attribute values in slot order, emissive, ior */
        const TerraMaterial* mt = &o->material;
        fprintf ( f, "m %s", mt->attributes_count == TERRA_PHONG_END ? "phong" : "diffuse" );
        for ( size_t a = 0; a < mt->attributes_count; ++a ) fprintf ( f, " %.9g %.9g %.9g", mt->attributes[a].value.x, mt->attributes[a].value.y, mt->attributes[a].value.z );
        fprintf ( f, " e %.9g %.9g %.9g ior %.9g\n", mt->emissive.value.x, mt->emissive.value.y, mt->emissive.value.z, mt->ior );
    }
    for ( size_t i = 0; i < o->triangles_count; ++i ) {
        const TerraTriangle* t = &o->triangles[i]; const TerraTriangleProperties* q = &o->properties[i];
        fprintf ( f, "t %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g n %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g uv %.9g %.9g %.9g %.9g %.9g %.9g\n",
                  t->a.x, t->a.y, t->a.z, t->b.x, t->b.y, t->b.z, t->c.x, t->c.y, t->c.z,
                  q->normal_a.x, q->normal_a.y, q->normal_a.z, q->normal_b.x, q->normal_b.y, q->normal_b.z, q->normal_c.x, q->normal_c.y, q->normal_c.z,
                  q->texcoord_a.x, q->texcoord_a.y, q->texcoord_b.x, q->texcoord_b.y, q->texcoord_c.x, q->texcoord_c.y );
    }
}

/* ---- image writers --------------------------------------------------------------------------- */
static unsigned char to_byte ( float v ) { v = v < 0.f ? 0.f : ( v > 1.f ? 1.f : v ); return ( unsigned char ) ( v * 255.f ); }   /* Visualization.cpp:331-338 */

static int write_ppm ( const char* path, const TerraFramebuffer* fb ) {
    FILE* f = fopen ( path, "wb" ); if ( !f ) return 0;
    fprintf ( f, "P6\n%zu %zu\n255\n", fb->width, fb->height );
    for ( size_t i = 0; i < fb->width * fb->height; ++i ) { unsigned char px[3] = { to_byte ( fb->pixels[i].x ), to_byte ( fb->pixels[i].y ), to_byte ( fb->pixels[i].z ) }; fwrite ( px, 1, 3, f ); }
    return fclose ( f ) == 0;
}
static int write_pfm ( const char* path, const TerraFramebuffer* fb ) {
    FILE* f = fopen ( path, "wb" ); if ( !f ) return 0;
    fprintf ( f, "PF\n%zu %zu\n-1.0\n", fb->width, fb->height );
    for ( size_t y = fb->height; y-- > 0; ) fwrite ( &fb->pixels[y * fb->width], sizeof ( TerraFloat3 ), fb->width, f );   /* bottom row first */
    return fclose ( f ) == 0;
}
static int write_hdr ( const char* path, const TerraFramebuffer* fb ) {       /* Radiance RGBE, flat scanlines */
    FILE* f = fopen ( path, "wb" ); if ( !f ) return 0;
    fprintf ( f, "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %zu +X %zu\n", fb->height, fb->width );
    for ( size_t i = 0; i < fb->width * fb->height; ++i ) {
        float r = fb->pixels[i].x, g = fb->pixels[i].y, b = fb->pixels[i].z, mx = r > g ? ( r > b ? r : b ) : ( g > b ? g : b );
        unsigned char px[4] = { 0, 0, 0, 0 };
        if ( mx > 1e-32f ) { int e; float s = frexpf ( mx, &e ) * 256.f / mx; px[0] = ( unsigned char ) ( r * s ); px[1] = ( unsigned char ) ( g * s ); px[2] = ( unsigned char ) ( b * s ); px[3] = ( unsigned char ) ( e + 128 ); }
        fwrite ( px, 1, 4, f );
    }
    return fclose ( f ) == 0;
}
/* PNG with stored (uncompressed) deflate blocks: valid everywhere, no zlib needed */
static unsigned long crc_table[256];
static unsigned long crc32_of ( unsigned long c, const unsigned char* p, size_t n ) {
    if ( !crc_table[1] ) for ( unsigned long i = 0; i < 256; ++i ) { unsigned long k = i; for ( int j = 0; j < 8; ++j ) k = k & 1 ? 0xedb88320ul ^ ( k >> 1 ) : k >> 1; crc_table[i] = k; }
    c ^= 0xfffffffful;
    for ( size_t i = 0; i < n; ++i ) c = crc_table[ ( c ^ p[i] ) & 0xff] ^ ( c >> 8 );
    return c ^ 0xfffffffful;
}
static void be32 ( unsigned char* p, unsigned long v ) { p[0] = ( unsigned char ) ( v >> 24 ); p[1] = ( unsigned char ) ( v >> 16 ); p[2] = ( unsigned char ) ( v >> 8 ); p[3] = ( unsigned char ) v; }
static void png_chunk ( FILE* f, const char* type, const unsigned char* data, size_t n ) {
    unsigned char hdr[8]; be32 ( hdr, ( unsigned long ) n ); memcpy ( hdr + 4, type, 4 ); fwrite ( hdr, 1, 8, f );
    if ( n ) fwrite ( data, 1, n, f );
    unsigned char* tmp = malloc ( n + 4 ); memcpy ( tmp, type, 4 ); if ( n ) memcpy ( tmp + 4, data, n );
    unsigned long c = crc32_of ( 0, tmp, n + 4 ); free ( tmp );       /* CRC covers type + data */
    unsigned char tail[4]; be32 ( tail, c ); fwrite ( tail, 1, 4, f );
}
static int write_png ( const char* path, const TerraFramebuffer* fb ) {
    FILE* f = fopen ( path, "wb" ); if ( !f ) return 0;
    const size_t W = fb->width, H = fb->height, row = 1 + 3 * W, raw_n = row * H;
    unsigned char* raw = malloc ( raw_n );
    for ( size_t y = 0; y < H; ++y ) {
        raw[y * row] = 0;       /* filter: none */
        for ( size_t x = 0; x < W; ++x ) { const TerraFloat3* p = &fb->pixels[y * W + x]; unsigned char* q = raw + y * row + 1 + 3 * x; q[0] = to_byte ( p->x ); q[1] = to_byte ( p->y ); q[2] = to_byte ( p->z ); }
    }
    size_t blocks = ( raw_n + 65534 ) / 65535, zn = 2 + raw_n + 5 * blocks + 4;
    unsigned char* z = malloc ( zn ); size_t o = 0;
    z[o++] = 0x78; z[o++] = 0x01;
    unsigned long a = 1, b = 0;
    for ( size_t i = 0; i < raw_n; ++i ) { a = ( a + raw[i] ) % 65521; b = ( b + a ) % 65521; }
    for ( size_t off = 0; off < raw_n; off += 65535 ) {
        size_t n = raw_n - off < 65535 ? raw_n - off : 65535;
        z[o++] = off + n == raw_n ? 1 : 0; z[o++] = ( unsigned char ) ( n & 0xff ); z[o++] = ( unsigned char ) ( n >> 8 ); z[o++] = ( unsigned char ) ( ~n & 0xff ); z[o++] = ( unsigned char ) ( ( ~n >> 8 ) & 0xff );
        memcpy ( z + o, raw + off, n ); o += n;
    }
    be32 ( z + o, ( b << 16 ) | a ); o += 4;
    static const unsigned char sig[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
    fwrite ( sig, 1, 8, f );
    unsigned char ihdr[13]; be32 ( ihdr, ( unsigned long ) W ); be32 ( ihdr + 4, ( unsigned long ) H ); ihdr[8] = 8; ihdr[9] = 2; ihdr[10] = ihdr[11] = ihdr[12] = 0;
    png_chunk ( f, "IHDR", ihdr, 13 ); png_chunk ( f, "IDAT", z, o ); png_chunk ( f, "IEND", NULL, 0 );
    free ( raw ); free ( z );
    return fclose ( f ) == 0;
}
static int write_image ( const char* path, const TerraFramebuffer* fb ) {
    const char* ext = strrchr ( path, '.' );
    if ( ext && strcmp ( ext, ".ppm" ) == 0 ) return write_ppm ( path, fb );
    if ( ext && strcmp ( ext, ".pfm" ) == 0 ) return write_pfm ( path, fb );
    if ( ext && strcmp ( ext, ".hdr" ) == 0 ) return write_hdr ( path, fb );
    return write_png ( path, fb );      /* png assumed by default, as Visualization.cpp:313-316 */
}

/* ---- main ------------------------------------------------------------------------------------ */
static int pick ( const char* v, const char* const* names, int n, int dflt ) { for ( int i = 0; i < n; ++i ) if ( strcmp ( v, names[i] ) == 0 ) return i; return dflt; }

static const char* kHelp =
    "usage: terra_headless scene.obj out.{png,ppm,pfm,hdr} [options]\n"
    "  --width W --height H --spp N --bounces N --integrator simple|direct|mis|normals|depth --tonemap none|linear|reinhard|filmic|uncharted2\n"
    "  --camera px py pz dx dy dz --fov deg --exposure e --gamma g --jitter j --no-flip-z --normals apollo|file\n"
    "  --dump-scene file --no-render --fast-tree | --replica-tree --sample-split n --seed n --tile n\n"
    "  --gpus N (libterra_amd.so only): the scene is replicated on the first N GPUs and the frame's 64-pixel tiles are dealt to them from this one process, one RCCL\n"
    "           gather at the end (terra_amd_set_devices / terra_amd_render_multi); N = 1 runs the same calls on one GPU. Without it: terra_render() on one GPU.\n"
    "OBJ/MTL import (--normals apollo, the default): the policy of the reference client's importer (satellite/include/Apollo.h under the\n"
    "options of satellite/src/Scene.cpp:83-93) RESTATED in this tool and pinned by hand-derived fixtures -- restated, not executed: Apollo.h\n"
    "does not compile with this image's toolchains. Everything after the TerraObject fill (commit, render, export) is the pinned path.\n";

int main ( int argc, char** argv ) {
    for ( int i = 1; i < argc; ++i ) if ( !strcmp ( argv[i], "--help" ) || !strcmp ( argv[i], "-h" ) ) { fputs ( kHelp, stdout ); return 0; }
    if ( argc < 3 ) { fprintf ( stderr, "usage: terra_headless scene.obj out.{png,ppm,pfm,hdr} [options]\n" ); return 64; }
    if ( terra_amd_init ) ( void ) terra_amd_init();      /* before anything touches the GPU (libterra_amd.so only) */
    size_t W = 800, H = 600, spp = 8, bounces = 4, tile = 0;     /* defaults of satellite/include/Config.hpp:19-113 */
    int integrator = kTerraIntegratorDirect, tonemap = kTerraTonemappingOperatorLinear, flip = 1, fast = -1, have_seed = 0, split = -1, apollo = 1, gpus = 0;
    const char* dump_path = NULL; int no_render = 0;
    float fov = 45.f, exposure = 1.f, gamma = 2.2f, jitter = 0.f;
    unsigned long long seed = 0;
    TerraCamera cam; cam.position = terra_f3_set ( 0.f, 1.f, -3.4f ); cam.direction = terra_f3_set ( 0.f, 0.f, 1.f ); cam.up = terra_f3_set ( 0.f, 1.f, 0.f );
    for ( int i = 3; i < argc; ++i ) {
        const char* a = argv[i];
#define NEXT() ( i + 1 < argc ? argv[++i] : "" )
        if ( !strcmp ( a, "--width" ) ) W = ( size_t ) atol ( NEXT() );
        else if ( !strcmp ( a, "--height" ) ) H = ( size_t ) atol ( NEXT() );
        else if ( !strcmp ( a, "--spp" ) ) spp = ( size_t ) atol ( NEXT() );
        else if ( !strcmp ( a, "--bounces" ) ) bounces = ( size_t ) atol ( NEXT() );
        else if ( !strcmp ( a, "--tile" ) ) tile = ( size_t ) atol ( NEXT() );
        else if ( !strcmp ( a, "--fov" ) ) fov = ( float ) atof ( NEXT() );
        else if ( !strcmp ( a, "--exposure" ) ) exposure = ( float ) atof ( NEXT() );
        else if ( !strcmp ( a, "--gamma" ) ) gamma = ( float ) atof ( NEXT() );
        else if ( !strcmp ( a, "--jitter" ) ) jitter = ( float ) atof ( NEXT() );
        else if ( !strcmp ( a, "--seed" ) ) { seed = strtoull ( NEXT(), NULL, 0 ); have_seed = 1; }
        else if ( !strcmp ( a, "--no-flip-z" ) ) flip = 0;
        else if ( !strcmp ( a, "--fast-tree" ) ) fast = 1;
        else if ( !strcmp ( a, "--auto-tree" ) ) fast = 2;
        else if ( !strcmp ( a, "--replica-tree" ) ) fast = 0;
        else if ( !strcmp ( a, "--normals" ) ) { const char* v = NEXT(); if ( !strcmp ( v, "apollo" ) ) apollo = 1; else if ( !strcmp ( v, "file" ) ) apollo = 0; else { fprintf ( stderr, "terra_headless: --normals apollo|file\n" ); return 64; } }
        else if ( !strcmp ( a, "--dump-scene" ) ) dump_path = NEXT();
        else if ( !strcmp ( a, "--no-render" ) ) no_render = 1;
        else if ( !strcmp ( a, "--sample-split" ) ) split = atoi ( NEXT() );
        else if ( !strcmp ( a, "--gpus" ) ) gpus = atoi ( NEXT() );
        else if ( !strcmp ( a, "--integrator" ) ) { static const char* const n[] = { "simple", "direct", "mis", "mono", "depth", "normals", "misweights" }; integrator = pick ( NEXT(), n, 7, integrator ); }
        else if ( !strcmp ( a, "--tonemap" ) ) { static const char* const n[] = { "none", "linear", "reinhard", "filmic", "uncharted2" }; tonemap = pick ( NEXT(), n, 5, tonemap ); }
        else if ( !strcmp ( a, "--camera" ) && i + 6 < argc ) {
            cam.position = terra_f3_set ( ( float ) atof ( argv[i + 1] ), ( float ) atof ( argv[i + 2] ), ( float ) atof ( argv[i + 3] ) );
            cam.direction = terra_f3_set ( ( float ) atof ( argv[i + 4] ), ( float ) atof ( argv[i + 5] ), ( float ) atof ( argv[i + 6] ) ); i += 6;
        } else { fprintf ( stderr, "terra_headless: unknown option %s\n", a ); return 64; }
    }
    cam.fov = fov;
    Model m; memset ( &m, 0, sizeof m );
    if ( !load_obj ( &m, argv[1] ) ) { fprintf ( stderr, "terra_headless: no faces in %s\n", argv[1] ); return 66; }
    FILE* dump = dump_path ? fopen ( dump_path, "w" ) : NULL;
    if ( dump_path && !dump ) { fprintf ( stderr, "terra_headless: cannot write %s\n", dump_path ); return 73; }
    HTerraScene scene = apollo ? build_scene_apollo ( &m, flip, dump ) : build_scene ( &m, flip, dump );
    if ( dump ) fclose ( dump );
    if ( no_render ) { printf ( "%s: %zu objects\n", argv[1], terra_scene_count_objects ( scene ) ); terra_scene_destroy ( scene ); return 0; }
    TerraSceneOptions* o = terra_scene_get_options ( scene );
    TerraFloat3 env = terra_f3_set ( 0.4f, 0.52f, 1.f );
    terra_attribute_init_constant ( &o->environment_map, &env );
    o->tonemapping_operator = ( TerraTonemappingOperator ) tonemap; o->accelerator = kTerraAcceleratorBVH; o->sampling_method = kTerraSamplingMethodRandom;
    o->integrator = ( TerraIntegrator ) integrator; o->subpixel_jitter = jitter; o->samples_per_pixel = spp; o->bounces = bounces; o->strata = 4;
    o->manual_exposure = exposure; o->gamma = gamma;
    if ( fast >= 0 && terra_amd_set_tree_mode ) terra_amd_set_tree_mode ( scene, fast );
    if ( split >= 0 && terra_amd_set_sample_split ) terra_amd_set_sample_split ( scene, split );
    if ( have_seed && terra_amd_set_frame_seed ) terra_amd_set_frame_seed ( scene, seed );
    if ( gpus > 0 ) {          /* several GPUs from this one process: the first `gpus` devices, the first of them primary */
        int devs[64];
        if ( !terra_amd_set_devices || !terra_amd_render_multi ) { fprintf ( stderr, "terra_headless: --gpus needs libterra_amd.so\n" ); return 64; }
        if ( gpus > 64 || ( terra_amd_device_count && gpus > terra_amd_device_count() ) ) { fprintf ( stderr, "terra_headless: --gpus %d but %d visible\n", gpus, terra_amd_device_count ? terra_amd_device_count() : 0 ); return 69; }
        for ( int k = 0; k < gpus; ++k ) devs[k] = k;
        if ( terra_amd_set_devices ( devs, gpus ) != 0 ) { fprintf ( stderr, "terra_headless: %s\n", terra_amd_last_error ? terra_amd_last_error() : "terra_amd_set_devices failed" ); return 69; }
    }
    terra_scene_commit ( scene );
    TerraFramebuffer fb;
    if ( !terra_framebuffer_create ( &fb, W, H ) ) { fprintf ( stderr, "terra_headless: bad framebuffer size\n" ); return 65; }
    if ( gpus > 0 ) ( void ) terra_amd_render_multi ( &cam, scene, &fb, 0, 0, W, H, tile );      /* (tile = the shard's tile size here; 0 = 64) */
    else if ( tile == 0 ) terra_render ( &cam, scene, &fb, 0, 0, W, H );
    else for ( size_t y = 0; y < H; y += tile ) for ( size_t x = 0; x < W; x += tile ) terra_render ( &cam, scene, &fb, x, y, W - x < tile ? W - x : tile, H - y < tile ? H - y : tile );
    if ( terra_amd_last_error && *terra_amd_last_error() ) { fprintf ( stderr, "terra_headless: %s\n", terra_amd_last_error() ); return 70; }
    if ( !write_image ( argv[2], &fb ) ) { fprintf ( stderr, "terra_headless: cannot write %s\n", argv[2] ); return 73; }
    printf ( "%s: %zu triangles, %zu materials -> %s (%zux%zu, %zu spp)\n", argv[1], m.faces.n, m.mtls.n, argv[2], W, H, spp );
    terra_framebuffer_destroy ( &fb );
    terra_scene_destroy ( scene );
    return 0;
}
