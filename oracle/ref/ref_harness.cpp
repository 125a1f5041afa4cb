// TEST INFRASTRUCTURE ONLY -- never linked into, imported by or shipped with the product.
//
// C-callable shim around the UNMODIFIED reference translation units that build in this
// container without any stand-in header:
//   kdtree.cpp ray.cpp Object.cpp intersection.cpp shapes/{triangle,sphere,plane,cube}.cpp
//   sampling/{sampling,sampler,stratified}.cpp bsdf.cpp Camera.cpp LightDistribution.cpp image.cpp
//   (+ libs/lodepng-master/lodepng.cpp, which image.cpp calls), and the header-only
//   tiny_obj_loader.h (its implementation is instantiated here exactly as scene.cpp:1 does),
//   textures.hpp, and the inline random_dir / CoordinateSystem of sampling.h / defines.h.
// (Material.cpp, Light.cpp, integrator.cpp, scene.cpp and example_scenes.cpp include
// Light.h, which needs <stb_image.h>; that header is not in the image, so those units are
// treated as unbuildable and are NOT part of this library.)
//
// The shim only constructs reference objects through their public interface, calls the
// reference functions, and copies results into flat arrays so that tests/golden/make_golden.py
// can record golden vectors. It is compiled by oracle/Makefile into oracle/_ref/libmrt_ref.so,
// which exists only in the build container (the reference sources do not travel).
#define TINYOBJLOADER_IMPLEMENTATION // scene.cpp:1 (scene.cpp itself is not buildable here)
#include "Camera.h"
#include "LightDistribution.h"
#include "Object.h"
#include "bsdf.hpp"
#include "kdtree.h"
#include "ray.h"
#include "sampling/sampling.h"
#include "sampling/stratified.h"
#include "scene.h"
#include "shapes/plane.h"
#include "shapes/sphere.h"
#include "shapes/triangle.h"
#include "textures.hpp"

#include "flatten_ref.h" // integration/: the reference-side adapter under test

#include <lodepng.h>

#include <cstdint>
#include <cstring>
#include <new>
#include <vector>

// Declared in kdtree.cpp without a header.
bool intersectAabb( Ray* theRay, vec3 min, vec3 max );

namespace {

// Scene::CreateMaterial needs Material::Material from Material.cpp (not buildable here).
// Material is trivially copyable, so a zero-filled image is copied in and the one field the
// traversal reads (m_emission, kdtree.cpp:533) is set afterwards.
size_t push_zero_material( Scene* scene, const float* emission ) {
    alignas( Material ) static unsigned char zero[sizeof( Material )] = { 0 };
    scene->m_Materials.push_back( *reinterpret_cast<const Material*>( zero ) );
    Material& m  = scene->m_Materials.back();
    m.m_emission = color3( emission[0], emission[1], emission[2] );
    m.m_UID      = scene->m_Materials.size() - 1;
    return m.m_UID;
}

vec3 v3( const float* p ) {
    return vec3( p[0], p[1], p[2] );
}

void put3( float* dst, const vec3& v ) {
    dst[0] = v.x;
    dst[1] = v.y;
    dst[2] = v.z;
}

void put_mat3( float* dst, const mat3& m ) {
    for ( int c = 0; c < 3; c++ )
        for ( int r = 0; r < 3; r++ )
            dst[3 * c + r] = m[c][r];
}

struct FlattenCtx {
    uint8_t* leaf;
    int32_t* axis;
    float* split;
    int32_t* left;
    int32_t* right;
    float* bmin;
    float* bmax;
    int32_t* ref_off;
    int32_t* ref_cnt;
    int32_t* refs;
    int64_t n_nodes = 0;
    int64_t n_refs  = 0;
    int depth       = 0;
};

// Pre-order walk. With null output arrays it only counts.
int64_t walk( const KdTreeNode* node, FlattenCtx& c, int depth ) {
    int64_t id = c.n_nodes++;
    if ( depth > c.depth ) c.depth = depth;
    if ( c.leaf ) {
        c.leaf[id]  = node->leaf ? 1 : 0;
        c.axis[id]  = node->leaf ? -1 : node->axis;
        c.split[id] = node->leaf ? 0.f : node->split;
        put3( c.bmin + 3 * id, node->min );
        put3( c.bmax + 3 * id, node->max );
        c.ref_off[id] = (int32_t)c.n_refs;
        c.ref_cnt[id] = node->leaf ? (int32_t)node->objects.size() : 0;
        c.left[id] = c.right[id] = -1;
    }
    if ( node->leaf ) {
        if ( c.refs )
            for ( size_t i = 0; i < node->objects.size(); i++ )
                c.refs[c.n_refs + i] = node->objects[i];
        c.n_refs += node->objects.size();
        return id;
    }
    int64_t l = walk( node->left, c, depth + 1 );
    int64_t r = walk( node->right, c, depth + 1 );
    if ( c.leaf ) {
        c.left[id]  = (int32_t)l;
        c.right[id] = (int32_t)r;
    }
    return id;
}

} // namespace

extern "C" {

// ---------------------------------------------------------------- Transform (defines.h:67-148)
void* ref_xf_new() {
    return new Transform();
}
void ref_xf_free( void* xf ) {
    delete static_cast<Transform*>( xf );
}
void ref_xf_scale( void* xf, float sx, float sy, float sz ) {
    static_cast<Transform*>( xf )->scale( sx, sy, sz );
}
void ref_xf_rotate( void* xf, float ax, float ay, float az, float degrees ) {
    static_cast<Transform*>( xf )->rotate( vec3( ax, ay, az ), degrees );
}
void ref_xf_translate( void* xf, float x, float y, float z ) {
    static_cast<Transform*>( xf )->translate( vec3( x, y, z ) );
}
void ref_xf_get( void* xf, float* m9, float* inv9, float* t3 ) {
    auto* t = static_cast<Transform*>( xf );
    put_mat3( m9, t->getTransform() );
    put_mat3( inv9, t->getInvTransform() );
    put3( t3, t->getTranslation() );
}

// ---------------------------------------------------------------- Scene (scene.h, scene.cpp:33-110)
void* ref_scene_new() {
    return new Scene();
}

// Mirrors initSmoothTriangle (scene.cpp:85-108). The object's material index is its own
// object index, so Intersection::materialIndex of a closest hit IS the primitive id.
int ref_add_triangle( void* scene_,
                      const float* p9,
                      const float* n9,
                      const float* fn3,
                      const float* tex6,
                      void* xf,
                      const float* emission3 ) {
    auto* scene = static_cast<Scene*>( scene_ );
    size_t mat  = push_zero_material( scene, emission3 );
    auto* ret   = new Triangle( mat, *static_cast<Transform*>( xf ) );
    ret->geom.type            = TRIANGLE;
    ret->geom.triangle.p1     = v3( p9 );
    ret->geom.triangle.p2     = v3( p9 + 3 );
    ret->geom.triangle.p3     = v3( p9 + 6 );
    ret->geom.triangle.n1     = v3( n9 );
    ret->geom.triangle.n2     = v3( n9 + 3 );
    ret->geom.triangle.n3     = v3( n9 + 6 );
    ret->geom.triangle.normal = v3( fn3 );
    ret->geom.triangle.tex[0] = vec2( tex6[0], tex6[1] );
    ret->geom.triangle.tex[1] = vec2( tex6[2], tex6[3] );
    ret->geom.triangle.tex[2] = vec2( tex6[4], tex6[5] );
    scene->objects.push_back( ret );
    return (int)scene->objects.size() - 1;
}

// Mirrors initSphere (scene.cpp:33-44) including the glm::length2 column scan.
int ref_add_sphere( void* scene_, void* xf, const float* emission3, float* center_radius4 ) {
    auto* scene             = static_cast<Scene*>( scene_ );
    size_t mat              = push_zero_material( scene, emission3 );
    Transform transform     = *static_cast<Transform*>( xf );
    auto* ret               = new Sphere( mat, transform );
    ret->geom.type          = SPHERE;
    ret->geom.sphere.center = transform.transformFrom( vec3( 0 ) );
    mat3 m                  = transform.getTransform();
    float s0                = dot( m[0], m[0] );
    float s1                = dot( m[1], m[1] );
    float s2                = dot( m[2], m[2] );
    float mx                = std::max( s0, std::max( s1, s2 ) );
    ret->geom.sphere.radius = 1.f * std::sqrt( mx );
    if ( center_radius4 ) {
        put3( center_radius4, ret->geom.sphere.center );
        center_radius4[3] = ret->geom.sphere.radius;
    }
    scene->objects.push_back( ret );
    return (int)scene->objects.size() - 1;
}

// Mirrors the Plane construction used by the scenes (example_scenes.cpp:1407-1411).
int ref_add_plane( void* scene_, void* xf, const float* emission3 ) {
    auto* scene    = static_cast<Scene*>( scene_ );
    size_t mat     = push_zero_material( scene, emission3 );
    auto* ret      = new Plane( mat, *static_cast<Transform*>( xf ) );
    ret->geom.type = PLANE;
    scene->objects.push_back( ret );
    return (int)scene->objects.size() - 1;
}

// ---------------------------------------------------------------- kd-tree (kdtree.cpp:54-158)
void* ref_build( void* scene ) {
    return initKdTree( static_cast<Scene*>( scene ) );
}

void ref_tree_counts( void* tree_, int64_t* n_nodes, int64_t* n_refs, int32_t* depth ) {
    auto* tree = static_cast<KdTree*>( tree_ );
    FlattenCtx c {};
    walk( tree->root, c, 0 );
    *n_nodes = c.n_nodes;
    *n_refs  = c.n_refs;
    *depth   = c.depth;
}

void ref_tree_flatten( void* tree_,
                       uint8_t* leaf,
                       int32_t* axis,
                       float* split,
                       int32_t* left,
                       int32_t* right,
                       float* bmin,
                       float* bmax,
                       int32_t* ref_off,
                       int32_t* ref_cnt,
                       int32_t* refs ) {
    auto* tree = static_cast<KdTree*>( tree_ );
    FlattenCtx c { leaf, axis, split, left, right, bmin, bmax, ref_off, ref_cnt, refs };
    walk( tree->root, c, 0 );
}

// ---------------------------------------------------------------- intersectKdTree (kdtree.cpp:555-613)
// shadow[i] != 0 sets Ray::shadow. Outputs: hit flag, prim (= materialIndex = object index),
// t (= ray.tmax after the call), Intersection::{u,v,position,normal,isOutside}.
void ref_intersect( void* scene_,
                    void* tree_,
                    int64_t n,
                    const float* o,
                    const float* d,
                    const float* tmin,
                    const float* tmax,
                    const uint8_t* shadow,
                    uint8_t* hit,
                    int32_t* prim,
                    float* t,
                    float* u,
                    float* v,
                    float* pos,
                    float* nrm,
                    uint8_t* outside ) {
    auto* scene = static_cast<Scene*>( scene_ );
    auto* tree  = static_cast<KdTree*>( tree_ );
    for ( int64_t i = 0; i < n; i++ ) {
        Ray ray( v3( o + 3 * i ), v3( d + 3 * i ), tmin[i], tmax[i], 0 );
        ray.shadow           = shadow[i] != 0;
        ray.hasDifferentials = false;
        Intersection isect;
        isect.u = isect.v = 0.f;
        isect.materialIndex = 0;
        isect.isOutside     = false;
        isect.normal        = vec3( 0 );
        isect.position      = vec3( 0 );
        bool h              = intersectKdTree( scene, tree, &ray, &isect );
        hit[i]              = h ? 1 : 0;
        prim[i]             = h ? (int32_t)isect.materialIndex : -1;
        t[i]                = ray.tmax;
        u[i]                = isect.u;
        v[i]                = isect.v;
        put3( pos + 3 * i, isect.position );
        put3( nrm + 3 * i, isect.normal );
        outside[i] = isect.isOutside ? 1 : 0;
    }
}

// One object, no tree: Object::intersect (triangle.cpp:4-84, sphere.cpp:52-180, plane.cpp:6-68).
void ref_object_intersect( void* scene_,
                           int obj,
                           int64_t n,
                           const float* o,
                           const float* d,
                           const float* tmin,
                           const float* tmax,
                           const uint8_t* shadow,
                           uint8_t* hit,
                           float* t,
                           float* u,
                           float* v,
                           float* pos,
                           float* nrm,
                           uint8_t* outside ) {
    auto* scene = static_cast<Scene*>( scene_ );
    for ( int64_t i = 0; i < n; i++ ) {
        Ray ray( v3( o + 3 * i ), v3( d + 3 * i ), tmin[i], tmax[i], 0 );
        ray.shadow           = shadow[i] != 0;
        ray.hasDifferentials = false;
        Intersection isect;
        isect.u = isect.v = 0.f;
        isect.isOutside   = false;
        isect.normal      = vec3( 0 );
        isect.position    = vec3( 0 );
        bool h            = scene->objects[obj]->intersect( &ray, &isect );
        hit[i]            = h ? 1 : 0;
        t[i]              = ray.tmax;
        u[i]              = isect.u;
        v[i]              = isect.v;
        put3( pos + 3 * i, isect.position );
        put3( nrm + 3 * i, isect.normal );
        outside[i] = isect.isOutside ? 1 : 0;
    }
}

// intersectAabb (kdtree.cpp:162-180): returns hit flag and the clipped tmin/tmax.
void ref_intersect_aabb( int64_t n,
                         const float* o,
                         const float* d,
                         const float* tmin,
                         const float* tmax,
                         const float* bmin3,
                         const float* bmax3,
                         uint8_t* hit,
                         float* otmin,
                         float* otmax ) {
    for ( int64_t i = 0; i < n; i++ ) {
        Ray ray( v3( o + 3 * i ), v3( d + 3 * i ), tmin[i], tmax[i], 0 );
        hit[i]   = intersectAabb( &ray, v3( bmin3 ), v3( bmax3 ) ) ? 1 : 0;
        otmin[i] = ray.tmin;
        otmax[i] = ray.tmax;
    }
}

// ---------------------------------------------------------------- shape sampling
// Object::sample(inter,u,&pdf) / Area / pdf(inter,wi)  (triangle.cpp:86-104, sphere.cpp:182-236,
// Object.cpp:13-22).
void ref_object_sample( void* scene_,
                        int obj,
                        int64_t n,
                        const float* ipos,
                        const float* inrm,
                        const uint8_t* ioutside,
                        const float* u2,
                        float* opos,
                        float* onrm,
                        float* opdf ) {
    auto* scene = static_cast<Scene*>( scene_ );
    for ( int64_t i = 0; i < n; i++ ) {
        Intersection inter;
        inter.position  = v3( ipos + 3 * i );
        inter.normal    = v3( inrm + 3 * i );
        inter.isOutside = ioutside[i] != 0;
        float pdf       = -1.f; // Sphere::sample leaves pdf unwritten on its inside branch
        Intersection it = scene->objects[obj]->sample( inter, point2( u2[2 * i], u2[2 * i + 1] ), &pdf );
        put3( opos + 3 * i, it.position );
        put3( onrm + 3 * i, it.normal );
        opdf[i] = pdf;
    }
}

float ref_object_area( void* scene_, int obj ) {
    return static_cast<Scene*>( scene_ )->objects[obj]->Area();
}

void ref_object_pdf( void* scene_,
                     int obj,
                     int64_t n,
                     const float* ipos,
                     const float* inrm,
                     const float* wi,
                     float* opdf ) {
    auto* scene = static_cast<Scene*>( scene_ );
    for ( int64_t i = 0; i < n; i++ ) {
        Intersection inter;
        inter.position  = v3( ipos + 3 * i );
        inter.normal    = v3( inrm + 3 * i );
        inter.isOutside = true;
        opdf[i]         = scene->objects[obj]->pdf( inter, v3( wi + 3 * i ) );
    }
}

// ---------------------------------------------------------------- bsdf.cpp
void ref_beckmann( int64_t n, const float* ndoth, const float* alpha, float* out ) {
    for ( int64_t i = 0; i < n; i++ )
        out[i] = RDM_Beckmann( ndoth[i], alpha[i] );
}
void ref_fresnel( int64_t n, const float* ldoth, const float* ext, const float* in, float* out ) {
    for ( int64_t i = 0; i < n; i++ )
        out[i] = RDM_Fresnel( ldoth[i], ext[i], in[i] );
}
void ref_smith( int64_t n,
                const float* ldoth,
                const float* ldotn,
                const float* vdoth,
                const float* vdotn,
                const float* alpha,
                float* out ) {
    for ( int64_t i = 0; i < n; i++ )
        out[i] = RDM_Smith( ldoth[i], ldotn[i], vdoth[i], vdotn[i], alpha[i] );
}
// RDM_bsdf on a caller-filled BrdfData (bsdf.cpp:225-259). in: 5 dots, F(3), diffuse(3), alpha.
void ref_bsdf( int64_t n, const float* in12, float* out3 ) {
    for ( int64_t i = 0; i < n; i++ ) {
        const float* p = in12 + 12 * i;
        BrdfData data {};
        data.LdotH              = p[0];
        data.NdotH              = p[1];
        data.VdotH              = p[2];
        data.LdotN              = p[3];
        data.VdotN              = p[4];
        data.F                  = vec3( p[5], p[6], p[7] );
        data.diffuseReflectance = color3( p[8], p[9], p[10] );
        data.alpha              = p[11];
        data.roughness          = p[11];
        data.alphaSq            = p[11] * p[11];
        put3( out3 + 3 * i, RDM_bsdf( data, nullptr, -1 ) );
    }
}
// Inline helpers of Material.h:40-76 (header-only, so they are reachable).
void ref_schlick_f( int64_t n, const float* base3, const float* metal, const float* ldoth, float* f0_3, float* diff3, float* F3 ) {
    for ( int64_t i = 0; i < n; i++ ) {
        vec3 base = v3( base3 + 3 * i );
        vec3 f0   = baseColorToSpecularF0( base, metal[i] );
        vec3 df   = baseColorToDiffuseReflectance( base, metal[i] );
        vec3 F    = evalFresnel( f0, shadowedF90( f0 ), ldoth[i] );
        put3( f0_3 + 3 * i, f0 );
        put3( diff3 + 3 * i, df );
        put3( F3 + 3 * i, F );
    }
}

// ---------------------------------------------------------------- sampling/*.cpp, rng.h
void ref_rng_u32( uint64_t seq, int64_t n, uint32_t* out ) {
    RNG rng( seq );
    for ( int64_t i = 0; i < n; i++ )
        out[i] = rng.UniformUInt32();
}
void ref_rng_float( uint64_t seq, int64_t n, float* out ) {
    RNG rng( seq );
    for ( int64_t i = 0; i < n; i++ )
        out[i] = rng.Uniformfloat();
}
void ref_rng_bounded( uint64_t seq, int64_t n, const uint32_t* b, uint32_t* out ) {
    RNG rng( seq );
    for ( int64_t i = 0; i < n; i++ )
        out[i] = rng.UniformUInt32( b[i] );
}

// StratifiedSampler(nx,ny,true,ndims).Clone(seed): for each of `npixels` consecutive StartPixel
// calls records the camera sample and then `nget` rounds of (Get1D, Get2D, Get2D, Get2D) per
// sample -- the draw pattern of CPURenderer.cpp:50 + integrator.cpp:152-186.
// out layout: [pixel][sample][5 + 7*nget] floats.
void ref_stratified_stream( int nx, int ny, int ndims, int seed, int npixels, int nget, float* out ) {
    StratifiedSampler proto( nx, ny, true, ndims );
    std::unique_ptr<Sampler> s = proto.Clone( seed );
    int64_t stride             = 5 + 7 * nget;
    int64_t spp                = s->samplesPerPixel;
    for ( int p = 0; p < npixels; p++ ) {
        s->StartPixel( point2( p, 0 ) );
        int64_t k = 0;
        do {
            float* dst      = out + ( p * spp + k ) * stride;
            CameraSample cs = s->GetCameraSample( point2( 0, 0 ) );
            dst[0]          = cs.xy.x;
            dst[1]          = cs.xy.y;
            dst[2]          = cs.t;
            dst[3]          = cs.uv.x;
            dst[4]          = cs.uv.y;
            dst += 5;
            for ( int g = 0; g < nget; g++ ) {
                dst[0]   = s->Get1D();
                point2 a = s->Get2D();
                point2 b = s->Get2D();
                point2 c = s->Get2D();
                dst[1]   = a.x;
                dst[2]   = a.y;
                dst[3]   = b.x;
                dst[4]   = b.y;
                dst[5]   = c.x;
                dst[6]   = c.y;
                dst += 7;
            }
            k++;
        } while ( s->StartNextSample() );
    }
}

void ref_uniform_sample_triangle( int64_t n, const float* u2, float* out2 ) {
    for ( int64_t i = 0; i < n; i++ ) {
        point2 b       = UniformSampleTriangle( point2( u2[2 * i], u2[2 * i + 1] ) );
        out2[2 * i]     = b.x;
        out2[2 * i + 1] = b.y;
    }
}
void ref_concentric_disk( int64_t n, const float* u2, float* out2 ) {
    for ( int64_t i = 0; i < n; i++ ) {
        point2 b       = ConcentricSampleDisk( point2( u2[2 * i], u2[2 * i + 1] ) );
        out2[2 * i]     = b.x;
        out2[2 * i + 1] = b.y;
    }
}
void ref_power_heuristic( int64_t n, const float* f, const float* g, float* out ) {
    for ( int64_t i = 0; i < n; i++ )
        out[i] = PowerHeuristic( 1, f[i], 1, g[i] );
}
// Distribution1D(func,n): SampleDiscrete(u,&pdf) (sampling.h:69-119).
void ref_dist1d_discrete( const float* func, int n, int64_t m, const float* u, int32_t* off, float* pdf ) {
    Distribution1D d( func, n );
    for ( int64_t i = 0; i < m; i++ )
        off[i] = d.SampleDiscrete( u[i], &pdf[i] );
}
// Distribution2D(data,nu,nv).Pdf(p) (sampling.h:133-157, sampling.cpp:159-171); also returns the
// marginal integral so the table build can be pinned.
void ref_dist2d_pdf( const float* data, int nu, int nv, int64_t m, const float* p2, float* pdf ) {
    Distribution2D d( data, nu, nv );
    for ( int64_t i = 0; i < m; i++ )
        pdf[i] = d.Pdf( point2( p2[2 * i], p2[2 * i + 1] ) );
}

// ---------------------------------------------------------------- Camera.cpp:5-67
void* ref_camera_new( const float* from3,
                      const float* at3,
                      const float* up3,
                      float fov,
                      float w,
                      float h,
                      float radius,
                      float focus ) {
    // setSimpleCamera (scene.cpp:151-161)
    return new SimpleCamera( v3( from3 ), v3( at3 ), v3( up3 ), fov, ( w / h ), w, h, radius, focus );
}
void ref_camera_rays( void* cam_, int64_t n, const float* st2, const float* lens2, float* o, float* d ) {
    auto* cam = static_cast<SimpleCamera*>( cam_ );
    for ( int64_t i = 0; i < n; i++ ) {
        Ray r = cam->get_ray( st2[2 * i], st2[2 * i + 1], lens2[2 * i], lens2[2 * i + 1], vec2( 0, 0 ), false );
        put3( o + 3 * i, r.orig );
        put3( d + 3 * i, r.dir );
    }
}
void ref_camera_params( void* cam_, float* out12 ) {
    auto* cam = static_cast<SimpleCamera*>( cam_ );
    put3( out12, cam->nearPlaneTopLeft );
    put3( out12 + 3, cam->dXPixel );
    put3( out12 + 6, cam->dYPixel );
    put3( out12 + 9, cam->xDir );
}

// ---------------------------------------------------------------- textures.hpp:12-125
// texture::value(u, v, duv) -- the overload Material::eval / sample call (Material.cpp:990-993) --
// of a checker_texture of two solid colours, under the texture transform built from `xf`.
void ref_checker_value( const float* even3, const float* odd3, void* xf, int64_t n, const float* uv2, float* out3 ) {
    checker_texture tex( color3( even3[0], even3[1], even3[2] ), color3( odd3[0], odd3[1], odd3[2] ) );
    if ( xf ) tex.m_transform = *static_cast<Transform*>( xf );
    const vec3 duv[2] = { vec3( 0.01f, 0, 0 ), vec3( 0, 0.01f, 0 ) };
    const texture* t = &tex; // Material::m_texture is a texture*: the call resolves to the base overload
    for ( int64_t i = 0; i < n; i++ )
        put3( out3 + 3 * i, t->value( uv2[2 * i], uv2[2 * i + 1], duv ) );
}
// image_texture: the 8-bit RGB image is written as a PNG with the reference's own lodepng and read
// back through image_texture's constructor (loadPng, image.cpp:14-41).
int ref_image_value( const char* png_path, int w, int h, const uint8_t* rgb8, void* xf, int64_t n, const float* uv2, float* out3 ) {
    if ( lodepng_encode24_file( png_path, rgb8, (unsigned)w, (unsigned)h ) ) return -1;
    image_texture tex( png_path );
    if ( xf ) tex.m_transform = *static_cast<Transform*>( xf );
    const vec3 duv[2] = { vec3( 0.01f, 0, 0 ), vec3( 0, 0.01f, 0 ) };
    const texture* t = &tex; // Material::m_texture is a texture*: the call resolves to the base overload
    for ( int64_t i = 0; i < n; i++ )
        put3( out3 + 3 * i, t->value( uv2[2 * i], uv2[2 * i + 1], duv ) );
    return 0;
}

// ---------------------------------------------------------------- random_dir (sampling.h:199-211)
// The function draws five values from the translation unit's thread-local `engine`
// (defines.h:63-64).  For each row the engine is seeded, the five uniform01(engine) values it will
// deliver are recorded in drawing order, the engine is seeded again and random_dir is called: the
// recorded values and the result pin the arithmetic and which draw feeds which component.
void ref_random_dir( int64_t n, const float* normal3, const uint32_t* seed, float* draws5, float* out3 ) {
    for ( int64_t i = 0; i < n; i++ ) {
        engine.seed( seed[i] );
        uniform01.reset();
        for ( int k = 0; k < 5; k++ )
            draws5[5 * i + k] = uniform01( engine );
        engine.seed( seed[i] );
        uniform01.reset();
        put3( out3 + 3 * i, random_dir( v3( normal3 + 3 * i ) ) );
    }
}
// CoordinateSystem (defines.h:150-156)
void ref_coordinate_system( int64_t n, const float* v1_3, float* v2_3, float* v3_3 ) {
    for ( int64_t i = 0; i < n; i++ ) {
        vec3 a, b;
        CoordinateSystem( v3( v1_3 + 3 * i ), &a, &b );
        put3( v2_3 + 3 * i, a );
        put3( v3_3 + 3 * i, b );
    }
}

// ---------------------------------------------------------------- LightDistribution.cpp:5-12
// UniformLightDistribution over a scene with n_lights lights (only lights.size() is read), then
// lookup(p)->SampleDiscrete(u, &pdf) as integrator.cpp:152-160 calls it.
void ref_uniform_light_distribution( int n_lights, int64_t m, const float* u, int32_t* off, float* pdf ) {
    Scene scene;
    scene.lights.assign( (size_t)n_lights, nullptr );
    UniformLightDistribution d( &scene );
    const Distribution1D* dist = d.lookup( point3( 0 ) );
    for ( int64_t i = 0; i < m; i++ )
        off[i] = dist->SampleDiscrete( u[i], &pdf[i] );
}

// ---------------------------------------------------------------- tiny_obj_loader.h v1.0.6
// LoadObj exactly as readObjToTriangleMesh calls it (scene.cpp:179-187), then the per-face walk of
// addObjectsFromFile (example_scenes.cpp:32-69) flattened into arrays: for every face its three
// positions, normals (when the file has any) and texture coordinates (when it has any) and its
// material id; for every MTL material ior, diffuse, emission, specular.
// Two-call protocol: with null outputs only the counts are returned.
int ref_obj_load( const char* obj_path, const char* mtl_dir, int64_t* n_faces, int64_t* n_materials, int32_t* has_normals, int32_t* has_texcoords,
                  float* pos9, float* nrm9, float* tex6, int32_t* material_id, float* mtl10 ) {
    tinyobj::attrib_t attrib;
    std::vector<tinyobj::shape_t> shapes;
    std::vector<tinyobj::material_t> materials;
    std::string err;
    if ( !tinyobj::LoadObj( &attrib, &shapes, &materials, &err, obj_path, mtl_dir ) ) return -1;
    int64_t f_total = 0;
    for ( size_t s = 0; s < shapes.size(); s++ ) {
        size_t index_offset = 0;
        for ( size_t f = 0; f < shapes[s].mesh.num_face_vertices.size(); f++ ) {
            int fv = shapes[s].mesh.num_face_vertices[f];
            if ( fv != 3 ) return -2; // LoadObj triangulates by default
            if ( pos9 ) {
                for ( int v = 0; v < fv; v++ ) {
                    tinyobj::index_t idx = shapes[s].mesh.indices[index_offset + v];
                    for ( int k = 0; k < 3; k++ )
                        pos9[9 * f_total + 3 * v + k] = attrib.vertices[3 * idx.vertex_index + k];
                    if ( !attrib.normals.empty() )
                        for ( int k = 0; k < 3; k++ )
                            nrm9[9 * f_total + 3 * v + k] = attrib.normals[3 * idx.normal_index + k];
                    if ( !attrib.texcoords.empty() )
                        for ( int k = 0; k < 2; k++ )
                            tex6[6 * f_total + 2 * v + k] = attrib.texcoords[2 * idx.texcoord_index + k];
                }
                material_id[f_total] = shapes[s].mesh.material_ids[f];
            }
            index_offset += fv;
            f_total++;
        }
    }
    *n_faces       = f_total;
    *n_materials   = (int64_t)materials.size();
    *has_normals   = attrib.normals.empty() ? 0 : 1;
    *has_texcoords = attrib.texcoords.empty() ? 0 : 1;
    if ( mtl10 )
        for ( size_t m = 0; m < materials.size(); m++ ) {
            float* d = mtl10 + 10 * m;
            d[0]     = materials[m].ior;
            for ( int k = 0; k < 3; k++ ) {
                d[1 + k] = materials[m].diffuse[k];
                d[4 + k] = materials[m].emission[k];
                d[7 + k] = materials[m].specular[k];
            }
        }
    return 0;
}

// Face normal and vertex normals of a loaded face, the statements of example_scenes.cpp:71-81
// evaluated with the reference's glm.
void ref_face_normals( int64_t n, const float* pos9, int has_normals, const float* nrm9, float* face3, float* out_nrm9 ) {
    for ( int64_t i = 0; i < n; i++ ) {
        vec3 v0 = v3( pos9 + 9 * i ), v1 = v3( pos9 + 9 * i + 3 ), v2 = v3( pos9 + 9 * i + 6 );
        std::vector<point3> normals;
        if ( has_normals )
            for ( int k = 0; k < 3; k++ )
                normals.push_back( v3( nrm9 + 9 * i + 3 * k ) );
        vec3 nn;
        if ( normals.empty() ) {
            vec3 v2v0 = v2 - v0;
            vec3 v1v0 = v1 - v0;
            nn        = normalize( cross( v1v0, v2v0 ) );
            normals   = { nn, nn, nn };
        }
        else { nn = normalize( normals[0] + normals[1] + normals[2] ); }
        put3( face3 + 3 * i, nn );
        for ( int k = 0; k < 3; k++ )
            put3( out_nrm9 + 9 * i + 3 * k, normals[k] );
    }
}

// ---------------------------------------------------------------- integration/flatten_ref.cpp
// The reference-side adapter run on a reference Scene + KdTree; returns the description it made
// (owned by the returned handle).
void* ref_adapter_flatten( void* scene, void* tree ) {
    auto* f = new MrtFlat;
    try {
        mrtFlattenGeometry( static_cast<Scene*>( scene ), static_cast<KdTree*>( tree ), *f );
    }
    catch ( const std::exception& ) {
        delete f;
        return nullptr;
    }
    return f;
}
const mrt_scene_desc* ref_adapter_desc( void* flat ) {
    return &static_cast<MrtFlat*>( flat )->desc;
}
void ref_adapter_free( void* flat ) {
    delete static_cast<MrtFlat*>( flat );
}
void ref_scene_set_camera( void* scene, void* cam ) {
    static_cast<Scene*>( scene )->cam = static_cast<Camera*>( cam );
}
// A checker_texture of two solid colours on material `mat` of the scene (Material::m_texture)
void ref_material_set_checker( void* scene_, int mat, const float* even3, const float* odd3, void* xf ) {
    auto* scene     = static_cast<Scene*>( scene_ );
    auto* tex       = new checker_texture( color3( even3[0], even3[1], even3[2] ), color3( odd3[0], odd3[1], odd3[2] ) );
    if ( xf ) tex->m_transform = *static_cast<Transform*>( xf );
    scene->m_Materials[mat].m_texture  = tex;
    scene->m_Materials[mat].m_MatModel = MaterialModel::COOK_TORRANCE;
}
// An image_texture of a PNG file on material `mat` (what example_scenes.cpp:1244 does)
void ref_material_set_image( void* scene_, int mat, const char* png_path, void* xf ) {
    auto* scene = static_cast<Scene*>( scene_ );
    auto* tex   = new image_texture( png_path );
    if ( xf ) tex->m_transform = *static_cast<Transform*>( xf );
    scene->m_Materials[mat].m_texture  = tex;
    scene->m_Materials[mat].m_MatModel = MaterialModel::COOK_TORRANCE;
}

// ---------------------------------------------------------------- image.cpp
// loadPng (image.cpp:14-41) of a file: width, height and the float pixels (byte / 255).
int ref_load_png( const char* path, int* w, int* h, float* rgb, int64_t capacity_floats ) {
    RenderImage* img = loadPng( path );
    if ( !img ) return -1;
    *w = (int)img->width, *h = (int)img->height;
    const int64_t n = 3 * (int64_t)img->width * img->height;
    if ( rgb && n <= capacity_floats )
        for ( size_t i = 0; i < img->width * img->height; i++ )
            put3( rgb + 3 * i, img->data[i] );
    freeImage( img );
    return 0;
}
// saveImage (image.cpp:60-146) of float pixels: writes <basename>.png
void ref_save_image( const char* basename, int w, int h, const float* rgb ) {
    RenderImage* img = initImage( w, h );
    for ( size_t i = 0; i < (size_t)w * h; i++ )
        img->data[i] = color3( rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2] );
    std::string name( basename );
    saveImage( img, name.data() );
    freeImage( img );
}

} // extern "C"
