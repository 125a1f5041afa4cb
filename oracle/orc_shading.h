// TEST INFRASTRUCTURE ONLY (CPU oracle) -- see mrt_oracle.h.
//
// Restatement of the shading side of the hot path: warps and distributions, Cook-Torrance BSDF,
// Material::eval/sample, ShapeLight / IBL, SimpleCamera, directIllumination and
// Pathtracer::trace_ray.
#pragma once
#include "orc_sampler.h"
#include "orc_scene.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <vector>

namespace orc {

// sampling.h:40-45 -- all float constants derived from the double M_PI
static const float Pi     = (float)3.14159265358979323846;
static const float InvPi  = 1.f / Pi;
static const float Inv2Pi = 1.f / ( 2.f * Pi );
static const float PiOver2 = Pi / 2.f;
static const float PiOver4 = Pi / 4.f;
static const double M_PI_D = 3.14159265358979323846;

inline bool isBlack( V3 c ) { // defines.h:48-50 (float sum compared with the double 0.)
    return c.x + c.y + c.z == 0.;
}

// ------------------------------------------------------------------ warps (sampling.cpp)
inline V2 UniformSampleTriangle( V2 u ) { // sampling.cpp:154-157
    float su0 = std::sqrt( u.x );
    return V2 { 1 - su0, u.y * su0 };
}
inline V2 ConcentricSampleDisk( V2 u ) { // sampling.cpp:113-131
    V2 uOffset { 2.f * u.x - 1, 2.f * u.y - 1 };
    if ( uOffset.x == 0 && uOffset.y == 0 ) return V2 { 0, 0 };
    float theta, r;
    if ( std::abs( uOffset.x ) > std::abs( uOffset.y ) ) {
        r     = uOffset.x;
        theta = PiOver4 * ( uOffset.y / uOffset.x );
    }
    else {
        r     = uOffset.y;
        theta = PiOver2 - PiOver4 * ( uOffset.x / uOffset.y );
    }
    return V2 { r * std::cos( theta ), r * std::sin( theta ) };
}
inline V3 UniformSampleSphere( V2 u ) { // sampling.cpp:95-100
    float z   = 1 - 2 * u.x;
    float r   = std::sqrt( std::max( (float)0, (float)1 - z * z ) );
    float phi = 2 * Pi * u.y;
    return v3( r * std::cos( phi ), r * std::sin( phi ), z );
}
inline float UniformConePdf( float cosThetaMax ) { // sampling.cpp:133-135
    return 1 / ( 2 * Pi * ( 1 - cosThetaMax ) );
}
inline float PowerHeuristic( int nf, float fPdf, int ng, float gPdf ) { // sampling.h:181-184
    float f = nf * fPdf, g = ng * gPdf;
    return ( f * f ) / ( f * f + g * g );
}
inline void CoordinateSystem( V3 v1, V3* v2, V3* v3_ ) { // defines.h:150-156
    if ( std::abs( v1.x ) > std::abs( v1.y ) )
        *v2 = v3( -v1.z, 0, v1.x ) / std::sqrt( v1.x * v1.x + v1.z * v1.z );
    else
        *v2 = v3( 0, v1.z, -v1.y ) / std::sqrt( v1.y * v1.y + v1.z * v1.z );
    *v3_ = cross( v1, *v2 );
}
inline V3 SphericalDirection( float sinTheta, float cosTheta, float phi, V3 x, V3 y, V3 z ) { // defines.h:158-165
    return sinTheta * std::cos( phi ) * x + sinTheta * std::sin( phi ) * y + cosTheta * z;
}

template <typename Predicate>
int FindInterval( int size, const Predicate& pred ) { // sampling.h:54-67
    int first = 0, len = size;
    while ( len > 0 ) {
        int half = len >> 1, middle = first + half;
        if ( pred( middle ) ) {
            first = middle + 1;
            len -= half + 1;
        }
        else
            len = half;
    }
    int r = first - 1;
    return r < 0 ? 0 : ( r > size - 2 ? size - 2 : r );
}

struct Distribution1D { // sampling.h:69-119
    std::vector<float> func, cdf;
    float funcInt;
    Distribution1D( const float* f, int n ) : func( f, f + n ), cdf( n + 1 ) {
        cdf[0] = 0;
        for ( int i = 1; i < n + 1; ++i )
            cdf[i] = cdf[i - 1] + func[i - 1] / n;
        funcInt = cdf[n];
        if ( funcInt == 0 ) {
            for ( int i = 1; i < n + 1; ++i )
                cdf[i] = float( i ) / float( n );
        }
        else {
            for ( int i = 1; i < n + 1; ++i )
                cdf[i] /= funcInt;
        }
    }
    int Count() const { return (int)func.size(); }
    int SampleDiscrete( float u, float* pdf ) const {
        int offset = FindInterval( (int)cdf.size(), [&]( int index ) { return cdf[index] <= u; } );
        if ( pdf ) *pdf = ( funcInt > 0 ) ? func[offset] / ( funcInt * Count() ) : 0;
        return offset;
    }
};

struct Distribution2D { // sampling.h:133-157, sampling.cpp:159-171
    std::vector<std::unique_ptr<Distribution1D>> pConditionalV;
    std::unique_ptr<Distribution1D> pMarginal;
    Distribution2D( const float* func, int nu, int nv ) {
        for ( int v = 0; v < nv; ++v )
            pConditionalV.emplace_back( new Distribution1D( &func[(size_t)v * nu], nu ) );
        std::vector<float> marginalFunc;
        for ( int v = 0; v < nv; ++v )
            marginalFunc.push_back( pConditionalV[v]->funcInt );
        pMarginal.reset( new Distribution1D( &marginalFunc[0], nv ) );
    }
    float Pdf( V2 p ) const {
        int cu = pConditionalV[0]->Count(), cv = pMarginal->Count();
        int iu = std::min( std::max( int( p.x * cu ), 0 ), cu - 1 );
        int iv = std::min( std::max( int( p.y * cv ), 0 ), cv - 1 );
        return pConditionalV[iv]->func[iu] / pMarginal->funcInt;
    }
};

// ------------------------------------------------------------------ bsdf.cpp
inline float RDM_chiplus( float c ) { // bsdf.cpp:19-21
    return ( c > 0.f ) ? 1.f : 0.f;
}
inline float RDM_Beckmann( float NdotH, float alpha ) { // bsdf.cpp:43-58
    float phi          = RDM_chiplus( NdotH );
    float cos2         = NdotH * NdotH;
    float tan2         = ( 1 - ( cos2 ) ) / cos2;
    float alpha2       = alpha * alpha;
    float e            = std::exp( -tan2 / ( alpha2 ) );
    float pi_alpha_cos = (float)( M_PI_D * alpha2 * ( cos2 * cos2 ) ); // M_PI is a double
    float d            = phi * ( e / pi_alpha_cos );
    return d;
}
inline float RDM_Fresnel( float LdotH, float extIOR, float intIOR ) { // bsdf.cpp:100-118
    LdotH         = std::clamp( LdotH, -1.f, 1.f );
    bool entering = LdotH > 0.f;
    if ( !entering ) {
        std::swap( extIOR, intIOR );
        LdotH = std::abs( LdotH );
    }
    float sinThetaI = std::sqrt( std::max( 0.f, 1.f - LdotH * LdotH ) );
    float sinThetaT = extIOR / intIOR * sinThetaI;
    if ( sinThetaT >= 1 ) return 1.f;
    float cosThetaT = std::sqrt( std::max( 0.f, 1.f - sinThetaT * sinThetaT ) );
    float Rparl = ( ( intIOR * LdotH ) - ( extIOR * cosThetaT ) ) / ( ( intIOR * LdotH ) + ( extIOR * cosThetaT ) );
    float Rperp = ( ( extIOR * LdotH ) - ( intIOR * cosThetaT ) ) / ( ( extIOR * LdotH ) + ( intIOR * cosThetaT ) );
    return ( Rparl * Rparl + Rperp * Rperp ) / 2.f;
}
inline float RDM_G1( float DdotH, float DdotN, float alpha ) { // bsdf.cpp:141-154
    float tanx = std::sqrt( 1.f - fmin_glm( 0.99999f, DdotN * DdotN ) );
    float b    = DdotN / ( fmax_glm( 0.0001f, alpha ) * tanx );
    if ( b < 1.6f ) { return ( ( 3.535f + 2.181f * b ) * b ) / ( 1.f + ( 2.276f + 2.577f * b ) * b ); }
    else { return 1.f; }
}
inline float RDM_Smith( float LdotH, float LdotN, float VdotH, float VdotN, float alpha ) { // bsdf.cpp:180-182
    return RDM_G1( LdotH, LdotN, alpha ) * RDM_G1( VdotH, VdotN, alpha );
}

struct BrdfData { // Material.h:16-31
    float LdotH, NdotH, VdotH, LdotN, VdotN;
    V3 F, specularF0, diffuseReflectance;
    float roughness, alpha, alphaSq, metalness;
};

inline V3 RDM_bsdf_s( const BrdfData& data ) { // bsdf.cpp:190-200
    float d = RDM_Beckmann( data.NdotH, data.alpha );
    float g = RDM_Smith( data.LdotH, data.LdotN, data.VdotH, data.VdotN, data.alpha );
    return ( d * data.F * g ) / ( 4.f * data.LdotN * data.VdotN );
}
inline V3 RDM_bsdf( const BrdfData& data ) { // bsdf.cpp:225-259 with COMBINE_BRDFS_WITH_FRESNEL 1
    float pi    = (float)M_PI_D;
    V3 diffuse  = data.diffuseReflectance / pi;
    V3 specular = RDM_bsdf_s( data );
    return ( v3( 1.f ) - data.F ) * diffuse + specular;
}

// ------------------------------------------------------------------ Material.h:40-76
inline float luminance( V3 rgb ) {
    return dot( rgb, v3( 0.2126f, 0.7152f, 0.0722f ) );
}
inline float shadowedF90( V3 F0 ) {
    const float t = ( 1.0f / 0.04f );
    return fmin_glm( 1.0f, t * luminance( F0 ) );
}
inline V3 evalFresnelSchlick( V3 f0, float f90, float NdotS ) {
    // f0 + (f90 - f0) * pow(1.0f - NdotS, 5.0f): (scalar - vec3) * scalar
    V3 d    = v3( f90 - f0.x, f90 - f0.y, f90 - f0.z );
    float p = std::pow( 1.0f - NdotS, 5.0f );
    return f0 + d * p;
}
inline V3 baseColorToSpecularF0( V3 baseColor, float metalness ) {
    return mix( v3( 0.04f ), baseColor, metalness );
}
inline V3 baseColorToDiffuseReflectance( V3 baseColor, float metalness ) {
    return baseColor * ( 1.0f - metalness );
}

inline bool computeBrdfData( BrdfData& data, V3 v, V3 wi, V3 n, V3 baseColor, float metalness, float roughness ) {
    // Material.cpp:1277-1316
    V3 h        = normalize( v + wi );
    float LdotN = dot( wi, n );
    float VdotN = dot( v, n );
    if ( LdotN <= 0.0f || VdotN <= 0.0f ) return false;
    data.LdotN              = fmin_glm( fmax_glm( 0.0001f, LdotN ), 1.0f );
    data.VdotN              = fmin_glm( fmax_glm( 0.0001f, VdotN ), 1.0f );
    data.LdotH              = clampf( dot( wi, h ), 0.0f, 1.0f );
    data.NdotH              = clampf( dot( n, h ), 0.0f, 1.0f );
    data.VdotH              = clampf( dot( v, h ), 0.0f, 1.0f );
    data.specularF0         = baseColorToSpecularF0( baseColor, metalness );
    data.diffuseReflectance = baseColorToDiffuseReflectance( baseColor, metalness );
    data.roughness          = roughness;
    data.alpha              = data.roughness;
    data.alphaSq            = data.alpha * data.alpha;
    data.F                  = evalFresnelSchlick( data.specularF0, shadowedF90( data.specularF0 ), data.LdotH );
    data.metalness          = metalness;
    return true;
}

inline float Beckmann_D( float alphaSquared, float NdotH ) { // Material.cpp:220-225
    float cos2Theta   = NdotH * NdotH;
    float numerator   = std::exp( ( cos2Theta - 1.0f ) / ( alphaSquared * cos2Theta ) );
    float denominator = Pi * alphaSquared * cos2Theta * cos2Theta;
    return numerator / denominator;
}
inline float sampleBeckmannWalterReflectionPdf( float alphaSquared, float NdotH, float LdotH ) { // Material.cpp:231-239
    NdotH = fmax_glm( 0.00001f, NdotH );
    LdotH = fmax_glm( 0.00001f, LdotH );
    return Beckmann_D( fmax_glm( 0.00001f, alphaSquared ), NdotH ) * NdotH / ( 4.0f * LdotH );
}
inline V3 sampleBeckmannWalter( float alpha, V2 u ) { // Material.cpp:55-68
    // alpha = dot(vec2(alpha, alpha), vec2(0.5, 0.5))
    alpha                 = dot( V2 { alpha, alpha }, V2 { 0.5f, 0.5f } );
    float tanThetaSquared = -( alpha * alpha ) * std::log( 1.0f - u.x );
    float phi             = ( 2.f * Pi ) * u.y;
    float cosTheta        = inversesqrt( 1.0f + tanThetaSquared );
    float sinTheta        = std::sqrt( 1.0f - cosTheta * cosTheta );
    return normalize( v3( sinTheta * std::cos( phi ), sinTheta * std::sin( phi ), cosTheta ) );
}
inline V4 getRotationToZAxis( V3 input ) { // Material.cpp:72-78
    if ( input.z < -0.99999f ) return V4 { 1.0f, 0.0f, 0.0f, 0.0f };
    return normalize( V4 { input.y, -input.x, 0.0f, 1.0f + input.z } );
}
inline V4 invertRotation( V4 q ) { // Material.cpp:110-112
    return V4 { -q.x, -q.y, -q.z, q.w };
}
inline V3 rotatePoint( V4 q, V3 v ) { // Material.cpp:116-120
    const V3 qAxis = v3( q.x, q.y, q.z );
    return 2.0f * dot( qAxis, v ) * qAxis + ( q.w * q.w - dot( qAxis, qAxis ) ) * v + 2.0f * q.w * cross( qAxis, v );
}

// ------------------------------------------------------------------ textures.hpp
struct Texture {
    int kind;
    uint32_t w, h;
    const float* texels;
    V3 even, odd;
    Xf xf;
};
inline int texel_index( float f, uint32_t n ) {
    if ( !( f >= 0.f ) ) return 0;
    if ( f >= (float)( n - 1 ) ) return (int)n - 1;
    return (int)f;
}
inline V3 texture_value( const Texture& t, float u, float v ) {
    // texture::value(u, v, duv) (textures.hpp:21-43): only the transform, then a point lookup
    V3 uvw = t.xf.to( v3( u, v, 0 ) );
    u      = uvw.x;
    v      = uvw.y;
    float ux = u - (int)u, vy = v - (int)v;
    if ( ux < 0.0 ) ux += 1.0;
    if ( vy < 0.0 ) vy += 1.0;
    if ( t.kind == MRT_TEX_CHECKER ) { // textures.hpp:76-96
        if ( ux <= 0.5 ) return ( vy <= 0.5 ) ? t.even : t.odd;
        return ( vy <= 0.5 ) ? t.odd : t.even;
    }
    // textures.hpp:110-119.  ux, vy are in [0, 1] for every finite coordinate below 2^31; a NaN
    // coordinate (a NaN ray "hits" a plane: every comparison of Plane::intersect is false) or one
    // beyond the int range makes the reference's float -> int conversions undefined (on x86 it then
    // reads far outside the image).  texel_index is that conversion with those cases defined, the
    // same way in the kernels: not >= 0 -> 0, beyond the last texel -> the last texel.
    int u2 = texel_index( ux * ( t.w - 1 ), t.w );
    int v2 = texel_index( vy * ( t.h - 1 ), t.h );
    size_t px = (size_t)( t.h - v2 - 1 ) * t.w + u2;
    return v3( t.texels + 3 * px );
}

// UniformLightDistribution::UniformLightDistribution (LightDistribution.cpp:5-8)
inline Distribution1D* uniform_light_distribution( int n_lights ) {
    std::vector<float> prob( n_lights, 1.f );
    return new Distribution1D( &prob[0], int( prob.size() ) );
}

// ------------------------------------------------------------------ shading scene
struct ShadeScene : TraceScene {
    std::vector<Shape> light_shapes;
    std::vector<Texture> textures;
    std::unique_ptr<Distribution1D> lightDistrib; // UniformLightDistribution (LightDistribution.cpp:5-8)
    std::unique_ptr<Distribution2D> envDistrib;   // IBL::m_distribution (Light.h:221)
    M3 env_inv;
    explicit ShadeScene( const mrt_scene_desc* desc ) : TraceScene( desc ) {
        for ( uint32_t i = 0; i < desc->n_lights; i++ )
            light_shapes.push_back( light_shape( desc, i ) );
        for ( uint32_t i = 0; i < desc->n_textures; i++ ) {
            const mrt_texture& t = desc->textures[i];
            textures.push_back( Texture { t.kind, t.width, t.height, desc->texels ? desc->texels + t.texel_offset : nullptr,
                                          v3( t.even ), v3( t.odd ), xf_from( t.xf ) } );
        }
        if ( desc->n_lights > 0 ) lightDistrib.reset( uniform_light_distribution( (int)desc->n_lights ) );
        if ( desc->env_rgb && desc->env_width > 0 )
            envDistrib.reset( new Distribution2D( desc->env_rgb, desc->env_width, desc->env_height ) );
        env_inv = m3_from( desc->env_inv );
    }
};

struct Counters {
    mrt_trace_stats cls[3] {}; // traversal counters by ray class: 0 continuation, 1 shadow, 2 MIS
    uint64_t rays[3] = { 0, 0, 0 };
};

// Sampler interface used by the integrator; both samplers provide it.
struct SamplerIface {
    virtual float Get1D()      = 0;
    virtual V2 Get2D()         = 0;
    virtual void Skip2D()      = 0;
    virtual float engine01()   = 0;
    virtual ~SamplerIface() {}
};
struct CounterIface : SamplerIface {
    CounterSampler s;
    CounterIface( uint32_t seed, uint32_t pixel, uint32_t sx, uint32_t sy, uint32_t sample ) : s( seed, pixel, sx, sy, sample ) {}
    float Get1D() override { return s.Get1D(); }
    V2 Get2D() override { return s.Get2D(); }
    void Skip2D() override { s.Skip2D(); }
    float engine01() override { return s.engine01(); }
};
struct SequentialIface : SamplerIface {
    StratifiedSampler* s;
    RNG* engine;
    SequentialIface( StratifiedSampler* s_, RNG* e ) : s( s_ ), engine( e ) {}
    float Get1D() override { return s->Get1D(); }
    V2 Get2D() override { return s->Get2D(); }
    void Skip2D() override { s->Get2D(); }
    float engine01() override { return engine->Uniformfloat(); }
};

// random_dir (sampling.h:199-211)
inline V3 random_dir( V3 normal, SamplerIface& smp ) {
    float r1 = smp.engine01();
    float r2 = smp.engine01();
    float theta = 2.f * Pi * r1;
    float r     = std::sqrt( 1.f - r2 );
    V3 local    = v3( std::cos( theta ) * r, std::sin( theta ) * r, std::sqrt( r2 ) );
    // uniform01(engine) - 0.5: float minus double, narrowed to float by the vec3 constructor.
    // g++ evaluates the three constructor arguments last to first (recorded from the reference's
    // own random_dir: the third engine draw feeds z, the fifth x).
    float rz = (float)( (double)smp.engine01() - 0.5 );
    float ry = (float)( (double)smp.engine01() - 0.5 );
    float rx = (float)( (double)smp.engine01() - 0.5 );
    V3 randV = v3( rx, ry, rz );
    V3 tangent1 = normalize( cross( normal, randV ) );
    V3 tangent2 = cross( tangent1, normal );
    return local.z * normal + local.x * tangent1 + local.y * tangent2;
}

// ------------------------------------------------------------------ Material::eval / sample
inline V3 material_albedo( const ShadeScene& sc, const mrt_material& m, const Isect& isect, bool hasDifferentials ) {
    if ( m.texture >= 0 && hasDifferentials ) return texture_value( sc.textures[m.texture], isect.u, isect.v );
    return v3( m.albedo );
}

inline V3 material_eval( const ShadeScene& sc, const mrt_material& m, const Ray& ray, const Isect& isect, V3 wi, float* scatteringPdf ) {
    // Material.cpp:983-1013
    if ( m.type == MRT_MAT_TRANSPARENT ) return v3( 0.f );
    BrdfData data;
    V3 bsdf   = v3( 0.f );
    V3 albedo = material_albedo( sc, m, isect, ray.hasDifferentials );
    if ( computeBrdfData( data, -ray.dir, wi, isect.normal, albedo, m.metalness, m.roughness ) ) {
        bsdf           = RDM_bsdf( data );
        *scatteringPdf = sampleBeckmannWalterReflectionPdf( data.alphaSq, data.NdotH, data.LdotH );
    }
    return bsdf;
}

inline V3 material_sample( const ShadeScene& sc, const mrt_material& m, const Ray& ray, const Isect& isect, SamplerIface& smp, V3* wi, float* scatteringPdf ) {
    // Material.cpp:1015-1100
    if ( m.type == MRT_MAT_TRANSPARENT ) {
        V3 normal   = isect.isOutside ? isect.normal : -isect.normal;
        float LdotH = dot( -ray.dir, normal );
        float f     = isect.isOutside ? RDM_Fresnel( LdotH, 1.f, m.ior ) : RDM_Fresnel( LdotH, m.ior, 1.f );
        if ( f < 1.0f ) {
            if ( smp.engine01() < f ) { *wi = reflect( ray.dir, normal ); }
            else {
                float refractionRatio = isect.isOutside ? ( 1.f / m.ior ) : m.ior;
                *wi                   = refract( ray.dir, normal, refractionRatio );
            }
        }
        else { *wi = reflect( ray.dir, normal ); }
        *scatteringPdf = 1.f;
        return v3( 1.f );
    }
    V3 V = -ray.dir;
    float ux = smp.engine01();
    float uy = smp.engine01();
    V2 u { ux, uy };
    V3 bsdf = v3( 0.f );
    if ( !isect.isOutside ) return v3( 0.f );

    V4 qRotationToZ = getRotationToZAxis( isect.normal );
    V3 Vlocal       = rotatePoint( qRotationToZ, V );
    float alpha     = m.roughness;
    V3 Hlocal;
    if ( alpha == 0.0f ) { Hlocal = v3( 0.0f, 0.0f, 1.0f ); }
    else { Hlocal = sampleBeckmannWalter( alpha, u ); }
    V3 Llocal = reflect( -Vlocal, Hlocal );
    *wi       = normalize( rotatePoint( invertRotation( qRotationToZ ), Llocal ) );

    BrdfData data;
    V3 albedo = material_albedo( sc, m, isect, ray.hasDifferentials );
    if ( computeBrdfData( data, -ray.dir, *wi, isect.normal, albedo, m.metalness, m.roughness ) ) {
        bsdf           = RDM_bsdf( data );
        *scatteringPdf = sampleBeckmannWalterReflectionPdf( data.alphaSq, data.NdotH, data.LdotH );
    }
    return bsdf;
}

// ------------------------------------------------------------------ shape sampling
inline float triangle_area( const Shape& s ) { // triangle.cpp:99-104 (0.5 is a double)
    return (float)( 0.5 * (double)length( cross( s.p2 - s.p1, s.p3 - s.p1 ) ) );
}
inline Isect triangle_sample( const Shape& s, V2 u, float* pdf ) { // triangle.cpp:86-97
    V2 b = UniformSampleTriangle( u );
    Isect it;
    it.position = b.x * s.p1 + b.y * s.p2 + ( 1.f - b.x - b.y ) * s.p3;
    it.normal   = normalize( s.n2 * b.x + s.n3 * b.y + s.n1 * ( 1.f - b.x - b.y ) );
    *pdf        = 1 / triangle_area( s );
    return it;
}
inline Isect sphere_sample_uniform( const Shape& s, V2 u ) { // sphere.cpp:182-189
    V3 pObj = v3( 0.f ) + s.radius * UniformSampleSphere( u );
    Isect it;
    it.normal   = normalize( mul( s.xf.m, pObj ) );
    pObj        = pObj * ( s.radius / length( pObj - v3( 0.f ) ) );
    it.position = s.xf.from( pObj );
    return it;
}
inline Isect sphere_sample( const Shape& s, const Isect& inter, V2 u, float* pdf ) { // sphere.cpp:191-223
    V3 pCenter = s.center;
    V3 wc      = normalize( pCenter - inter.position );
    V3 wcX, wcY;
    CoordinateSystem( wc, &wcX, &wcY );

    V3 normal     = inter.isOutside ? inter.normal : -inter.normal;
    V3 pOrigin    = inter.position + ( acne_eps * normal );
    float dist    = length( pOrigin - pCenter );
    float radius2 = s.radius * s.radius;
    if ( ( dist * dist ) <= radius2 ) return sphere_sample_uniform( s, u ); // *pdf left untouched

    dist               = length( inter.position - pCenter );
    float sinThetaMax2 = radius2 / ( dist * dist );
    float cosThetaMax  = std::sqrt( fmax_glm( 0.f, 1.f - sinThetaMax2 ) );
    float cosTheta     = ( 1.f - u.x ) + u.x * cosThetaMax;
    float sinTheta     = std::sqrt( fmax_glm( 0.f, 1.f - cosTheta * cosTheta ) );
    float phi          = u.y * 2.f * Pi;

    float ds       = dist * cosTheta - std::sqrt( fmax_glm( 0.f, radius2 - dist * dist * sinTheta * sinTheta ) );
    float cosAlpha = ( dist * dist + radius2 - ds * ds ) / ( 2.f * dist * s.radius );
    float sinAlpha = std::sqrt( fmax_glm( 0.f, 1.f - cosAlpha * cosAlpha ) );

    V3 nObj = SphericalDirection( sinAlpha, cosAlpha, phi, -wcX, -wcY, -wc );
    V3 pObj = s.radius * v3( nObj.x, nObj.y, nObj.z );

    Isect it;
    pObj        = pObj * ( s.radius / length( pObj - v3( 0.f ) ) );
    it.position = s.xf.from( pObj );
    it.normal   = mul( s.xf.m, nObj );
    *pdf        = 1 / ( 2 * Pi * ( 1 - cosThetaMax ) );
    return it;
}
inline Isect shape_sample( const Shape& s, const Isect& inter, V2 u, float* pdf ) {
    if ( s.kind == MRT_SPHERE ) return sphere_sample( s, inter, u, pdf );
    return triangle_sample( s, u, pdf );
}
inline float shape_area( const Shape& s ) { // Object.h:19: the base class answers 1
    return s.kind == MRT_TRIANGLE ? triangle_area( s ) : 1.f;
}
inline float shape_pdf( const Shape& s, const Isect& inter, V3 wi ) {
    if ( s.kind == MRT_SPHERE ) { // sphere.cpp:225-236
        V3 pCenter = s.xf.from( v3( 0.f ) );
        V3 pOrigin = inter.position + ( acne_eps * inter.normal );
        V3 dc      = pOrigin - pCenter;
        if ( dot( dc, dc ) <= s.radius * s.radius ) return 1.f;
        V3 dp              = inter.position - pCenter;
        float sinThetaMax2 = s.radius * s.radius / dot( dp, dp );
        float cosThetaMax  = std::sqrt( fmax_glm( 0.f, 1.f - sinThetaMax2 ) );
        return UniformConePdf( cosThetaMax );
    }
    // Object::pdf (Object.cpp:13-22)
    Ray ray( inter.position + wi * acne_eps, wi, 0, 10000 );
    Isect isectLight;
    if ( !shape_intersect( s, &ray, &isectLight ) ) return 0;
    V3 dp     = isectLight.position - inter.position;
    float pdf = dot( dp, dp ) / ( std::abs( dot( isectLight.normal, -wi ) * shape_area( s ) ) );
    if ( std::isinf( pdf ) ) pdf = 0.f;
    return pdf;
}

// ------------------------------------------------------------------ lights (Light.cpp:247-342)
// The column / row of a direction: the reference converts phi / 2pi * width and theta / pi * height
// to int (Light.cpp:309-310, 336-337).  theta is NaN when rounding leaves |dir.y| a unit in the last
// place above 1 (acos), and the conversion of a NaN is undefined (INT_MIN on x86, and the index
// computed from it overflows); defined here, and the same way in the kernels, as 0 -- the row of the
// pole such a direction points at.
inline int env_coord( double x ) {
    return x == x ? (int)x : 0;
}
inline V3 env_texel( const ShadeScene& sc, int i, int j ) {
    const mrt_scene_desc* d = sc.d;
    long index              = 3L * i + 3L * (long)d->env_width * j;
    long last               = 3L * ( (long)d->env_width * d->env_height - 1 );
    if ( index > last ) index = last; // the reference reads past the image here (theta == pi)
    if ( index < 0 ) index = 0;
    return v3( d->env_rgb + index );
}
inline V3 ibl_Le( const ShadeScene& sc, V3 raydir ) { // Light.cpp:331-342
    V3 dir       = mul( sc.env_inv, raydir );
    double theta = std::acos( dir.y );
    double phi   = std::atan2( dir.z, dir.x );
    if ( phi < 0 ) phi += 2 * M_PI_D;
    int i = env_coord( phi / ( 2 * M_PI_D ) * sc.d->env_width );
    int j = env_coord( theta / M_PI_D * sc.d->env_height );
    return env_texel( sc, i, j );
}
inline float ibl_pdf_Li( const ShadeScene& sc, V3 wi ) { // Light.cpp:317-329
    V3 dir      = mul( sc.env_inv, wi );
    float theta = std::acos( dir.y );
    float phi   = std::atan2( dir.z, dir.x );
    if ( phi < 0 ) phi += 2 * M_PI_D;
    float sinTheta = std::sin( theta );
    if ( sinTheta == 0 ) return 0;
    return sc.envDistrib->Pdf( V2 { phi * Inv2Pi, theta * InvPi } ) / ( 2 * Pi * Pi * sinTheta );
}
inline V3 shapelight_L( V3 color, V3 n, V3 w ) { // Light.cpp:247-249
    return dot( n, w ) > 0.f ? color : v3( 0.f );
}

bool trace( const ShadeScene& sc, Ray* ray, Isect* isect, Counters& c, int cls );

inline V3 light_sample_Li( const ShadeScene& sc, uint32_t li, const Isect& inter, V2 u, SamplerIface& smp, V3* wi, float* pdf, bool* occluded, Counters& c ) {
    const mrt_light& l = sc.d->lights[li];
    if ( l.kind == MRT_LIGHT_SHAPE ) { // Light.cpp:251-271
        Isect pShape = shape_sample( sc.light_shapes[li], inter, u, pdf );
        *wi          = normalize( pShape.position - inter.position );
        V3 origin    = inter.position + ( acne_eps * inter.normal );
        Ray rayS( origin, *wi, 0.f, distance( pShape.position, inter.position ) * 0.99f );
        rayS.hasDifferentials = false;
        rayS.shadow           = true;
        Isect temp;
        *occluded = trace( sc, &rayS, &temp, c, 1 );
        return shapelight_L( v3( l.color ), pShape.normal, -*wi );
    }
    // IBL::sample_Li (Light.cpp:277-315)
    V3 normal = inter.isOutside ? inter.normal : -inter.normal;
    *wi       = normalize( random_dir( normal, smp ) );
    *pdf      = fmax_glm( dot( normal, *wi ), 0.f ) / Pi;
    V3 origin = inter.position + *wi * acne_eps;
    Ray rayS( origin, *wi, 0.f, 10000 );
    rayS.hasDifferentials = false;
    rayS.shadow           = true;
    Isect temp;
    *occluded = trace( sc, &rayS, &temp, c, 1 );

    V3 dir      = mul( sc.env_inv, rayS.dir );
    float theta = std::acos( dir.y );
    float phi   = std::atan2( dir.z, dir.x );
    if ( phi < 0 ) phi += 2 * M_PI_D;
    int i = env_coord( phi / ( 2 * M_PI_D ) * sc.d->env_width );
    int j = env_coord( theta / M_PI_D * sc.d->env_height );
    return env_texel( sc, i, j );
}
inline float light_pdf_Li( const ShadeScene& sc, uint32_t li, const Isect& it, V3 wi ) {
    if ( sc.d->lights[li].kind == MRT_LIGHT_SHAPE ) return shape_pdf( sc.light_shapes[li], it, wi ); // Light.cpp:273-275
    return ibl_pdf_Li( sc, wi );
}
inline V3 light_Le( const ShadeScene& sc, uint32_t li, V3 raydir ) { // Light.h:31 / Light.cpp:331
    if ( sc.d->lights[li].kind == MRT_LIGHT_IBL ) return ibl_Le( sc, raydir );
    return v3( 0.f );
}

inline bool trace( const ShadeScene& sc, Ray* ray, Isect* isect, Counters& c, int cls ) {
    c.rays[cls]++;
    return intersect_kdtree( sc, ray, isect, &c.cls[cls] );
}

// directIllumination (integrator.cpp:69-120)
inline V3 directIllumination( const ShadeScene& sc, const Ray& ray, const Isect& isect, uint32_t li, V2 uLight, SamplerIface& smp, Counters& c ) {
    V3 ret = v3( 0.f );
    V3 wi { 0, 0, 0 };
    float lightPdf = 0, scatteringPdf = 0;
    bool vis = false;
    V3 Li    = light_sample_Li( sc, li, isect, uLight, smp, &wi, &lightPdf, &vis, c );
    const mrt_material& mat = sc.d->materials[isect.materialIndex];
    if ( lightPdf > 0 && !isBlack( Li ) ) {
        V3 normal = isect.isOutside ? isect.normal : -isect.normal;
        V3 bsdf   = material_eval( sc, mat, ray, isect, wi, &scatteringPdf ) * fmax_glm( dot( normal, wi ), 0.f );
        if ( !isBlack( bsdf ) ) {
            if ( vis ) { Li = v3( 0.f ); }
            if ( !isBlack( Li ) ) {
                float weight = PowerHeuristic( 1, lightPdf, 1, scatteringPdf );
                ret          = ret + bsdf * Li * weight / lightPdf;
            }
        }
    }

    // integrator.cpp:97-99 multiplies Material::sample(..., &wi, ...) by max(dot(normal, wi), 0) in one
    // expression.  The two operands of `*` are unsequenced; g++ (the compiler of the reference's
    // Linux build, and of this container) evaluates the right operand first, so the cosine is taken
    // with the wi left behind by the LIGHT sample, before Material::sample overwrites it.  Stated
    // explicitly here so that the result does not depend on the compiler.
    float cosPrevWi = fmax_glm( dot( isect.normal, wi ), 0.f );
    V3 bsdf         = material_sample( sc, mat, ray, isect, smp, &wi, &scatteringPdf ) * cosPrevWi;
    if ( !isBlack( bsdf ) && scatteringPdf > 0 ) {
        float weight = 1;
        lightPdf     = light_pdf_Li( sc, li, isect, wi );
        if ( lightPdf == 0 ) return ret;
        weight = PowerHeuristic( 1, scatteringPdf, 1, lightPdf );
        Ray ray_ref( isect.position + ( wi * acne_eps ), normalize( wi ), 0, 10000 );
        ray_ref.hasDifferentials = false;
        Isect temp;
        bool found = trace( sc, &ray_ref, &temp, c, 2 );
        V3 Li2     = v3( 0.f );
        if ( found ) {
            if ( sc.mat_emissive_sum[temp.materialIndex] ) Li2 = shapelight_L( v3( sc.d->lights[li].color ), temp.normal, -wi );
        }
        else
            Li2 = light_Le( sc, li, ray_ref.dir );
        if ( !isBlack( Li2 ) ) ret = ret + bsdf * Li2 * weight / scatteringPdf;
    }
    return ret;
}

// Pathtracer::trace_ray (integrator.cpp:128-211)
inline V3 trace_ray( const ShadeScene& sc, Ray ray, int maxDepth, SamplerIface& smp, Counters& c ) {
    V3 ret        = v3( 0.f );
    V3 throughput = v3( 1.f );
    Isect isect;
    for ( int depth = 0;; depth++ ) {
        if ( trace( sc, &ray, &isect, c, 0 ) ) {
            const mrt_material& mat = sc.d->materials[isect.materialIndex];
            if ( !isBlack( v3( mat.emission ) ) && depth == 0 ) {
                V3 col0 = sc.d->n_lights > 0 ? v3( sc.d->lights[0].color ) : v3( 0.f );
                ret     = ret + throughput * shapelight_L( col0, isect.normal, -ray.dir );
            }
            int nLights = (int)sc.d->n_lights;
            if ( nLights > 0 ) {
                float lightPdf;
                int lightNum = sc.lightDistrib->SampleDiscrete( smp.Get1D(), &lightPdf );
                if ( lightPdf > 0 ) {
                    V2 uLight = smp.Get2D();
                    smp.Skip2D(); // uScattering: drawn by the reference, never read
                    ret = ret + throughput * directIllumination( sc, ray, isect, lightNum, uLight, smp, c ) / lightPdf;
                }
            }
            float pdf = 0;
            V3 wi { 0, 0, 0 };
            smp.Skip2D(); // the Get2D() passed to Material::sample, which ignores it
            V3 bsdf = material_sample( sc, mat, ray, isect, smp, &wi, &pdf );
            wi      = normalize( wi );
            if ( isBlack( bsdf ) || pdf == 0.f ) break;
            if ( mat.type != MRT_MAT_TRANSPARENT ) throughput = throughput * ( bsdf * std::abs( dot( wi, isect.normal ) ) / pdf );
            ray = Ray( isect.position + acne_eps * wi, wi, 0, 10000 );
        }
        else {
            if ( sc.d->env_light >= 0 ) ret = ret + throughput * ibl_Le( sc, ray.dir );
            break;
        }
        if ( depth >= maxDepth ) break;
    }
    return ret;
}

// SimpleCamera (Camera.cpp:5-35)
inline void make_camera( V3 lookfrom, V3 lookat, V3 vup, float vfov, float width, float height, float lensRadius, float focalDistance, mrt_camera* out ) {
    // setSimpleCamera (scene.cpp:151-161) passes aspect = w/h and int(width), int(height)
    V3 pos       = lookfrom;
    int imgWidth = (int)width, imgHeight = (int)height;
    V3 dir       = lookat;
    dir          = dir - pos;
    dir          = normalize( dir );
    V3 xDir      = cross( dir, vup );
    V3 up        = normalize( cross( xDir, dir ) );
    V3 left      = normalize( cross( up, dir ) );
    // fov * (float)M_PI / 180.0 : float*float, then divided by a double, narrowed to float
    float radfov = (float)( (double)( vfov * (float)M_PI_D ) / 180.0 );
    float l      = focalDistance;
    float h      = std::tan( radfov * .5f ) * ( 2.0f * l );
    // Camera::imgWidth / imgHeight are floats (Camera.h:19-20)
    float fw = (float)imgWidth, fh = (float)imgHeight;
    float w  = h * fw / fh;
    V3 B     = pos + ( l * dir ) + ( h / 2.0f * up );
    V3 nearPlaneTopLeft = B + w / 2.0f * ( left );
    V3 dXPixel          = -left * ( w / fw );
    V3 dYPixel          = -up * ( w / fw );
    std::memcpy( out->pos, &pos, 12 );
    std::memcpy( out->near_top_left, &nearPlaneTopLeft, 12 );
    std::memcpy( out->dx_pixel, &dXPixel, 12 );
    std::memcpy( out->dy_pixel, &dYPixel, 12 );
    std::memcpy( out->x_dir, &xDir, 12 );
    std::memcpy( out->up, &up, 12 );
    out->lens_radius = lensRadius;
}
// SimpleCamera::get_ray (Camera.cpp:37-67), differentials omitted
inline Ray camera_ray( const mrt_camera& cam, float s, float t, float uLens, float vLens ) {
    V3 pos = v3( cam.pos ), ntl = v3( cam.near_top_left ), dX = v3( cam.dx_pixel ), dY = v3( cam.dy_pixel );
    if ( cam.lens_radius > 0 ) {
        V2 d2     = ConcentricSampleDisk( V2 { uLens, vLens } );
        V2 pLens  = V2 { cam.lens_radius * d2.x, cam.lens_radius * d2.y };
        V3 offset = v3( cam.x_dir ) * pLens.x + v3( cam.up ) * pLens.y;
        V3 d      = ( ntl + s * dX + t * dY ) - pos - offset;
        return Ray( pos + offset, normalize( d ) );
    }
    V3 d = ( ntl + s * dX + t * dY ) - pos;
    return Ray( pos, normalize( d ) );
}

} // namespace orc
