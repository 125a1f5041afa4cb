// Shading functions of the gfx950 path tracer: Cook-Torrance BSDF, Material::eval/sample, shape
// sampling, ShapeLight / IBL, SimpleCamera.  Each function states the reference lines it follows
// (paths relative to the reference's src/).  Arithmetic order matches the reference; the only
// numerical difference to the CPU path is the last-ulp behaviour of the device's
// expf/logf/powf/sinf/cosf/acosf/atan2f against glibc's.
#pragma once
#include "device_sampler.hpp"
#include "device_trace.hpp"

#define MRT_PI_D 3.14159265358979323846
#define MRT_PI ( (float)MRT_PI_D )

struct DMaterial {
    float albedo[3];
    float emission[3];
    float ior, roughness, metalness;
    int32_t type, texture;
};
struct DTexture {
    int32_t kind;
    uint32_t w, h;
    uint32_t pad;
    uint64_t offset;
    float even[3], odd[3];
    DXf xf;
};
struct DLight {
    int32_t kind;
    float color[3];
    int32_t shape_kind;
    uint32_t xf;
    float p[9];
    float n[9];
    float sphere[4];
};
struct DCamera {
    float pos[3], ntl[3], dx[3], dy[3], xdir[3], up[3];
    float lens_radius;
};

struct DShade {
    const float* shade_n;  // [obj][9] n1 n2 n3
    const float* shade_uv; // [obj][6] tex[0..2]
    const DMaterial* materials;
    const DTexture* textures;
    const float* texels;
    const DLight* lights;
    const float* light_cdf; // Distribution1D::cdf of the uniform light distribution, n_lights + 1
    uint32_t n_lights;
    float light_func_int;   // Distribution1D::funcInt
    const float* env_rgb;
    uint32_t env_w, env_h;
    float env_marginal_int; // Distribution2D::pMarginal->funcInt
    float env_inv[9];
    int32_t env_light;      // light index that is also an environment light, -1 none
    float light0_color[3];
    DCamera cam;
};

MRT_DEV V3 ld3( const float* p ) {
    return mk3( p[0], p[1], p[2] );
}
MRT_DEV bool is_black( V3 c ) { // defines.h:48-50: float sum compared with the double 0.
    return ( c.x + c.y ) + c.z == 0.f;
}

// ------------------------------------------------------------------ sampling.cpp warps
MRT_DEV V2 uniform_sample_triangle( V2 u ) { // sampling.cpp:154-157
    float su0 = sqrtf( u.x );
    V2 r;
    r.x = 1 - su0;
    r.y = u.y * su0;
    return r;
}
MRT_DEV V2 concentric_sample_disk( V2 u ) { // sampling.cpp:113-131
    const float PiOver4 = MRT_PI / 4.f, PiOver2 = MRT_PI / 2.f;
    float ox = 2.f * u.x - 1, oy = 2.f * u.y - 1;
    V2 out;
    out.x = out.y = 0.f;
    if ( ox == 0 && oy == 0 ) return out;
    float theta, r;
    if ( fabsf( ox ) > fabsf( oy ) ) {
        r     = ox;
        theta = PiOver4 * ( oy / ox );
    }
    else {
        r     = oy;
        theta = PiOver2 - PiOver4 * ( ox / oy );
    }
    out.x = r * cosf( theta );
    out.y = r * sinf( theta );
    return out;
}
MRT_DEV V3 uniform_sample_sphere( V2 u ) { // sampling.cpp:95-100
    float z   = 1 - 2 * u.x;
    float r   = sqrtf( fmaxf( 0.f, 1.f - z * z ) );
    float phi = 2 * MRT_PI * u.y;
    return mk3( r * cosf( phi ), r * sinf( phi ), z );
}
MRT_DEV float power_heuristic( float fPdf, float gPdf ) { // sampling.h:181-184 with nf = ng = 1
    float f = 1 * fPdf, g = 1 * gPdf;
    return ( f * f ) / ( f * f + g * g );
}
MRT_DEV void coordinate_system( V3 v1, V3& v2, V3& v3 ) { // defines.h:150-156
    if ( fabsf( v1.x ) > fabsf( v1.y ) ) v2 = mk3( -v1.z, 0, v1.x ) / sqrtf( v1.x * v1.x + v1.z * v1.z );
    else v2 = mk3( 0, v1.z, -v1.y ) / sqrtf( v1.y * v1.y + v1.z * v1.z );
    v3 = cross3( v1, v2 );
}

// Distribution1D::SampleDiscrete (sampling.h:54-67, 103-111) on a cdf of `size` = n+1 entries
MRT_DEV int sample_discrete( const float* cdf, int size, float func_int, float u, float& pdf ) {
    int first = 0, len = size;
    while ( len > 0 ) {
        int half = len >> 1, middle = first + half;
        if ( cdf[middle] <= u ) {
            first = middle + 1;
            len -= half + 1;
        }
        else
            len = half;
    }
    int r      = first - 1;
    int offset = r < 0 ? 0 : ( r > size - 2 ? size - 2 : r );
    int count  = size - 1;
    // uniform distribution: func[offset] == 1
    pdf = ( func_int > 0 ) ? 1.f / ( func_int * count ) : 0;
    return offset;
}

// ------------------------------------------------------------------ bsdf.cpp
MRT_DEV float rdm_beckmann( float NdotH, float alpha ) { // bsdf.cpp:43-58
    float phi          = ( NdotH > 0.f ) ? 1.f : 0.f;
    float cos2         = NdotH * NdotH;
    float tan2         = ( 1 - ( cos2 ) ) / cos2;
    float alpha2       = alpha * alpha;
    float e            = expf( -tan2 / ( alpha2 ) );
    float pi_alpha_cos = (float)( MRT_PI_D * (double)alpha2 * (double)( cos2 * cos2 ) );
    return phi * ( e / pi_alpha_cos );
}
MRT_DEV float rdm_fresnel( float LdotH, float extIOR, float intIOR ) { // bsdf.cpp:100-118
    LdotH = ( LdotH < -1.f ) ? -1.f : ( ( 1.f < LdotH ) ? 1.f : LdotH ); // std::clamp
    if ( !( LdotH > 0.f ) ) {
        float tmp = extIOR;
        extIOR    = intIOR;
        intIOR    = tmp;
        LdotH     = fabsf( LdotH );
    }
    float sinThetaI = sqrtf( fmaxf( 0.f, 1.f - LdotH * LdotH ) );
    float sinThetaT = extIOR / intIOR * sinThetaI;
    if ( sinThetaT >= 1 ) return 1.f;
    float cosThetaT = sqrtf( fmaxf( 0.f, 1.f - sinThetaT * sinThetaT ) );
    float Rparl     = ( ( intIOR * LdotH ) - ( extIOR * cosThetaT ) ) / ( ( intIOR * LdotH ) + ( extIOR * cosThetaT ) );
    float Rperp     = ( ( extIOR * LdotH ) - ( intIOR * cosThetaT ) ) / ( ( extIOR * LdotH ) + ( intIOR * cosThetaT ) );
    return ( Rparl * Rparl + Rperp * Rperp ) / 2.f;
}
MRT_DEV float rdm_g1( float DdotN, float alpha ) { // bsdf.cpp:141-154
    float tanx = sqrtf( 1.f - fmin_glm( 0.99999f, DdotN * DdotN ) );
    float b    = DdotN / ( fmax_glm( 0.0001f, alpha ) * tanx );
    if ( b < 1.6f ) return ( ( 3.535f + 2.181f * b ) * b ) / ( 1.f + ( 2.276f + 2.577f * b ) * b );
    return 1.f;
}

struct BrdfData { // Material.h:16-31
    float LdotH, NdotH, VdotH, LdotN, VdotN;
    V3 F, diffuse;
    float alpha, alphaSq;
};

MRT_DEV V3 rdm_bsdf( const BrdfData& d ) { // bsdf.cpp:190-200, 219-259 (COMBINE_BRDFS_WITH_FRESNEL 1)
    float D     = rdm_beckmann( d.NdotH, d.alpha );
    float g     = rdm_g1( d.LdotN, d.alpha ) * rdm_g1( d.VdotN, d.alpha );
    V3 specular = ( D * d.F * g ) / ( 4.f * d.LdotN * d.VdotN );
    V3 diffuse  = d.diffuse / MRT_PI;
    return ( mk3( 1.f ) - d.F ) * diffuse + specular;
}

MRT_DEV bool compute_brdf_data( BrdfData& data, V3 v, V3 wi, V3 n, V3 base, float metalness, float roughness ) {
    // Material.cpp:1277-1316, Material.h:40-76
    V3 h        = normalize3( v + wi );
    float LdotN = dot3( wi, n );
    float VdotN = dot3( v, n );
    if ( LdotN <= 0.0f || VdotN <= 0.0f ) return false;
    data.LdotN   = fmin_glm( fmax_glm( 0.0001f, LdotN ), 1.0f );
    data.VdotN   = fmin_glm( fmax_glm( 0.0001f, VdotN ), 1.0f );
    data.LdotH   = clampf( dot3( wi, h ), 0.0f, 1.0f );
    data.NdotH   = clampf( dot3( n, h ), 0.0f, 1.0f );
    data.VdotH   = clampf( dot3( v, h ), 0.0f, 1.0f );
    V3 f0        = mix3( mk3( 0.04f ), base, metalness );
    data.diffuse = base * ( 1.0f - metalness );
    data.alpha   = roughness;
    data.alphaSq = data.alpha * data.alpha;
    const float t = ( 1.0f / 0.04f );
    float f90    = fmin_glm( 1.0f, t * dot3( f0, mk3( 0.2126f, 0.7152f, 0.0722f ) ) );
    float p      = powf( 1.0f - data.LdotH, 5.0f );
    data.F       = f0 + mk3( f90 - f0.x, f90 - f0.y, f90 - f0.z ) * p;
    return true;
}
MRT_DEV float beckmann_walter_reflection_pdf( float alphaSquared, float NdotH, float LdotH ) { // Material.cpp:220-239
    NdotH             = fmax_glm( 0.00001f, NdotH );
    LdotH             = fmax_glm( 0.00001f, LdotH );
    alphaSquared      = fmax_glm( 0.00001f, alphaSquared );
    float cos2Theta   = NdotH * NdotH;
    float numerator   = expf( ( cos2Theta - 1.0f ) / ( alphaSquared * cos2Theta ) );
    float denominator = MRT_PI * alphaSquared * cos2Theta * cos2Theta;
    return numerator / denominator * NdotH / ( 4.0f * LdotH );
}
MRT_DEV V3 sample_beckmann_walter( float alpha, V2 u ) { // Material.cpp:55-68
    alpha                 = alpha * 0.5f + alpha * 0.5f; // dot(vec2(alpha, alpha), vec2(0.5, 0.5))
    float tanThetaSquared = -( alpha * alpha ) * logf( 1.0f - u.x );
    float phi             = ( 2.f * MRT_PI ) * u.y;
    float cosTheta        = inversesqrt_glm( 1.0f + tanThetaSquared );
    float sinTheta        = sqrtf( 1.0f - cosTheta * cosTheta );
    return normalize3( mk3( sinTheta * cosf( phi ), sinTheta * sinf( phi ), cosTheta ) );
}
MRT_DEV V4 rotation_to_z( V3 in ) { // Material.cpp:72-78
    V4 q;
    if ( in.z < -0.99999f ) {
        q.x = 1.0f;
        q.y = q.z = q.w = 0.0f;
        return q;
    }
    q.x     = in.y;
    q.y     = -in.x;
    q.z     = 0.0f;
    q.w     = 1.0f + in.z;
    float s = inversesqrt_glm( dot4( q, q ) );
    q.x *= s;
    q.y *= s;
    q.z *= s;
    q.w *= s;
    return q;
}
MRT_DEV V3 rotate_point( V4 q, V3 v ) { // Material.cpp:116-120
    V3 a = mk3( q.x, q.y, q.z );
    return 2.0f * dot3( a, v ) * a + ( q.w * q.w - dot3( a, a ) ) * v + 2.0f * q.w * cross3( a, v );
}

// ------------------------------------------------------------------ textures.hpp:21-43, 64-125
MRT_DEV int texel_index( float f, uint32_t n ) {
    if ( !( f >= 0.f ) ) return 0;
    if ( f >= (float)( n - 1 ) ) return (int)n - 1;
    return (int)f;
}
MRT_DEV V3 texture_value( const DShade& s, int32_t ti, float u, float v ) {
    const DTexture& t = s.textures[ti];
    V3 uvw = mat_mul( t.xf.inv, mk3( u, v, 0.f ) - mk3( t.xf.t[0], t.xf.t[1], t.xf.t[2] ) );
    u      = uvw.x;
    v      = uvw.y;
    float ux = u - (float)(int)u, vy = v - (float)(int)v;
    if ( ux < 0.0f ) ux = (float)( (double)ux + 1.0 );
    if ( vy < 0.0f ) vy = (float)( (double)vy + 1.0 );
    if ( t.kind == 1 ) {
        bool even = ( ux <= 0.5f ) == ( vy <= 0.5f );
        return even ? ld3( t.even ) : ld3( t.odd );
    }
    // ux, vy are in [0, 1] for every finite coordinate below 2^31.  A NaN coordinate (a NaN ray
    // "hits" a plane: every comparison of Plane::intersect is false) or one beyond the int range
    // makes the reference's float -> int conversions undefined; texel_index defines them (not >= 0
    // -> 0, beyond the last texel -> the last texel) and is the plain conversion otherwise.
    int u2     = texel_index( ux * (float)( t.w - 1 ), t.w );
    int v2     = texel_index( vy * (float)( t.h - 1 ), t.h );
    uint64_t p = (uint64_t)( t.h - v2 - 1 ) * t.w + u2;
    return ld3( s.texels + t.offset + 3 * p );
}

// The surface point a path is shading (Intersection, intersection.h:6-43, as far as it is read).
struct Surf {
    V3 position, normal;
    float u, v; // texture coordinates
    uint32_t material;
    bool outside;
};

MRT_DEV V3 material_albedo( const DShade& s, const DMaterial& m, const Surf& sf ) {
    // every path ray has hasDifferentials == true (ray.h:30), so a texture is always used
    if ( m.texture >= 0 ) return texture_value( s, m.texture, sf.u, sf.v );
    return ld3( m.albedo );
}

// Material::eval (Material.cpp:983-1013)
MRT_DEV V3 material_eval( const DShade& s, const DMaterial& m, V3 raydir, const Surf& sf, V3 wi, float& pdf ) {
    if ( m.type == 2 ) return mk3( 0.f );
    BrdfData data;
    V3 bsdf = mk3( 0.f );
    if ( compute_brdf_data( data, -raydir, wi, sf.normal, material_albedo( s, m, sf ), m.metalness, m.roughness ) ) {
        bsdf = rdm_bsdf( data );
        pdf  = beckmann_walter_reflection_pdf( data.alphaSq, data.NdotH, data.LdotH );
    }
    return bsdf;
}

// Material::sample (Material.cpp:1015-1100); engine draws come from the path's PCG32 stream
MRT_DEV V3 material_sample( const DShade& s, const DMaterial& m, V3 raydir, const Surf& sf, Pcg32& rng, V3& wi, float& pdf ) {
    if ( m.type == 2 ) {
        V3 normal   = sf.outside ? sf.normal : -sf.normal;
        float LdotH = dot3( -raydir, normal );
        float f     = sf.outside ? rdm_fresnel( LdotH, 1.f, m.ior ) : rdm_fresnel( LdotH, m.ior, 1.f );
        if ( f < 1.0f ) {
            if ( pcg_float( rng ) < f ) wi = reflect3( raydir, normal );
            else wi = refract3( raydir, normal, sf.outside ? ( 1.f / m.ior ) : m.ior );
        }
        else { wi = reflect3( raydir, normal ); }
        pdf = 1.f;
        return mk3( 1.f );
    }
    V3 V = -raydir;
    V2 u;
    u.x     = pcg_float( rng );
    u.y     = pcg_float( rng );
    V3 bsdf = mk3( 0.f );
    if ( !sf.outside ) return bsdf;
    V4 q        = rotation_to_z( sf.normal );
    V3 Vlocal   = rotate_point( q, V );
    float alpha = m.roughness;
    V3 Hlocal   = ( alpha == 0.0f ) ? mk3( 0.f, 0.f, 1.f ) : sample_beckmann_walter( alpha, u );
    V3 Llocal   = reflect3( -Vlocal, Hlocal );
    V4 qi;
    qi.x = -q.x;
    qi.y = -q.y;
    qi.z = -q.z;
    qi.w = q.w;
    wi   = normalize3( rotate_point( qi, Llocal ) );
    BrdfData data;
    if ( compute_brdf_data( data, -raydir, wi, sf.normal, material_albedo( s, m, sf ), m.metalness, m.roughness ) ) {
        bsdf = rdm_bsdf( data );
        pdf  = beckmann_walter_reflection_pdf( data.alphaSq, data.NdotH, data.LdotH );
    }
    return bsdf;
}

// ------------------------------------------------------------------ light shapes
MRT_DEV float triangle_area( const DLight& l ) { // triangle.cpp:99-104 (0.5 is a double)
    V3 p0 = ld3( l.p ), p1 = ld3( l.p + 3 ), p2 = ld3( l.p + 6 );
    return (float)( 0.5 * (double)length3( cross3( p1 - p0, p2 - p0 ) ) );
}

struct ShapePoint {
    V3 position, normal;
};

// Object::sample(inter, u, &pdf) (triangle.cpp:86-97, sphere.cpp:182-223); pdf is left untouched on
// the inside-the-sphere branch, as in the reference
MRT_DEV ShapePoint light_shape_sample( const DTrace& tr, const DLight& l, const Surf& inter, V2 u, float& pdf ) {
    ShapePoint it;
    if ( l.shape_kind == MRT_TRIANGLE ) {
        V2 b        = uniform_sample_triangle( u );
        V3 p0 = ld3( l.p ), p1 = ld3( l.p + 3 ), p2 = ld3( l.p + 6 );
        V3 n1 = ld3( l.n ), n2 = ld3( l.n + 3 ), n3 = ld3( l.n + 6 );
        float w     = 1.f - b.x - b.y;
        it.position = b.x * p0 + b.y * p1 + w * p2;
        it.normal   = normalize3( n2 * b.x + n3 * b.y + n1 * w );
        pdf         = 1 / triangle_area( l );
        return it;
    }
    const DXf& xf  = tr.xforms[l.xf];
    V3 pCenter     = ld3( l.sphere );
    float radius   = l.sphere[3];
    V3 wc          = normalize3( pCenter - inter.position );
    V3 wcX, wcY;
    coordinate_system( wc, wcX, wcY );
    V3 normal     = inter.outside ? inter.normal : -inter.normal;
    V3 pOrigin    = inter.position + ( MRT_ACNE_EPS * normal );
    float dist    = length3( pOrigin - pCenter );
    float radius2 = radius * radius;
    if ( ( dist * dist ) <= radius2 ) { // Sphere::sample(u), sphere.cpp:182-189
        V3 pObj     = radius * uniform_sample_sphere( u );
        it.normal   = normalize3( mat_mul( xf.m, pObj ) );
        pObj        = pObj * ( radius / length3( pObj ) );
        it.position = mat_mul( xf.m, pObj ) + ld3( xf.t );
        return it;
    }
    dist               = length3( inter.position - pCenter );
    float sinThetaMax2 = radius2 / ( dist * dist );
    float cosThetaMax  = sqrtf( fmax_glm( 0.f, 1.f - sinThetaMax2 ) );
    float cosTheta     = ( 1.f - u.x ) + u.x * cosThetaMax;
    float sinTheta     = sqrtf( fmax_glm( 0.f, 1.f - cosTheta * cosTheta ) );
    float phi          = u.y * 2.f * MRT_PI;
    float ds           = dist * cosTheta - sqrtf( fmax_glm( 0.f, radius2 - dist * dist * sinTheta * sinTheta ) );
    float cosAlpha     = ( dist * dist + radius2 - ds * ds ) / ( 2.f * dist * radius );
    float sinAlpha     = sqrtf( fmax_glm( 0.f, 1.f - cosAlpha * cosAlpha ) );
    // SphericalDirection(sinAlpha, cosAlpha, phi, -wcX, -wcY, -wc) (defines.h:158-165)
    V3 nObj     = sinAlpha * cosf( phi ) * ( -wcX ) + sinAlpha * sinf( phi ) * ( -wcY ) + cosAlpha * ( -wc );
    V3 pObj     = radius * nObj;
    pObj        = pObj * ( radius / length3( pObj ) );
    it.position = mat_mul( xf.m, pObj ) + ld3( xf.t );
    it.normal   = mat_mul( xf.m, nObj );
    pdf         = 1 / ( 2 * MRT_PI * ( 1 - cosThetaMax ) );
    return it;
}

// Object::pdf / Sphere::pdf (Object.cpp:13-22, sphere.cpp:225-236)
MRT_DEV float light_shape_pdf( const DTrace& tr, const DLight& l, const Surf& inter, V3 wi ) {
    const DXf& xf = tr.xforms[l.xf];
    if ( l.shape_kind == MRT_SPHERE ) {
        float radius = l.sphere[3];
        V3 pCenter   = mat_mul( xf.m, mk3( 0.f ) ) + ld3( xf.t );
        V3 pOrigin   = inter.position + ( MRT_ACNE_EPS * inter.normal );
        V3 dc        = pOrigin - pCenter;
        if ( dot3( dc, dc ) <= radius * radius ) return 1.f;
        V3 dp              = inter.position - pCenter;
        float sinThetaMax2 = radius * radius / dot3( dp, dp );
        float cosThetaMax  = sqrtf( fmax_glm( 0.f, 1.f - sinThetaMax2 ) );
        return 1 / ( 2 * MRT_PI * ( 1 - cosThetaMax ) );
    }
    // re-intersect the light's own triangle: Triangle::intersect (triangle.cpp:4-44) on
    // Ray(position + wi*eps, wi, 0, 10000)
    V3 ro = inter.position + wi * MRT_ACNE_EPS, rd = wi;
    V3 origin, dir;
    if ( l.xf == 0 ) {
        origin = ro;
        dir    = rd;
    }
    else {
        origin = mat_mul( xf.inv, ro - ld3( xf.t ) );
        dir    = mat_mul( xf.inv, rd );
    }
    V3 p1 = ld3( l.p ), v1v2 = ld3( l.p + 3 ) - p1, v1v3 = ld3( l.p + 6 ) - p1;
    V3 c      = cross3( dir, v1v3 );
    float det = dot3( v1v2, c );
    if ( det > -MRT_ACNE_EPS && det < MRT_ACNE_EPS ) return 0;
    float inv_det = 1.f / det;
    V3 sv         = origin - p1;
    float u       = dot3( sv, c ) * inv_det;
    if ( u < 0.f || u > 1.f ) return 0;
    V3 q    = cross3( sv, v1v2 );
    float v = dot3( dir, q ) * inv_det;
    if ( v < 0.f || u + v > 1.f ) return 0;
    float t = dot3( v1v3, q ) * inv_det;
    if ( !( t >= 0.f && t <= 10000.f ) ) return 0;
    V3 position     = ro + ( t * rd );
    V3 objectNormal = normalize3( ld3( l.n + 3 ) * u + ld3( l.n + 6 ) * v + ld3( l.n ) * ( 1.f - u - v ) );
    V3 normal       = normalize3( mat_tmul( xf.inv, objectNormal ) );
    V3 dp           = position - inter.position;
    float pdf       = dot3( dp, dp ) / ( fabsf( dot3( normal, -wi ) * triangle_area( l ) ) );
    if ( isinf( pdf ) ) pdf = 0.f;
    return pdf;
}

// ------------------------------------------------------------------ IBL (Light.cpp:277-342)
MRT_DEV V3 env_texel( const DShade& s, int i, int j ) {
    long index = 3L * i + 3L * (long)s.env_w * j;
    long last  = 3L * ( (long)s.env_w * s.env_h - 1 );
    if ( index > last ) index = last; // theta == pi indexes one row past the image in the reference
    if ( index < 0 ) index = 0;
    return ld3( s.env_rgb + index );
}
// Column / row of a direction.  theta is NaN when rounding leaves |dir.y| a unit in the last place
// above 1, and the reference's conversion of a NaN to int is undefined: 0 here (the pole's row).
MRT_DEV int env_coord( double x ) {
    return x == x ? (int)x : 0;
}
MRT_DEV V3 ibl_Le( const DShade& s, V3 raydir ) { // Light.cpp:331-342 (double theta / phi)
    V3 dir       = mat_mul( s.env_inv, raydir );
    double theta = (double)acosf( dir.y );
    double phi   = (double)atan2f( dir.z, dir.x );
    if ( phi < 0 ) phi += 2 * MRT_PI_D;
    int i = env_coord( phi / ( 2 * MRT_PI_D ) * (double)(int)s.env_w );
    int j = env_coord( theta / MRT_PI_D * (double)(int)s.env_h );
    return env_texel( s, i, j );
}
MRT_DEV V3 ibl_sample_texel( const DShade& s, V3 raydir ) { // Light.cpp:304-314 (float theta / phi)
    V3 dir      = mat_mul( s.env_inv, raydir );
    float theta = acosf( dir.y );
    float phi   = atan2f( dir.z, dir.x );
    if ( phi < 0 ) phi = (float)( (double)phi + 2 * MRT_PI_D );
    int i = env_coord( (double)phi / ( 2 * MRT_PI_D ) * (double)(int)s.env_w );
    int j = env_coord( (double)theta / MRT_PI_D * (double)(int)s.env_h );
    return env_texel( s, i, j );
}
MRT_DEV float ibl_pdf_Li( const DShade& s, V3 wi ) { // Light.cpp:317-329, sampling.h:143-150
    const float InvPi = 1.f / MRT_PI, Inv2Pi = 1.f / ( 2.f * MRT_PI );
    V3 dir      = mat_mul( s.env_inv, wi );
    float theta = acosf( dir.y );
    float phi   = atan2f( dir.z, dir.x );
    if ( phi < 0 ) phi = (float)( (double)phi + 2 * MRT_PI_D );
    float sinTheta = sinf( theta );
    if ( sinTheta == 0 ) return 0;
    int cu = (int)s.env_w, cv = (int)s.env_h;
    int iu = (int)( ( phi * Inv2Pi ) * (float)cu );
    int iv = (int)( ( theta * InvPi ) * (float)cv );
    iu     = iu < 0 ? 0 : ( iu > cu - 1 ? cu - 1 : iu );
    iv     = iv < 0 ? 0 : ( iv > cv - 1 ? cv - 1 : iv );
    // the Distribution2D is built over the interleaved RGB floats read as a w x h scalar image
    float f = s.env_rgb[(size_t)iv * cu + iu];
    return ( f / s.env_marginal_int ) / ( 2 * MRT_PI * MRT_PI * sinTheta );
}
// random_dir (sampling.h:199-211) on its five engine draws u[0..4], in drawing order.
// vec3( uniform01(engine) - 0.5, ... ): the reference as g++ builds it evaluates the three
// constructor arguments last to first, so the third draw is z and the fifth x.
MRT_DEV V3 random_dir_from_draws( V3 normal, const float* u ) {
    float r1    = u[0];
    float r2    = u[1];
    float theta = 2.f * MRT_PI * r1;
    float r     = sqrtf( 1.f - r2 );
    V3 local    = mk3( cosf( theta ) * r, sinf( theta ) * r, sqrtf( r2 ) );
    float rz    = (float)( (double)u[2] - 0.5 );
    float ry    = (float)( (double)u[3] - 0.5 );
    float rx    = (float)( (double)u[4] - 0.5 );
    V3 tangent1 = normalize3( cross3( normal, mk3( rx, ry, rz ) ) );
    V3 tangent2 = cross3( tangent1, normal );
    return local.z * normal + local.x * tangent1 + local.y * tangent2;
}
MRT_DEV V3 random_dir( V3 normal, Pcg32& rng ) {
    float u[5];
    for ( int k = 0; k < 5; k++ )
        u[k] = pcg_float( rng );
    return random_dir_from_draws( normal, u );
}

// ------------------------------------------------------------------ SimpleCamera::get_ray (Camera.cpp:37-67)
MRT_DEV void camera_ray( const DCamera& cam, float sx, float sy, float uLens, float vLens, V3& o, V3& d ) {
    V3 pos = ld3( cam.pos ), ntl = ld3( cam.ntl ), dX = ld3( cam.dx ), dY = ld3( cam.dy );
    if ( cam.lens_radius > 0 ) {
        V2 u;
        u.x       = uLens;
        u.y       = vLens;
        V2 disk   = concentric_sample_disk( u );
        float lx = cam.lens_radius * disk.x, ly = cam.lens_radius * disk.y;
        V3 offset = ld3( cam.xdir ) * lx + ld3( cam.up ) * ly;
        V3 dd     = ( ntl + sx * dX + sy * dY ) - pos - offset;
        o         = pos + offset;
        d         = normalize3( dd );
        return;
    }
    V3 dd = ( ntl + sx * dX + sy * dY ) - pos;
    o     = pos;
    d     = normalize3( dd );
}

// ------------------------------------------------------------------ hit record -> surface point
// Recomputes what Object::intersect stores in the Intersection for a closest hit
// (triangle.cpp:36-49, sphere.cpp:134-158, plane.cpp:32-43) from (prim, t, barycentrics).
MRT_DEV Surf surface_from_hit( const DTrace& tr, const DShade& s, V3 o, V3 d, const DHit& h ) {
    Surf sf;
    const uint2 info     = tr.prim_info[h.prim];
    const uint32_t flags = info.x;
    const uint32_t kind  = flags & 3u;
    const uint32_t xfi   = flags >> MRT_PRIM_XF_SHIFT;
    const DXf& xf        = tr.xforms[xfi];
    sf.material          = info.y;
    sf.outside           = h.outside != 0;
    sf.position          = o + ( h.t * d );
    if ( kind == MRT_PRIM_TRI ) {
        const float* n  = s.shade_n + 9 * (size_t)h.prim;
        const float* uv = s.shade_uv + 6 * (size_t)h.prim;
        float u = h.u, v = h.v, w = 1.f - u - v;
        V3 objectNormal = normalize3( ld3( n + 3 ) * u + ld3( n + 6 ) * v + ld3( n ) * w );
        sf.normal       = normalize3( mat_tmul( xf.inv, objectNormal ) );
        sf.u            = uv[2] * u + uv[4] * v + uv[0] * w;
        sf.v            = uv[3] * u + uv[5] * v + uv[1] * w;
        return sf;
    }
    if ( kind == MRT_PRIM_PLANE ) {
        sf.normal = normalize3( mat_tmul( xf.inv, mk3( 0.f, 0.f, 1.f ) ) );
        sf.u      = h.u;
        sf.v      = h.v;
        return sf;
    }
    // sphere (two-root branch, sphere.cpp:134-158)
    V3 origin, dir;
    if ( xfi == 0 ) {
        origin = o;
        dir    = d;
    }
    else {
        origin = mat_mul( xf.inv, o - ld3( xf.t ) );
        dir    = mat_mul( xf.inv, d );
    }
    V3 objectPoint  = origin + ( h.t * dir );
    V3 objectNormal = normalize3( objectPoint );
    sf.normal       = normalize3( mat_tmul( xf.inv, objectNormal ) );
    float radius    = length3( objectPoint );
    float phi       = atan2f( objectPoint.y, objectPoint.x );
    if ( phi < 0 ) phi += 2.f * MRT_PI;
    sf.u        = (float)( 1.0 - (double)phi / ( 2.0 * MRT_PI_D ) );
    float theta = acosf( clampf( objectPoint.z / radius, -1.f, 1.f ) );
    sf.v        = (float)( (double)theta / MRT_PI_D );
    return sf;
}
