// Device-side fp32 helpers for the gfx950 kernels.
//
// Every function evaluates in the order of the scalar paths of glm 0.9.9.4 (the maths library
// under the reference renderer), with one IEEE rounding per operation: the translation unit is
// compiled with -ffp-contract=off, `/` and sqrtf are the correctly rounded forms, and nothing
// here may be rewritten into an fma or a reciprocal-multiply.  That is what makes hit
// primitive ids bit-identical to the CPU reference.
//   dot(vec3)  = (a.x*b.x + a.y*b.y) + a.z*b.z
//   dot(vec4)  = (x + y) + (z + w)
//   cross      = (x.y*y.z - y.y*x.z, x.z*y.x - y.z*x.x, x.x*y.y - y.x*x.y)
//   normalize  = v * (1 / sqrt(dot(v,v)))
//   mat3 * v   = m[0][r]*x + m[1][r]*y + m[2][r]*z
//   min(x,y)   = (y < x) ? y : x ;  max(x,y) = (x < y) ? y : x
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MRT_DEV __device__ __forceinline__

struct V2 {
    float x, y;
};
struct V3 {
    float x, y, z;
};
struct V4 {
    float x, y, z, w;
};

MRT_DEV V3 mk3( float a, float b, float c ) {
    V3 r;
    r.x = a;
    r.y = b;
    r.z = c;
    return r;
}
MRT_DEV V3 mk3( float a ) {
    return mk3( a, a, a );
}
MRT_DEV float comp( V3 v, int i ) {
    // all three components are read before the choice, so the choice is two selects and not a branch
    const float x = v.x, y = v.y, z = v.z;
    const float yz = i == 1 ? y : z;
    return i == 0 ? x : yz;
}
MRT_DEV V3 operator+( V3 a, V3 b ) {
    return mk3( a.x + b.x, a.y + b.y, a.z + b.z );
}
MRT_DEV V3 operator-( V3 a, V3 b ) {
    return mk3( a.x - b.x, a.y - b.y, a.z - b.z );
}
MRT_DEV V3 operator-( V3 a ) {
    return mk3( -a.x, -a.y, -a.z );
}
MRT_DEV V3 operator*( V3 a, V3 b ) {
    return mk3( a.x * b.x, a.y * b.y, a.z * b.z );
}
MRT_DEV V3 operator*( V3 a, float s ) {
    return mk3( a.x * s, a.y * s, a.z * s );
}
MRT_DEV V3 operator*( float s, V3 a ) {
    return mk3( s * a.x, s * a.y, s * a.z );
}
MRT_DEV V3 operator/( V3 a, float s ) {
    return mk3( a.x / s, a.y / s, a.z / s );
}
MRT_DEV float fmin_glm( float x, float y ) {
    return ( y < x ) ? y : x;
}
MRT_DEV float fmax_glm( float x, float y ) {
    return ( x < y ) ? y : x;
}
MRT_DEV float clampf( float x, float lo, float hi ) {
    return fmin_glm( fmax_glm( x, lo ), hi );
}
MRT_DEV float dot3( V3 a, V3 b ) {
    float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z;
    return tx + ty + tz;
}
MRT_DEV float dot4( V4 a, V4 b ) {
    float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z, tw = a.w * b.w;
    return ( tx + ty ) + ( tz + tw );
}
MRT_DEV V3 cross3( V3 x, V3 y ) {
    return mk3( x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y );
}
MRT_DEV float inversesqrt_glm( float x ) {
    return 1.0f / sqrtf( x );
}
MRT_DEV V3 normalize3( V3 v ) {
    return v * inversesqrt_glm( dot3( v, v ) );
}
MRT_DEV float length3( V3 v ) {
    return sqrtf( dot3( v, v ) );
}
MRT_DEV V3 reflect3( V3 I, V3 N ) {
    return I - N * dot3( N, I ) * 2.0f;
}
MRT_DEV V3 refract3( V3 I, V3 N, float eta ) {
    float dotValue = dot3( N, I );
    float k        = 1.0f - eta * eta * ( 1.0f - dotValue * dotValue );
    if ( k >= 0.0f ) return eta * I - ( eta * dotValue + sqrtf( k ) ) * N;
    return mk3( 0.f );
}
MRT_DEV V3 mix3( V3 x, V3 y, float a ) {
    return x + a * ( y - x );
}

// Transform record on the device: 24 floats.
//   [0..8]  m_invTransform (column-major), [9..11] m_translation, [12..20] m_transform, [21..23] pad
struct DXf {
    float inv[9];
    float t[3];
    float m[9];
    float pad[3];
};
MRT_DEV V3 mat_mul( const float* m, V3 v ) {
    return mk3( m[0] * v.x + m[3] * v.y + m[6] * v.z, m[1] * v.x + m[4] * v.y + m[7] * v.z, m[2] * v.x + m[5] * v.y + m[8] * v.z );
}
// Transform::transposeMult (dot of each column with the vector)
MRT_DEV V3 mat_tmul( const float* m, V3 d ) {
    return mk3( dot3( mk3( m[0], m[1], m[2] ), d ), dot3( mk3( m[3], m[4], m[5] ), d ), dot3( mk3( m[6], m[7], m[8] ), d ) );
}
