// Internal fp32 helpers of the host library.  Evaluation order follows the scalar paths of
// glm 0.9.9.4 (the maths library the reference is built on), because the kd-tree the host builds
// and the transforms it uploads must be bit-identical to the reference's:
//   dot = (a.x*b.x + a.y*b.y) + a.z*b.z, cross component order, normalize = v * (1/sqrt(dot)),
//   mat3*vec3 = m[0][r]*x + m[1][r]*y + m[2][r]*z, mat3 inverse by cofactors * (1/det).
// Compiled with -ffp-contract=off.
#pragma once
#include "mrt_host.hpp"

#include <cmath>

namespace mrt {

inline vec3 operator+( vec3 a, vec3 b ) {
    return vec3( a.x + b.x, a.y + b.y, a.z + b.z );
}
inline vec3 operator-( vec3 a, vec3 b ) {
    return vec3( a.x - b.x, a.y - b.y, a.z - b.z );
}
inline vec3 operator-( vec3 a ) {
    return vec3( -a.x, -a.y, -a.z );
}
inline vec3 operator*( vec3 a, float s ) {
    return vec3( a.x * s, a.y * s, a.z * s );
}
inline vec3 operator*( float s, vec3 a ) {
    return vec3( s * a.x, s * a.y, s * a.z );
}
inline vec3 operator+( vec3 a, float s ) {
    return vec3( a.x + s, a.y + s, a.z + s );
}
inline vec3 operator-( vec3 a, float s ) {
    return vec3( a.x - s, a.y - s, a.z - s );
}
inline float dot( vec3 a, vec3 b ) {
    float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z;
    return tx + ty + tz;
}
inline vec3 cross( vec3 x, vec3 y ) {
    return vec3( x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y );
}
inline vec3 normalize( vec3 v ) {
    return v * ( 1.0f / std::sqrt( dot( v, v ) ) );
}
inline float length( vec3 v ) {
    return std::sqrt( dot( v, v ) );
}
inline vec3 mul( const mat3& m, vec3 v ) {
    return vec3( m.c[0][0] * v.x + m.c[1][0] * v.y + m.c[2][0] * v.z,
                 m.c[0][1] * v.x + m.c[1][1] * v.y + m.c[2][1] * v.z,
                 m.c[0][2] * v.x + m.c[1][2] * v.y + m.c[2][2] * v.z );
}
inline mat3 mul( const mat3& a, const mat3& b ) {
    mat3 r;
    for ( int j = 0; j < 3; j++ )
        for ( int i = 0; i < 3; i++ )
            r.c[j][i] = a.c[0][i] * b.c[j][0] + a.c[1][i] * b.c[j][1] + a.c[2][i] * b.c[j][2];
    return r;
}
inline mat3 inverse( const mat3& mm ) {
    const float( *m )[3] = mm.c;
    float ood = 1.0f / ( +m[0][0] * ( m[1][1] * m[2][2] - m[2][1] * m[1][2] ) -
                         m[1][0] * ( m[0][1] * m[2][2] - m[2][1] * m[0][2] ) +
                         m[2][0] * ( m[0][1] * m[1][2] - m[1][1] * m[0][2] ) );
    mat3 r;
    r.c[0][0] = +( m[1][1] * m[2][2] - m[2][1] * m[1][2] ) * ood;
    r.c[1][0] = -( m[1][0] * m[2][2] - m[2][0] * m[1][2] ) * ood;
    r.c[2][0] = +( m[1][0] * m[2][1] - m[2][0] * m[1][1] ) * ood;
    r.c[0][1] = -( m[0][1] * m[2][2] - m[2][1] * m[0][2] ) * ood;
    r.c[1][1] = +( m[0][0] * m[2][2] - m[2][0] * m[0][2] ) * ood;
    r.c[2][1] = -( m[0][0] * m[2][1] - m[2][0] * m[0][1] ) * ood;
    r.c[0][2] = +( m[0][1] * m[1][2] - m[1][1] * m[0][2] ) * ood;
    r.c[1][2] = -( m[0][0] * m[1][2] - m[1][0] * m[0][2] ) * ood;
    r.c[2][2] = +( m[0][0] * m[1][1] - m[1][0] * m[0][1] ) * ood;
    return r;
}
inline mat3 zero3() {
    mat3 r;
    for ( int i = 0; i < 3; i++ )
        for ( int j = 0; j < 3; j++ )
            r.c[i][j] = 0.f;
    return r;
}

// png_read.cpp: r, g, b bytes of every pixel as lodepng::decode's default RGBA8 output holds them
bool decode_png_rgb8( const unsigned char* file, size_t size, uint32_t& w, uint32_t& h, std::vector<unsigned char>& rgb );
bool load_png_rgb8( const char* filename, uint32_t& w, uint32_t& h, std::vector<unsigned char>& rgb );

} // namespace mrt
