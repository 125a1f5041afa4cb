// TEST INFRASTRUCTURE ONLY (CPU oracle) -- see mrt_oracle.h.
//
// Scalar fp32 vector helpers that evaluate in the same order as the glm 0.9.9.4 scalar paths
// the reference is compiled against (libs/glm-master/glm/detail):
//   dot(vec3)   = (a.x*b.x + a.y*b.y) + a.z*b.z                  func_geometric.inl:48-53
//   dot(vec4)   = (x + y) + (z + w)                               func_geometric.inl:58-65
//   cross       = (x.y*y.z - y.y*x.z, x.z*y.x - y.z*x.x, x.x*y.y - y.x*x.y)   :74-77
//   normalize   = v * (1 / sqrt(dot(v,v)))                        :88, func_exponential.inl:134-139
//   mat3 * vec3 = m[0][r]*x + m[1][r]*y + m[2][r]*z               type_mat3x3.inl:468-474
//   reflect / refract                                              func_geometric.inl:104-123
//   mix(x,y,a)  = x + a*(y - x)                                    func_common.inl:103-111
//   min(x,y) = (y<x)?y:x ; max(x,y) = (x<y)?y:x ; clamp = min(max(x,lo),hi)   func_common.inl:16-30,503-507
// Built with -ffp-contract=off: no operation below may be fused.
#pragma once
#include <cmath>
#include <cstdint>

namespace orc {

struct V2 {
    float x, y;
};

struct V3 {
    float x, y, z;
    float operator[]( int i ) const { return i == 0 ? x : ( i == 1 ? y : z ); }
    float& at( int i ) { return i == 0 ? x : ( i == 1 ? y : z ); }
};

struct V4 {
    float x, y, z, w;
};

struct M3 { // column-major like glm: c[col][row]
    float c[3][3];
};

inline V3 v3( float a, float b, float c ) {
    return V3 { a, b, c };
}
inline V3 v3( float a ) {
    return V3 { a, a, a };
}
inline V3 v3( const float* p ) {
    return V3 { p[0], p[1], p[2] };
}
inline V3 operator+( V3 a, V3 b ) {
    return V3 { a.x + b.x, a.y + b.y, a.z + b.z };
}
inline V3 operator-( V3 a, V3 b ) {
    return V3 { a.x - b.x, a.y - b.y, a.z - b.z };
}
inline V3 operator-( V3 a ) {
    return V3 { -a.x, -a.y, -a.z };
}
inline V3 operator*( V3 a, V3 b ) {
    return V3 { a.x * b.x, a.y * b.y, a.z * b.z };
}
inline V3 operator*( V3 a, float s ) {
    return V3 { a.x * s, a.y * s, a.z * s };
}
inline V3 operator*( float s, V3 a ) {
    return V3 { s * a.x, s * a.y, s * a.z };
}
inline V3 operator/( V3 a, float s ) {
    return V3 { a.x / s, a.y / s, a.z / s };
}
inline V3 operator/( V3 a, V3 b ) {
    return V3 { a.x / b.x, a.y / b.y, a.z / b.z };
}
inline V3 operator+( V3 a, float s ) {
    return V3 { a.x + s, a.y + s, a.z + s };
}
inline V3 operator-( V3 a, float s ) {
    return V3 { a.x - s, a.y - s, a.z - s };
}
inline float fmin_glm( float x, float y ) {
    return ( y < x ) ? y : x;
}
inline float fmax_glm( float x, float y ) {
    return ( x < y ) ? y : x;
}
inline float clampf( float x, float lo, float hi ) {
    return fmin_glm( fmax_glm( x, lo ), hi );
}
inline float dot( V3 a, V3 b ) {
    float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z;
    return tx + ty + tz;
}
inline float dot( V4 a, V4 b ) {
    float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z, tw = a.w * b.w;
    return ( tx + ty ) + ( tz + tw );
}
inline float dot( V2 a, V2 b ) {
    float tx = a.x * b.x, ty = a.y * b.y;
    return tx + ty;
}
inline V3 cross( V3 x, V3 y ) {
    return V3 { x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y };
}
inline float inversesqrt( float x ) {
    return 1.0f / std::sqrt( x );
}
inline V3 normalize( V3 v ) {
    return v * inversesqrt( dot( v, v ) );
}
inline V4 normalize( V4 v ) {
    float s = inversesqrt( dot( v, v ) );
    return V4 { v.x * s, v.y * s, v.z * s, v.w * s };
}
inline float length( V3 v ) {
    return std::sqrt( dot( v, v ) );
}
inline float distance( V3 p0, V3 p1 ) {
    return length( p1 - p0 );
}
inline V3 reflect( V3 I, V3 N ) {
    return I - N * dot( N, I ) * 2.0f;
}
inline V3 refract( V3 I, V3 N, float eta ) {
    float dotValue = dot( N, I );
    float k        = 1.0f - eta * eta * ( 1.0f - dotValue * dotValue );
    if ( k >= 0.0f ) return eta * I - ( eta * dotValue + std::sqrt( k ) ) * N;
    return v3( 0.f );
}
inline V3 mix( V3 x, V3 y, float a ) {
    return x + a * ( y - x );
}
inline V3 mul( const M3& m, V3 v ) {
    return V3 { m.c[0][0] * v.x + m.c[1][0] * v.y + m.c[2][0] * v.z,
                m.c[0][1] * v.x + m.c[1][1] * v.y + m.c[2][1] * v.z,
                m.c[0][2] * v.x + m.c[1][2] * v.y + m.c[2][2] * v.z };
}
inline V3 column( const M3& m, int i ) {
    return V3 { m.c[i][0], m.c[i][1], m.c[i][2] };
}
// Transform::transposeMult (defines.h:134-140)
inline V3 transpose_mult( const M3& m, V3 d ) {
    return V3 { dot( column( m, 0 ), d ), dot( column( m, 1 ), d ), dot( column( m, 2 ), d ) };
}
inline M3 mul( const M3& a, const M3& b ) { // type_mat3x3.inl:486-520
    M3 r;
    for ( int j = 0; j < 3; j++ )
        for ( int i = 0; i < 3; i++ )
            r.c[j][i] = a.c[0][i] * b.c[j][0] + a.c[1][i] * b.c[j][1] + a.c[2][i] * b.c[j][2];
    return r;
}
inline M3 inverse( const M3& m_ ) { // func_matrix.inl:269-291
    const float( *m )[3] = m_.c;
    float ood = 1.0f / ( +m[0][0] * ( m[1][1] * m[2][2] - m[2][1] * m[1][2] ) -
                         m[1][0] * ( m[0][1] * m[2][2] - m[2][1] * m[0][2] ) +
                         m[2][0] * ( m[0][1] * m[1][2] - m[1][1] * m[0][2] ) );
    M3 r;
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
inline M3 m3_from( const float* p9 ) {
    M3 r;
    for ( int c = 0; c < 3; c++ )
        for ( int k = 0; k < 3; k++ )
            r.c[c][k] = p9[3 * c + k];
    return r;
}
inline M3 m3_diag( float d ) {
    M3 r {};
    r.c[0][0] = r.c[1][1] = r.c[2][2] = d;
    return r;
}

// Transform (defines.h:67-148)
struct Xf {
    M3 m   = m3_diag( 1.f );
    M3 inv = m3_diag( 1.f );
    V3 t   = V3 { 0, 0, 0 };
    V3 to( V3 p ) const { return mul( inv, p - t ); }             // transformTo
    V3 from( V3 p ) const { return mul( m, p ) + t; }             // transformFrom
    V3 vec_from( V3 d ) const { return transpose_mult( inv, d ); } // vectorTransformFrom
    void apply( const M3& a ) {                                     // transform(mat3)
        m   = mul( a, m );
        t   = mul( a, t );
        inv = inverse( m );
    }
    void translate( V3 p ) { t = t + p; }
    void scale( float sx, float sy, float sz ) {
        M3 a {};
        a.c[0][0] = sx;
        a.c[1][1] = sy;
        a.c[2][2] = sz;
        apply( a );
    }
    void rotate( V3 axis, float degrees ) { // defines.h:91-121
        // degrees_to_radians: degrees * M_PI / 180.0f evaluated in double, returned as float
        float theta = (float)( (double)degrees * 3.14159265358979323846 / (double)180.0f );
        float c     = (float)std::cos( theta ); // std::cos(float) -> float overload
        if ( c == 1 ) return;
        float s   = (float)std::sin( theta );
        float tt  = 1 - c;
        float tx  = tt * axis.x;
        float ty  = tt * axis.y;
        float tz  = tt * axis.z;
        float txy = tx * axis.y;
        float txz = tx * axis.z;
        float tyz = ty * axis.z;
        float sx  = s * axis.x;
        float sy  = s * axis.y;
        float sz  = s * axis.z;
        M3 a {};
        a.c[0][0] = tx * axis.x + c;
        a.c[0][1] = txy + sz;
        a.c[0][2] = txz - sy;
        a.c[1][0] = txy - sz;
        a.c[1][1] = ty * axis.y + c;
        a.c[1][2] = tyz + sx;
        a.c[2][0] = txz + sy;
        a.c[2][1] = tyz - sx;
        a.c[2][2] = tz * axis.z + c;
        apply( a );
    }
};

} // namespace orc
