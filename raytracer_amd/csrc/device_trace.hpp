// kd-tree traversal and ray/primitive intersection for the gfx950 kernels.
//
// One ray per lane.  The algorithm, its comparisons and its arithmetic are those of the
// reference's intersectKdTree (kdtree.cpp:162-180, 467-613; shapes/triangle.cpp:4-40,
// shapes/sphere.cpp:52-133, shapes/plane.cpp:6-30), so the primitive id, t and barycentrics of
// every hit are bit-identical to the CPU path:
//   - the root box (already padded by the builder) clips [tmin, tmax] with the multiply-by-invdir
//     slab test; inner nodes use a true division t = (split - o[axis]) / d[axis];
//   - near = left iff (split - o[axis]) > 0; (t > b || t < 0) -> near only; t < a -> far only;
//     otherwise the far child is deferred with interval [t, b] and the near child continues
//     with [a, t];
//   - at a leaf the ray interval becomes [a, b] (closed on both ends); primitives are tested in
//     stored order, each accepted hit shrinks tmax, a later hit replaces the current one only if
//     strictly nearer, and the search ends at the first leaf that produced a hit;
//   - shadow (any-hit) rays stop at the first hit on a non-emissive primitive; hits on emissive
//     primitives are ignored but still shrink tmax for the rest of that leaf;
//   - planes live outside the tree and are tested afterwards against the ORIGINAL interval
//     (upper end replaced by the tree hit distance if there was one), as a closest-hit ray.
//
// The deferred-node stack stores (node, tmax) only: the entry interval of a popped node is always
// the exit value b of the leaf visited just before it (the same float, carried by assignment),
// so it is not stored.
#pragma once
#include "device_math.hpp"

#define MRT_STACK_MAX 64
#define MRT_ACNE_EPS 1e-4f

// prim record: 3 x float4
//   q0 = p1.xyz , flags      flags: bits 0-1 kind (0 triangle, 1 sphere, 2 plane),
//   q1 = e1.xyz , e2.x              bit 2 emission != (0,0,0) component-wise (kdtree.cpp:533-534),
//   q2 = e2.yz  , material , 0      bit 3 emission.r+g+b != 0 (isBlack, integrator.cpp:110),
//                                   bits 4.. transform index (0 = identity)
//   e1 = p2 - p1, e2 = p3 - p1 (fp32, as triangle.cpp:9-10)
#define MRT_PRIM_TRI 0u
#define MRT_PRIM_SPHERE 1u
#define MRT_PRIM_PLANE 2u
#define MRT_PRIM_EMISSIVE 4u
#define MRT_PRIM_EMISSIVE_SUM 8u
#define MRT_PRIM_XF_SHIFT 4

struct DTrace {
    const uint2* nodes;
    const uint32_t* refs;
    const float4* prims;
    const DXf* xforms;
    const uint32_t* out_of_tree;
    uint32_t n_nodes;
    uint32_t n_out;
    float root_min[3];
    float root_max[3];
};

struct DHit {
    int32_t prim;
    float t, u, v;
    uint32_t outside; // Intersection::isOutside
};

struct DStats {
    uint32_t inner, leaves, refs, tests;
};

// Per-lane cache of the ray in the object space of the most recently used transform.
struct XfRay {
    uint32_t id;
    V3 o, d;
};

MRT_DEV void to_object( const DTrace& s, uint32_t xf, V3 wo, V3 wd, XfRay& c ) {
    if ( xf == c.id ) return;
    c.id = xf;
    if ( xf == 0 ) { // identity: 1*x + 0*y + 0*z leaves every finite component unchanged
        c.o = wo;
        c.d = wd;
        return;
    }
    const DXf& t = s.xforms[xf];
    // Object::transformRay (Object.cpp:8-11): o' = inv * (o - T), d' = inv * d
    c.o = mat_mul( t.inv, wo - mk3( t.t[0], t.t[1], t.t[2] ) );
    c.d = mat_mul( t.inv, wd );
}

// Object::intersect for one primitive against the ray interval [tmin, tmax]; on acceptance tmax
// becomes the hit distance.
MRT_DEV bool prim_intersect( const DTrace& s, uint32_t obj, V3 wo, V3 wd, float tmin, float& tmax, XfRay& xr, DHit& h ) {
    const float4 q0      = s.prims[3 * (size_t)obj];
    const uint32_t flags = __float_as_uint( q0.w );
    const uint32_t kind  = flags & 3u;
    to_object( s, flags >> MRT_PRIM_XF_SHIFT, wo, wd, xr );
    const V3 origin = xr.o, dir = xr.d;
    if ( kind == MRT_PRIM_TRI ) {
        const float4 q1 = s.prims[3 * (size_t)obj + 1];
        const float4 q2 = s.prims[3 * (size_t)obj + 2];
        const V3 p1 = mk3( q0.x, q0.y, q0.z ), v1v2 = mk3( q1.x, q1.y, q1.z ), v1v3 = mk3( q1.w, q2.x, q2.y );
        V3 c      = cross3( dir, v1v3 );
        float det = dot3( v1v2, c );
        if ( det > -MRT_ACNE_EPS && det < MRT_ACNE_EPS ) return false;
        float inv_det = 1.f / det;
        V3 sv         = origin - p1;
        float u       = dot3( sv, c ) * inv_det;
        if ( u < 0.f || u > 1.f ) return false;
        V3 q    = cross3( sv, v1v2 );
        float v = dot3( dir, q ) * inv_det;
        if ( v < 0.f || u + v > 1.f ) return false;
        float t = dot3( v1v3, q ) * inv_det;
        if ( t >= tmin && t <= tmax ) {
            tmax      = t;
            h.prim    = (int32_t)obj;
            h.t       = t;
            h.u       = u;
            h.v       = v;
            h.outside = !( det < 0.f );
            return true;
        }
        return false;
    }
    if ( kind == MRT_PRIM_SPHERE ) { // unit sphere in object space
        V3 oc       = origin;
        float a     = dot3( dir, dir );
        float b     = 2.f * dot3( dir, oc );
        float c     = dot3( oc, oc ) - 1.f;
        float delta = b * b - 4.f * a * c;
        float t;
        if ( delta == 0 ) {
            t = -b / ( 2.f * a );
            if ( !( t >= tmin && t <= tmax ) ) return false;
            // the one-root branch takes its normal from the world hit point mapped back
            V3 pos         = wo + ( t * wd );
            const DXf& xf  = s.xforms[flags >> MRT_PRIM_XF_SHIFT];
            V3 objectPoint = ( flags >> MRT_PRIM_XF_SHIFT ) == 0 ? pos : mat_mul( xf.inv, pos - mk3( xf.t[0], xf.t[1], xf.t[2] ) );
            h.outside      = dot3( dir, objectPoint ) < 0;
        }
        else if ( delta > 0 ) {
            float sq = sqrtf( delta );
            float t1 = ( -b + sq ) / ( 2.f * a );
            float t2 = ( -b - sq ) / ( 2.f * a );
            bool in1 = t1 >= tmin && t1 < tmax, in2 = t2 >= tmin && t2 < tmax;
            if ( in1 && in2 ) t = ( t2 < t1 ) ? t2 : t1;
            else if ( in1 ) t = t1;
            else if ( in2 ) t = t2;
            else return false;
            V3 objectNormal = normalize3( origin + ( t * dir ) );
            h.outside       = dot3( dir, objectNormal ) < 0;
        }
        else { return false; }
        tmax   = t;
        h.prim = (int32_t)obj;
        h.t    = t;
        h.u = h.v = 0.f;
        return true;
    }
    // plane: unit square z = 0, |x|,|y| <= 1 in object space; ignores tmin (plane.cpp:20)
    float t = -origin.z / dir.z;
    if ( t < 0 || t > tmax ) return false;
    V3 x = origin + t * dir;
    if ( x.x < -1 || x.x > 1 || x.y < -1 || x.y > 1 ) return false;
    tmax      = t;
    h.prim    = (int32_t)obj;
    h.t       = t;
    h.u       = ( 1.f + x.x ) * .5f; // Intersection::u, v of a plane hit
    h.v       = ( 1.f + x.y ) * .5f;
    h.outside = 1;
    return true;
}

template <bool ANY, bool STATS>
MRT_DEV bool trace_ray( const DTrace& s, V3 o, V3 d, float ray_tmin, float ray_tmax, DHit& out, DStats& st ) {
    bool found = false;
    XfRay xr;
    xr.id = 0xffffffffu;
    xr.o = xr.d = mk3( 0.f );
    const float orig_tmin = ray_tmin, orig_tmax = ray_tmax;
    out.prim = -1;
    out.t    = ray_tmax;
    out.u = out.v = 0.f;
    out.outside   = 0;

    // ---- intersectAabb on the root box (kdtree.cpp:162-180)
    bool in_box = false;
    if ( s.n_nodes > 0 ) {
        const float ix = 1.f / d.x, iy = 1.f / d.y, iz = 1.f / d.z;
        const bool sx = !( d.x >= 0 ), sy = !( d.y >= 0 ), sz = !( d.z >= 0 );
        float tmin  = ( ( sx ? s.root_max[0] : s.root_min[0] ) - o.x ) * ix;
        float tmax  = ( ( sx ? s.root_min[0] : s.root_max[0] ) - o.x ) * ix;
        float tymin = ( ( sy ? s.root_max[1] : s.root_min[1] ) - o.y ) * iy;
        float tymax = ( ( sy ? s.root_min[1] : s.root_max[1] ) - o.y ) * iy;
        in_box      = !( ( tmin > tymax ) || ( tymin > tmax ) );
        if ( in_box ) {
            if ( tymin > tmin ) tmin = tymin;
            if ( tymax < tmax ) tmax = tymax;
            float tzmin = ( ( sz ? s.root_max[2] : s.root_min[2] ) - o.z ) * iz;
            float tzmax = ( ( sz ? s.root_min[2] : s.root_max[2] ) - o.z ) * iz;
            in_box      = !( ( tmin > tzmax ) || ( tzmin > tmax ) );
            if ( in_box ) {
                if ( tzmin > tmin ) tmin = tzmin;
                if ( tzmax < tmax ) tmax = tzmax;
                if ( tmin > ray_tmin ) ray_tmin = tmin;
                if ( tmax < ray_tmax ) ray_tmax = tmax;
            }
        }
    }

    // ---- traverse (kdtree.cpp:467-553)
    if ( in_box ) {
        uint2 stack[MRT_STACK_MAX];
        int sp        = 0;
        float a       = ray_tmin, b = ray_tmax;
        uint32_t node = 0;
        while ( true ) {
            uint2 nd = s.nodes[node];
            while ( ( nd.y & 3u ) != 3u ) {
                if ( STATS ) st.inner++;
                const uint32_t axis = nd.y & 3u;
                const float split   = __uint_as_float( nd.x );
                const uint32_t left = nd.y >> 2;
                const float diff    = split - comp( o, axis );
                const float t       = diff / comp( d, axis );
                const bool lnear    = diff > 0.0f;
                const uint32_t nr   = lnear ? left : left + 1;
                const uint32_t fr   = lnear ? left + 1 : left;
                if ( ( t > b ) || ( t < 0.0f ) ) { node = nr; }
                else if ( t < a ) { node = fr; }
                else {
                    stack[sp++] = make_uint2( fr, __float_as_uint( b ) );
                    node        = nr;
                    b           = t;
                }
                nd = s.nodes[node];
            }
            if ( STATS ) st.leaves++;
            // leaf: interval becomes [a, b]
            float tmax         = b;
            const uint32_t off = nd.x, cnt = nd.y >> 2;
            bool has           = false;
            float dist         = 0.f;
            for ( uint32_t i = 0; i < cnt; i++ ) {
                const uint32_t obj = s.refs[off + i];
                if ( STATS ) {
                    st.refs++;
                    st.tests++;
                }
                DHit temp;
                if ( prim_intersect( s, obj, o, d, a, tmax, xr, temp ) ) {
                    if ( ANY ) {
                        const uint32_t flags = __float_as_uint( s.prims[3 * (size_t)obj].w );
                        if ( !( flags & MRT_PRIM_EMISSIVE ) ) {
                            out = temp;
                            return true;
                        }
                    }
                    else {
                        const float temp_dist = tmax;
                        if ( has ) {
                            if ( temp_dist < dist ) {
                                dist = temp_dist;
                                out  = temp;
                            }
                        }
                        else {
                            has  = true;
                            out  = temp;
                            dist = temp_dist;
                        }
                    }
                }
            }
            if ( has ) {
                found = true;
                break;
            }
            if ( sp == 0 ) break;
            --sp;
            node = stack[sp].x;
            a    = b;
            b    = __uint_as_float( stack[sp].y );
        }
    }

    // ---- out-of-tree objects (kdtree.cpp:563-611): the backup ray is a closest-hit ray with the
    // original interval
    if ( s.n_out > 0 ) {
        float btmax = found ? out.t : orig_tmax;
        float dist  = found ? out.t : 0.f;
        for ( uint32_t i = 0; i < s.n_out; i++ ) {
            const uint32_t obj = s.out_of_tree[i];
            if ( STATS ) st.tests++;
            DHit temp;
            if ( prim_intersect( s, obj, o, d, orig_tmin, btmax, xr, temp ) ) {
                const float temp_dist = btmax;
                if ( found ) {
                    if ( temp_dist < dist ) {
                        dist = temp_dist;
                        out  = temp;
                    }
                }
                else {
                    found = true;
                    out   = temp;
                    dist  = temp_dist;
                    if ( ANY ) return true;
                }
            }
        }
    }
    if ( !found ) out.prim = -1;
    return found;
}
