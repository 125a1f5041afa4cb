// kd-tree traversal and ray/primitive intersection for the gfx950 kernels.
//
// The algorithm, its comparisons and its arithmetic are those of the reference's intersectKdTree
// (kdtree.cpp:162-180, 467-613; shapes/triangle.cpp:4-40, shapes/sphere.cpp:52-133,
// shapes/plane.cpp:6-30), so the primitive id, t and barycentrics of every hit are bit-identical
// to the CPU path:
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
//   - planes live outside the tree.  The reference tests them after the tree (kdtree.cpp:588-611), in
//     list order, each accepted plane shrinking the interval for the next, as a closest-hit ray over
//     [0, D]: D is the tree hit distance if there was one -- even where that lies beyond the ray's own
//     tmax, which happens after a split distance came out NaN (an axis-parallel ray that starts in
//     the split plane: 0 / 0) -- and the ray's own tmax otherwise; a plane replaces the tree hit only
//     if strictly nearer.  The outcome is the first of the nearest planes if that one lies within D,
//     so here the planes are tested when a ray is fetched (a closest-hit ray: the nearest plane at
//     any distance, the first one on ties; an any-hit ray, which never reaches the planes once the
//     tree produced an occluder: the first plane accepted within its own interval) and D is applied
//     when the ray retires.  (One case is not reproduced: a plane whose distance is NaN listed after
//     a plane beyond D is a hit for the reference and none here.)
//
// Execution model: streaming traversal.  The waves draw chunks (MRT_CHUNK rays) of the ray list from
// counters (one per eighth of the list, see trace_stream) and every
// lane is a small state machine (idle / node / primitive test).  One trip of
// the wave's loop performs, for every lane, the work that depends on the loads issued in the trip
// before and issues the next loads, so a trip costs one memory round trip however the lanes'
// states are mixed; lanes whose ray retired are refilled from the wave's current chunk once enough
// of them are idle.  Rays differ in length by orders of magnitude, so without the refill a wave runs
// at the pace of its longest ray with most lanes idle.
// What a trip costs, measured: the first-level cache look-ups its lanes make (a divergent load is one
// look-up per lane) and its scalar instructions (three times as dear as vector instructions: masks of
// per-lane flags, branch bookkeeping) -- hence leaf references fetched four to a load, the mailbox of
// triangles a ray is known to miss, and flags that are not assigned under divergent branches.
//
// The deferred-node stack stores (node, tmax) only: the entry interval of a popped node is always
// the exit value b of the leaf visited just before it (the same float, carried by assignment),
// so it is not stored.  A deferred node is named by the record that holds it: bits 0-29 the
// record, bit 31 set when it is one of the record's children (bit 30: the right one) rather than
// the record's head.
#pragma once
#include "device_math.hpp"

#include <cstdlib>
#include <mutex>
#include <type_traits>

#ifndef MRT_STACK_MAX
#define MRT_STACK_MAX 128 // deferred nodes per ray: trees up to depth 126 (the entries beyond the LDS part live in scratch)
#endif
#define MRT_ACNE_EPS 1e-4f
#ifndef MRT_XF_LDS
#define MRT_XF_LDS 16 // transforms kept in LDS (12 floats each); the rest are read from HBM
#endif
#ifndef MRT_REFILL_MIN
#define MRT_REFILL_MIN 16 // idle lanes that trigger a refill of the wave
#endif
// per ray source (the IO type of trace_stream): specialised where a source wants another threshold
template <class IO>
struct RefillMin {
    static constexpr uint32_t value = MRT_REFILL_MIN;
};

// Per ray source: how many primitives a ray remembers to have missed (see trace_stream); 0 = none.
#ifndef MRT_MAILBOX
#define MRT_MAILBOX 8
#endif
#ifndef MRT_MAILBOX_SKIPS
#define MRT_MAILBOX_SKIPS 1
#endif
template <class IO>
struct Mailbox {
    static constexpr int size = MRT_MAILBOX;
};

// prim record: ten words, three records to a 128-byte line (slots at bytes 0, 40, 80; a record never
// straddles two cache lines, and a line fetched for one primitive of a leaf brings two neighbours):
//   words 0-3 = p1.xyz , flags      flags: bits 0-1 kind (0 triangle, 1 sphere, 2 plane),
//   words 4-7 = e1.xyz , e2.x              bit 2 emission != (0,0,0) component-wise (kdtree.cpp:533-534),
//   words 8-9 = e2.yz                      bit 3 emission.r+g+b != 0 (isBlack, integrator.cpp:110),
//                                          bits 4-5 shade kind of the object's material, bits 6.. transform index (0 = identity)
//   e1 = p2 - p1, e2 = p3 - p1 (fp32, as triangle.cpp:9-10)
// The traversal names a primitive by where its record lies: line << 2 | slot (prim_loc); that is what
// the tree records and the leaves' lists carry, and a ray's hit is turned back into the object index
// when the ray retires.  (MRT_PRIM_PACK 0: 64-byte slots, two to a line, named by the index itself.)
// What shading needs of an object -- (flags, material) -- lives in DTrace::prim_info.
// Device tree: 32-byte records, one per node of even depth (the root has depth 0), holding that
// node and its two children, so one fetch serves two levels of the descent:
//   q0 = head split bits, head axis (0..2; 3 = no head step, the left slot IS the node: a leaf of
//        even depth), left child w0, left child w1
//   q1 = right child w0, right child w1, first reference of the left child, of the right child
// A child is an inner node (split bits, axis | record << 2: `record` holds its left child and
// record + 1 its right child) or a leaf (w0, 3 | count << 2); a leaf's first reference
// arrives with the record, the further ones come from the reference list one trip ahead: words
// w0 + 1 .. w0 + count - 1 of DTrace::refs.  By default the lists live in the same pool of 32-byte
// slots as the records, each right behind the pair of sibling records that holds its leaf
// (build_tree_records), so refs points at the pool and record numbers count list slots too.
#ifndef MRT_PRIM_PACK
#define MRT_PRIM_PACK 1
#endif
#define MRT_PRIM_TRI 0u
#define MRT_PRIM_SPHERE 1u
#define MRT_PRIM_PLANE 2u
#define MRT_PRIM_EMISSIVE 4u
#define MRT_PRIM_EMISSIVE_SUM 8u
#define MRT_PRIM_SHADE_SHIFT 4 // bits 4-5: what shade does with a hit on the object (MRT_SHADE_*)
#define MRT_SHADE_PLAIN 0u        //   Cook-Torrance, constant albedo
#define MRT_SHADE_TEXTURED 1u     //   Cook-Torrance, albedo from a texture
#define MRT_SHADE_TRANSPARENT 2u  //   smooth dielectric
#define MRT_PRIM_XF_SHIFT 6

struct DTrace {
    const uint4* nodes; // two uint4 per record
    const uint32_t* refs; // the leaves' lists: the pool of records itself, or an array of their own
    const float4* prims;     // the records the traversal reads (layout above)
    const uint2* prim_info;  // per object: (flags, material)
    const DXf* xforms;
    const uint32_t* out_of_tree;
    uint32_t n_nodes;
    uint32_t n_records; // 32-byte slots in `nodes` (at least 1 when n_nodes != 0)
    uint32_t n_out;
    uint32_t n_xforms;
    float root_min[3];
    float root_max[3];
    uint32_t root[8]; // record 0: every ray starts from it without a load
    // workgroups go to the device's XCDs round-robin: blockIdx.x & xcd_mask names the XCD a block
    // runs on (7 on an unpartitioned MI355X; probed per device at scene creation, 0 = unknown
    // mapping or a single XCD)
    uint32_t xcd_mask;
    // rays remember the triangles they were found to miss (trace_stream: mailbox); 0 for measurements and tests
    uint32_t mailbox_on;
};

struct DHit {
    int32_t prim;
    float t, u, v;
    uint32_t outside; // Intersection::isOutside
};

struct DStats {
    uint32_t inner, leaves, refs, tests;
};
#ifdef MRT_TRIP_HISTOGRAM
// measurement aid: how many trips of the wave's loop the rays stayed in flight, in powers of two
// (bucket k: 2^k <= trips < 2^(k+1)); read and cleared by mrt_debug_trip_histogram
__device__ unsigned long long g_trip_hist[32];
#endif

// LDS of a traversal block (MRT_TRACE_BLOCK lanes).  Everything a lane touches rarely lives here
// instead of in registers, slot-major so that the lanes of a wave fall on distinct banks:
//   transform table   inv (9) + translation (3) of the first MRT_XF_LDS transforms
//   stack             the first deferred nodes of every lane (deeper ones in scratch, and every
//                     scratch access waits for all the gathers in flight: the more entries fit
//                     here the better)
//   hit record        the accepted hit of the current leaf (prim | outside << 31, u, v); until
//                     there is one, the u slot keeps the tmax the ray came with, which is what a
//                     miss reports as t
//   plane record      the plane hit found when the ray was fetched (prim, t, u, v)
// A traversal that only answers "occluded or not" (BOOL_ONLY) keeps no records and gives their
// space to the stack.  Sized for six blocks per CU (160 KB / 6 = 6826 words).
#ifndef MRT_STACK_LDS
#define MRT_STACK_LDS 9
#endif
#ifndef MRT_STACK_LDS_BOOL
#define MRT_STACK_LDS_BOOL 12
#endif
#define MRT_TRACE_BLOCK 256
#ifndef MRT_TRACE_WAVES
#define MRT_TRACE_WAVES 6 // waves per SIMD the traversal kernels are compiled for (register budget)
#endif
// MRT_TOP_LDS: tree records 1 .. MRT_TOP_LDS (the levels right below the root record, which every
// ray steps through; the caller's breadth-first top layout numbers them first) staged in LDS, 8 words
// each, and read from there instead of through the vector memory path.  0 = none.
#ifndef MRT_TOP_LDS
#define MRT_TOP_LDS 0
#endif
// waves per SIMD the occlusion-only kernels (which keep no hit record and fewer registers) are compiled for
#ifndef MRT_OCCLUSION_WAVES
#define MRT_OCCLUSION_WAVES MRT_TRACE_WAVES
#endif
#define MRT_TRACE_LDS_WORDS( BOOL_ONLY ) \
    ( MRT_XF_LDS * 12 + MRT_TOP_LDS * 8 + ( ( BOOL_ONLY ) ? MRT_STACK_LDS_BOOL * MRT_TRACE_BLOCK * 2 : MRT_STACK_LDS * MRT_TRACE_BLOCK * 2 + MRT_TRACE_BLOCK * 7 ) )

typedef __attribute__( ( address_space( 3 ) ) ) uint32_t lds_u32;
typedef __attribute__( ( address_space( 3 ) ) ) float lds_f32;
typedef float v4f __attribute__( ( ext_vector_type( 4 ) ) );
typedef uint32_t v2u __attribute__( ( ext_vector_type( 2 ) ) );
typedef float v2f __attribute__( ( ext_vector_type( 2 ) ) );
typedef __attribute__( ( address_space( 3 ) ) ) v4f lds_f4;
typedef __attribute__( ( address_space( 3 ) ) ) v2f lds_f2;
typedef __attribute__( ( address_space( 3 ) ) ) v2u lds_u2;
MRT_DEV v4f mk4( float x, float y, float z, float w ) {
    v4f r;
    r.x = x, r.y = y, r.z = z, r.w = w;
    return r;
}

// Copies inv (9 floats) and translation (3 floats) of the first MRT_XF_LDS transforms into LDS.
// Every thread of the block must call it; it ends with a barrier.
MRT_DEV void stage_transforms( const DTrace& s, lds_f32* xf_lds ) {
    const uint32_t n = ( s.n_xforms < MRT_XF_LDS ? s.n_xforms : MRT_XF_LDS ) * 12u;
    for ( uint32_t i = threadIdx.x; i < n; i += blockDim.x ) {
        const uint32_t x = i / 12u, k = i - x * 12u;
        xf_lds[i]        = k < 9u ? s.xforms[x].inv[k] : s.xforms[x].t[k - 9u];
    }
    __syncthreads();
}

// Object::transformRay (Object.cpp:8-11): o' = inv * (o - T), d' = inv * d
MRT_DEV void transform_ray( const DTrace& s, const lds_f32* xf_lds, uint32_t xf, V3 wo, V3 wd, V3& oo, V3& od ) {
    float inv[9], t[3];
    if ( xf < MRT_XF_LDS ) {
        const lds_f32* p = xf_lds + xf * 12u;
        for ( int k = 0; k < 9; k++ )
            inv[k] = p[k];
        t[0] = p[9], t[1] = p[10], t[2] = p[11];
    }
    else { // beyond the staged table: straight from HBM (four 16-byte loads, unlike the LDS path)
        const float4* g = (const float4*)&s.xforms[xf];
        const float4 g0 = g[0], g1 = g[1], g2 = g[2];
        inv[0] = g0.x, inv[1] = g0.y, inv[2] = g0.z, inv[3] = g0.w;
        inv[4] = g1.x, inv[5] = g1.y, inv[6] = g1.z, inv[7] = g1.w;
        inv[8] = g2.x, t[0] = g2.y, t[1] = g2.z, t[2] = g2.w;
    }
    oo = mat_mul( inv, wo - mk3( t[0], t[1], t[2] ) );
    od = mat_mul( inv, wd );
}

// Möller-Trumbore as shapes/triangle.cpp:4-40 evaluates it.  Returns true when the hit is accepted
// into [tmin, tmax]; det is returned for the sidedness.
// `apart` reports a miss that does not depend on [tmin, tmax]: the ray's line misses the triangle.
MRT_DEV bool tri_test( V3 origin, V3 dir, float4 q0, float4 q1, float2 q2, float tmin, float tmax, float& t, float& u, float& v, float& det, bool& apart ) {
    // Evaluated without early exits: in a wave whose lanes test unrelated triangles some lane
    // always needs every term, and straight-line code spares the execution-mask bookkeeping of
    // four nested returns.  The accept condition is the reference's chain of rejections, each
    // negated as written (so a NaN passes the same tests it passes there).
    const V3 p1 = mk3( q0.x, q0.y, q0.z ), v1v2 = mk3( q1.x, q1.y, q1.z ), v1v3 = mk3( q1.w, q2.x, q2.y );
    const V3 c  = cross3( dir, v1v3 );
    det         = dot3( v1v2, c );
    const float inv_det = 1.f / det;
    const V3 sv         = origin - p1;
    u                   = dot3( sv, c ) * inv_det;
    const V3 q          = cross3( sv, v1v2 );
    v                   = dot3( dir, q ) * inv_det;
    t                   = dot3( v1v3, q ) * inv_det;
    const bool flat = ( det > -MRT_ACNE_EPS ) & ( det < MRT_ACNE_EPS );
    const bool u_out = ( u < 0.f ) | ( u > 1.f );
    const bool v_out = ( v < 0.f ) | ( u + v > 1.f );
    const bool t_in  = ( t >= tmin ) & ( t <= tmax );
    apart = flat | u_out | v_out;
    return !apart & t_in;
}
MRT_DEV bool tri_test( V3 origin, V3 dir, float4 q0, float4 q1, float2 q2, float tmin, float tmax, float& t, float& u, float& v, float& det ) {
    bool apart;
    return tri_test( origin, dir, q0, q1, q2, tmin, tmax, t, u, v, det, apart );
}

// Unit sphere in object space (shapes/sphere.cpp:52-133).  `wo`, `wd`: world ray (the one-root
// branch maps the world hit point back through the inverse transform).
MRT_DEV bool sphere_test( const DTrace& s, const lds_f32* xf_lds, uint32_t xf, V3 origin, V3 dir, V3 wo, V3 wd, float tmin, float tmax, float& t, bool& outside ) {
    V3 oc       = origin;
    float a     = dot3( dir, dir );
    float b     = 2.f * dot3( dir, oc );
    float c     = dot3( oc, oc ) - 1.f;
    float delta = b * b - 4.f * a * c;
    if ( delta == 0 ) {
        t = -b / ( 2.f * a );
        if ( !( t >= tmin && t <= tmax ) ) return false;
        V3 pos         = wo + ( t * wd );
        V3 objectPoint = pos;
        if ( xf != 0 ) {
            V3 unused;
            transform_ray( s, xf_lds, xf, pos, mk3( 0.f ), objectPoint, unused );
        }
        outside = dot3( dir, objectPoint ) < 0;
        return true;
    }
    if ( delta > 0 ) {
        float sq = sqrtf( delta );
        float t1 = ( -b + sq ) / ( 2.f * a );
        float t2 = ( -b - sq ) / ( 2.f * a );
        bool in1 = t1 >= tmin && t1 < tmax, in2 = t2 >= tmin && t2 < tmax;
        if ( in1 && in2 ) t = ( t2 < t1 ) ? t2 : t1;
        else if ( in1 ) t = t1;
        else if ( in2 ) t = t2;
        else return false;
        V3 objectNormal = normalize3( origin + ( t * dir ) );
        outside         = dot3( dir, objectNormal ) < 0;
        return true;
    }
    return false;
}

// Unit square z = 0, |x|,|y| <= 1 in object space; ignores tmin (shapes/plane.cpp:6-30).
MRT_DEV bool plane_test( V3 origin, V3 dir, float tmax, float& t, float& u, float& v ) {
    t = -origin.z / dir.z;
    if ( t < 0 || t > tmax ) return false;
    V3 x = origin + t * dir;
    if ( x.x < -1 || x.x > 1 || x.y < -1 || x.y > 1 ) return false;
    u = ( 1.f + x.x ) * .5f; // Intersection::u, v of a plane hit
    v = ( 1.f + x.y ) * .5f;
    return true;
}

// intersectAabb (kdtree.cpp:162-180) on the box [bmin, bmax]: clips [ray_tmin, ray_tmax], false on a miss.
MRT_DEV bool clip_box( const float* bmin, const float* bmax, V3 o, V3 d, float& ray_tmin, float& ray_tmax ) {
    const float ix = 1.f / d.x, iy = 1.f / d.y, iz = 1.f / d.z;
    const bool sx = !( d.x >= 0 ), sy = !( d.y >= 0 ), sz = !( d.z >= 0 );
    float tmin  = ( ( sx ? bmax[0] : bmin[0] ) - o.x ) * ix;
    float tmax  = ( ( sx ? bmin[0] : bmax[0] ) - o.x ) * ix;
    float tymin = ( ( sy ? bmax[1] : bmin[1] ) - o.y ) * iy;
    float tymax = ( ( sy ? bmin[1] : bmax[1] ) - o.y ) * iy;
    if ( ( tmin > tymax ) || ( tymin > tmax ) ) return false;
    if ( tymin > tmin ) tmin = tymin;
    if ( tymax < tmax ) tmax = tymax;
    float tzmin = ( ( sz ? bmax[2] : bmin[2] ) - o.z ) * iz;
    float tzmax = ( ( sz ? bmin[2] : bmax[2] ) - o.z ) * iz;
    if ( ( tmin > tzmax ) || ( tzmin > tmax ) ) return false;
    if ( tzmin > tmin ) tmin = tzmin;
    if ( tzmax < tmax ) tmax = tzmax;
    if ( tmin > ray_tmin ) ray_tmin = tmin;
    if ( tmax < ray_tmax ) ray_tmax = tmax;
    return true;
}
// The same on the (padded) root box of the tree.
MRT_DEV bool clip_root( const DTrace& s, V3 o, V3 d, float& ray_tmin, float& ray_tmax ) {
    if ( s.n_nodes == 0 ) return false;
    return clip_box( s.root_min, s.root_max, o, d, ray_tmin, ray_tmax );
}

// Lane states of the streaming traversal.
#define MRT_ST_IDLE 0u  // no ray
#define MRT_ST_INNER 1u // node record in flight
#define MRT_ST_TEST 3u  // record of the current primitive in flight
#define MRT_NO_PRIM 0xffffffffu

// Byte-offset loads: tables are addressed as base + 32-bit byte offset (checked at scene creation),
// which the hardware forms from a scalar base and one vector register.
MRT_DEV float4 ld_node( const DTrace& s, uint32_t rec, uint32_t half ) {
    return *(const float4*)( (const char*)s.nodes + ( ( rec << 5 ) + half * 16u ) );
}
MRT_DEV uint32_t ld_ref( const DTrace& s, uint32_t i ) {
    return *(const uint32_t*)( (const char*)s.refs + ( i << 2 ) );
}
// four consecutive references with one load instruction (the lists are word-aligned; what lies behind a
// list's end -- the next list, padding -- is read and not used: the tables end with 16 spare bytes)
typedef uint32_t v4u_a4 __attribute__( ( ext_vector_type( 4 ), aligned( 4 ) ) );
MRT_DEV v4u_a4 ld_ref4( const DTrace& s, uint32_t i ) {
    return *(const v4u_a4*)( (const char*)s.refs + ( i << 2 ) );
}
// primitive records: where the record of object `idx` lies, and back
#define MRT_HD __host__ __device__ __forceinline__
MRT_HD uint32_t prim_loc( uint32_t idx ) {
#if MRT_PRIM_PACK
    return ( ( idx / 3u ) << 2 ) | ( idx % 3u );
#else
    return idx;
#endif
}
MRT_HD uint32_t prim_index( uint32_t loc ) {
#if MRT_PRIM_PACK
    return ( loc >> 2 ) * 3u + ( loc & 3u );
#else
    return loc;
#endif
}
MRT_HD uint32_t prim_offset( uint32_t loc ) { // in bytes
#if MRT_PRIM_PACK
    return ( ( loc >> 2 ) << 7 ) + ( loc & 3u ) * 40u;
#else
    return loc << 6;
#endif
}
typedef float v4f_a8 __attribute__( ( ext_vector_type( 4 ), aligned( 8 ) ) ); // (a record's slots are 8-byte aligned)
// the first two quarters of a record (k = 0, 1) and its last two words
MRT_DEV float4 ld_prim( const DTrace& s, uint32_t loc, uint32_t k ) {
    const v4f_a8 r = *(const v4f_a8*)( (const char*)s.prims + ( prim_offset( loc ) + k * 16u ) );
    return make_float4( r.x, r.y, r.z, r.w );
}
MRT_DEV float2 ld_prim_tail( const DTrace& s, uint32_t loc ) {
    return *(const float2*)( (const char*)s.prims + ( prim_offset( loc ) + 32u ) );
}

// The out-of-tree planes of a ray (kdtree.cpp:563-611), tested when the ray is fetched: an any-hit ray
// stops at the first plane accepted within its interval; a closest-hit ray takes the nearest one,
// the first on ties (how far it may lie is only known when the ray retires: merge_plane).  Returns
// whether a plane was hit, its record (prim, t, u, v) in `rec`, and in `tested` the number of planes
// the reference would have tested.
template <bool ANY>
MRT_DEV bool scan_planes( const DTrace& s, const lds_f32* xf_lds, V3 o, V3 d, float ray_tmax, v4f& rec, uint32_t& tested ) {
    bool has_pl = false;
    float btmax = ANY ? ray_tmax : __uint_as_float( 0x7f800000u );
    tested      = s.n_out;
    rec         = mk4( 0.f, 0.f, 0.f, 0.f );
    for ( uint32_t k = 0; k < s.n_out; k++ ) {
        const uint32_t po = s.out_of_tree[k];
        const uint32_t xf = s.prim_info[po].x >> MRT_PRIM_XF_SHIFT;
        V3 oo = o, od = d;
        if ( xf != 0 ) transform_ray( s, xf_lds, xf, o, d, oo, od );
        float t, u, v;
        if ( plane_test( oo, od, btmax, t, u, v ) ) {
            btmax = t; // accepted: later planes must be at least as near,
            // and replace this one only if strictly nearer
            if ( !has_pl || t < rec.y ) rec = mk4( __uint_as_float( po ), t, u, v );
            has_pl = true;
            if ( ANY ) {
                tested = k + 1;
                break;
            }
        }
    }
    return has_pl;
}

// What a ray reports when it leaves the tree with (`has`: hit record in h, h.t = its distance) or
// without a tree hit (h.prim = -1, h.t = the tmax the ray came with), given the plane record of
// scan_planes.  A closest-hit ray: the plane must lie within the tree hit distance or, without a tree
// hit, within the ray's own tmax (h.t in both cases), and replaces a tree hit only if strictly
// nearer.  An any-hit ray: only without a tree hit.  Returns `found`.
template <bool ANY>
MRT_DEV bool merge_plane( bool has, DHit& h, bool has_pl, v4f r ) {
    bool found = has;
    if ( has_pl ) {
        if ( ANY ? !has : ( has ? r.y < h.t : !( r.y > h.t ) ) ) {
            h.prim = (int32_t)__float_as_uint( r.x ), h.t = r.y, h.u = r.z, h.v = r.w, h.outside = 1;
            found  = true;
        }
    }
    return found;
}

// Traces rays 0..n-1 of `io` with the whole grid.
//   io.load(i, tag, o, d, tmin, tmax)  fetches ray i; `tag` is handed back to store
//   io.store(tag, found, hit)          consumes the result (hit.prim = -1 when nothing was hit)
// ANY: shadow-ray semantics.  BOOL_ONLY (with ANY): only `found` is meaningful, so a ray that hits
// a plane retires without entering the tree.  Every thread of the grid must call this (it uses
// wave-wide votes and a block barrier); blockDim.x must be MRT_TRACE_BLOCK; `lds` must hold
// MRT_TRACE_LDS_WORDS( BOOL_ONLY ) 32-bit words.
//
// Dealing: the list is cut into runs of MRT_DEAL_RUN chunks of MRT_CHUNK rays, dealt round-robin to eight
// parts, one per XCD (workgroups go to the XCDs round-robin, so blockIdx.x & DTrace::xcd_mask names the XCD), each
// with a counter in `deal` (eight counters MRT_DEAL_STRIDE words apart, zero at launch).  A wave
// draws the next chunk of its XCD's part, and of the following parts once that one is exhausted.
// The rays in flight on one XCD are therefore consecutive in the list - for path tracing:
// neighbouring pixels - and share the tree nodes and primitives they pull through that XCD's L2;
// the eight XCDs work on neighbouring runs, so what one of them brought into the memory-side cache
// serves the others; and no wave is left with a longer share than the others.
#ifndef MRT_CHUNK
#define MRT_CHUNK 128u // rays a wave draws at a time
#endif
#ifndef MRT_DEAL_RUN
#define MRT_DEAL_RUN 128u
#endif
#define MRT_DEAL_STRIDE 32u
#define MRT_DEAL_WORDS ( 8u * MRT_DEAL_STRIDE )
// Relay of long rays.  A launch lasts as long as its longest ray, and one ray in some thousands takes
// thousands of trips (it runs along a surface through hundreds of small leaves) where the median takes
// fifty: the waves that hold such rays keep the launch alive for milliseconds after the list ran out.
// So a ray that is still in flight `after` trips after its wave found the list exhausted is handed over
// the next time it has finished a leaf without a hit: (tag, exit distance of that leaf) goes to `list`.
// Everything in front of that distance has been searched, so what remains is the same ray restarted from
// the root with that distance as its tmin; a follow-up launch (relay_trace_kernel) cuts the remaining
// interval into MRT_RELAY_K segments, traces the segments as rays of their own, side by side, and
// relay_finish_kernel takes the first segment that found something.  Same result: the reference accepts
// a hit where its distance lies within the (closed) interval the ray spends in a leaf, those intervals
// follow one another along the ray, and the segments cut them at common points, so the first leaf with
// an accepted hit, and the nearest accepted hit in it, are found by the first segment that reaches
// into that leaf's part in front of and including the hit -- except where arithmetic on NaN decides (a
// ray with a zero or non-finite component is never handed over).  For an any-hit ray the same holds
// for WHETHER it is occluded, provided the scene has no emissive object (ignored hits on those shrink
// the interval for the rest of the leaf), but not for WHICH occluder it reports (the first accepted
// in the leaf's list order, which a cut through the leaf changes): the host enables the relay for
// any-hit launches only where they answer "occluded or not" in a scene without emissive objects.
// list == nullptr: no relay.
#ifndef MRT_RELAY_K
#define MRT_RELAY_K 64u // segments per relayed ray (a power of two)
#endif
struct Relay {
    uint2* list;     // (tag, bits of the distance reached)
    uint32_t* count; // entries appended (may run past cap: the surplus was not handed over)
    uint32_t cap;
    uint32_t after;  // trips a wave goes on after the list ran out before it hands its rays over
};

template <bool ANY, bool BOOL_ONLY, bool STATS, class IO>
MRT_DEV void trace_stream( const DTrace& s, lds_u32* lds, const IO& io, uint32_t n, uint32_t* deal, DStats& st, const Relay relay = Relay { nullptr, nullptr, 0u, 0u } ) {
    constexpr int NST          = BOOL_ONLY ? MRT_STACK_LDS_BOOL : MRT_STACK_LDS;
    constexpr uint32_t LDS_TOP = MRT_XF_LDS * 12, LDS_STACK = LDS_TOP + MRT_TOP_LDS * 8, LDS_OUT = LDS_STACK + NST * MRT_TRACE_BLOCK * 2, LDS_PL = LDS_OUT + MRT_TRACE_BLOCK * 3;
    const lds_f32* xf_lds = (const lds_f32*)lds;
#if MRT_TOP_LDS
    {
        // records 1 .. MRT_TOP_LDS as 8 words each (a tree with fewer records stages what it has)
        const uint32_t n_rec = ( s.n_records - 1u < (uint32_t)MRT_TOP_LDS ) ? s.n_records - 1u : (uint32_t)MRT_TOP_LDS;
        for ( uint32_t i = threadIdx.x; i < n_rec * 8u; i += blockDim.x )
            lds[LDS_TOP + i] = ( (const uint32_t*)s.nodes )[8u + i];
    }
    const lds_f4* top_lds = (const lds_f4*)( lds + LDS_TOP );
#endif
    stage_transforms( s, (lds_f32*)lds );
    lds_u2* stack_lds = (lds_u2*)( lds + LDS_STACK ) + threadIdx.x;
    // the records below exist only when !BOOL_ONLY
    lds_f2* out_lds   = (lds_f2*)( lds + LDS_OUT ) + threadIdx.x; // (prim bits, u)
    lds_f32* outv_lds = (lds_f32*)( lds + LDS_OUT + MRT_TRACE_BLOCK * 2 ) + threadIdx.x;
    lds_f4* pl_lds    = (lds_f4*)( lds + LDS_PL ) + threadIdx.x;
    V3 xro = mk3( 0.f ), xrd = mk3( 0.f ); // object-space ray under the lane's current transform
    auto put_xr = [&]( V3 ro, V3 rd ) { xro = ro, xrd = rd; };

    const uint32_t n_chunks = ( n + ( MRT_CHUNK - 1u ) ) / MRT_CHUNK;
    uint32_t part_try = 0;          // parts found exhausted so far, counted from this XCD's own
    uint32_t chunk = 0, left = 0;   // first ray of the wave's current chunk, rays not yet taken from it
    bool more      = n_chunks != 0; // some part may still have chunks

    uint32_t state = MRT_ST_IDLE;
    uint32_t tag   = 0;
    V3 o = mk3( 0.f ), d = mk3( 0.f );
    float a = 0.f, b = 0.f, tmax = 0.f;
    // Load registers: L0, L1 are the tree record in state INNER; L0, L1, L2 the primitive
    // record in state TEST.
    float4 L0 = make_float4( 0.f, 0.f, 0.f, 0.f ), L1 = L0;
    float2 L2 = make_float2( 0.f, 0.f );
    uint32_t ptr = 0, end = 0, obj = 0;
    // the references that follow in the current leaf's list, fetched four at a time: nq0 is the one at
    // position ptr when a record goes in flight, and the queue covers positions [ptr, qhi)
    uint32_t nq0 = 0, nq1 = 0, nq2 = 0, nq3 = 0, qhi = 0;
    // Mailbox: the last MB triangles this ray's LINE was found to miss (tri_test: apart), newest first.
    // A kd-tree references a triangle from every leaf it overlaps (here dozens), and half of the tests a
    // ray makes are of a triangle it met a leaf or two earlier; a miss of that kind holds whatever the
    // interval, so the record is not fetched again and the test not repeated: the reference's outcome
    // (and its count of tests) without the look-ups.  obj == MRT_NO_PRIM: no record in flight this trip.
    constexpr int MB = Mailbox<IO>::size;
    static_assert( MB >= 0 && MB <= 16, "mailbox sizes: 0 .. 16" );
    uint32_t mb[MB ? MB : 1];
    auto mb_clear = [&]() {
#pragma unroll
        for ( int i = 0; i < MB; i++ )
            mb[i] = 0xffffffffu;
    };
    auto mb_has = [&]( uint32_t id ) -> bool {
        uint32_t x = 0xffffffffu;
#pragma unroll
        for ( int i = 0; i < MB; i++ )
            x = min( x, id ^ mb[i] );
        return x == 0u;
    };
    mb_clear();
    uint32_t cur  = 0; // the record in L0, L1
    uint32_t pick = 0; // after a pop: 2 = resume at the left child of the record, 3 = at the right one
    uint32_t xr_id = 0;    // transform of the object-space ray in LDS
    bool has       = false; // the current leaf has produced a hit (record in LDS, distance = tmax)
    bool has_pl    = false; // a plane was hit at fetch time (record in LDS)
    uint32_t pl_tests = 0;  // planes the reference would have tested for this ray (counters only)
    v2u stack_deep[MRT_STACK_MAX - NST];
    int sp = 0;
    auto push = [&]( uint32_t what, float upto ) {
        v2u e;
        e.x = what, e.y = __float_as_uint( upto );
        if ( sp < NST ) stack_lds[sp * MRT_TRACE_BLOCK] = e;
        else stack_deep[sp - NST] = e;
        sp++;
    };

#ifdef MRT_EXTRA_VALU
    float extra_valu = 1.f; // measurement aid: MRT_EXTRA_VALU more vector-ALU instructions per trip
#endif
#ifdef MRT_TRIP_HISTOGRAM
    uint32_t trips = 0;
#endif
    uint32_t tail_trips = 0; // trips since this wave found the list exhausted (wave-uniform)
    const uint32_t relay_after = relay.list ? relay.after : 0xffffffffu;
    // One trip of the wave's loop.  The loop exists twice: while the list may still have rays for this
    // wave (TAIL false: with the refill, never ends by itself), and after that (TAIL true: without the
    // refill, with the hand-over of long rays, returns false once every lane is idle).
    auto trip = [&]( auto tail_c ) -> bool {
        constexpr bool TAIL = decltype( tail_c )::value;
        if ( TAIL ) tail_trips++;
#ifdef MRT_TRIP_HISTOGRAM
        trips++;
#endif
#ifdef MRT_EXTRA_VALU
#pragma unroll
        for ( int i = 0; i < MRT_EXTRA_VALU; i++ )
            asm volatile( "v_add_f32 %0, %0, %0" : "+v"( extra_valu ) );
#endif
#ifdef MRT_EXTRA_SALU
#pragma unroll
        for ( int i = 0; i < MRT_EXTRA_SALU; i++ )
            asm volatile( "s_add_u32 s101, s101, 1" : : : "s101", "scc" ); // measurement aid: MRT_EXTRA_SALU more scalar-ALU instructions per trip
#endif
        // ---------------------------------------------------------------- refill
        const unsigned long long idle = __ballot( state == MRT_ST_IDLE );
        const uint32_t n_idle         = (uint32_t)__popcll( idle );
        if ( TAIL && n_idle == 64u ) return false;
        if ( !TAIL && n_idle >= RefillMin<IO>::value ) {
            while ( left == 0 && part_try < 8u ) {
                const uint32_t part = ( ( blockIdx.x & s.xcd_mask ) + part_try ) & 7u;
                uint32_t got        = 0;
                if ( __lane_id() == 0 ) got = atomicAdd( deal + part * MRT_DEAL_STRIDE, 1u );
                got                 = (uint32_t)__builtin_amdgcn_readfirstlane( (int)got );
                const uint32_t c    = ( ( got / MRT_DEAL_RUN ) * 8u + part ) * MRT_DEAL_RUN + got % MRT_DEAL_RUN;
                if ( c < n_chunks ) chunk = c * MRT_CHUNK, left = MRT_CHUNK;
                else part_try++;
            }
            more = left != 0;
            const uint32_t take = n_idle < left ? n_idle : left;
            if ( state == MRT_ST_IDLE ) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi( (uint32_t)( idle >> 32 ), __builtin_amdgcn_mbcnt_lo( (uint32_t)idle, 0u ) );
                const uint32_t idx  = chunk + ( MRT_CHUNK - left ) + rank;
                if ( rank < take && idx < n ) {
                    float ray_tmin, ray_tmax;
                    io.load( idx, tag, o, d, ray_tmin, ray_tmax );
#ifdef MRT_TRIP_HISTOGRAM
                    trips = 0;
#endif
                    xr_id    = 0;
                    mb_clear();
                    if ( !BOOL_ONLY ) {
                        v2f m;
                        m.x = __uint_as_float( 0xffffffffu ), m.y = ray_tmax;
                        *out_lds = m;
                    }
                    put_xr( o, d );
                    // planes: the logic of scan_planes, written out here with the record kept in LDS (calling
                    // the function, which keeps it in registers, made the whole frame 1.2 % slower)
                    has_pl      = false;
                    float btmax = ANY ? ray_tmax : __uint_as_float( 0x7f800000u );
                    if ( STATS ) pl_tests = s.n_out;
                    for ( uint32_t k = 0; k < s.n_out; k++ ) {
                        const uint32_t po = s.out_of_tree[k];
                        const uint32_t xf = s.prim_info[po].x >> MRT_PRIM_XF_SHIFT;
                        V3 oo = o, od = d;
                        if ( xf != 0 ) transform_ray( s, xf_lds, xf, o, d, oo, od );
                        float t, u, v;
                        if ( plane_test( oo, od, btmax, t, u, v ) ) {
                            btmax = t; // accepted: later planes must be at least as near,
                            // and replace this one only if strictly nearer
                            bool better = true;
                            if ( !BOOL_ONLY ) {
                                if ( has_pl ) {
                                    const v4f cur = *pl_lds;
                                    better        = t < cur.y;
                                }
                                if ( better ) *pl_lds = mk4( __uint_as_float( po ), t, u, v );
                            }
                            has_pl = true;
                            if ( ANY ) {
                                if ( STATS ) pl_tests = k + 1;
                                break;
                            }
                        }
                    }
                    if ( ANY && BOOL_ONLY && has_pl ) {
                        DHit h;
                        h.prim = 0, h.t = 0.f, h.u = h.v = 0.f, h.outside = 1;
                        io.store( tag, true, h );
                    }
                    // A ray with a NaN component takes the "visit both children" branch at every
                    // node and fails every primitive test: the reference walks the whole tree to
                    // find nothing.  Same result without the walk.
                    else if ( !( isnan( o.x ) | isnan( o.y ) | isnan( o.z ) | isnan( d.x ) | isnan( d.y ) | isnan( d.z ) ) && clip_root( s, o, d, ray_tmin, ray_tmax ) ) {
                        a     = ray_tmin;
                        b     = ray_tmax;
                        sp    = 0;
                        has   = false;
                        L0    = make_float4( __uint_as_float( s.root[0] ), __uint_as_float( s.root[1] ), __uint_as_float( s.root[2] ), __uint_as_float( s.root[3] ) );
                        L1    = make_float4( __uint_as_float( s.root[4] ), __uint_as_float( s.root[5] ), __uint_as_float( s.root[6] ), __uint_as_float( s.root[7] ) );
                        cur   = 0;
                        pick  = 0;
                        state = MRT_ST_INNER; // the root record is stepped through in this very trip
                    }
                    else {
                        if ( STATS ) st.tests += pl_tests;
                        DHit h;
                        h.prim = -1, h.t = ray_tmax, h.u = h.v = 0.f, h.outside = 0;
                        bool found = has_pl;
                        if ( !BOOL_ONLY && has_pl ) {
                            const v4f r = *pl_lds;
                            found       = ANY | !( r.y > ray_tmax ); // Plane::intersect rejects t > tmax
                            if ( found ) h.prim = (int32_t)__float_as_uint( r.x ), h.t = r.y, h.u = r.z, h.v = r.w, h.outside = 1;
                        }
                        io.store( tag, found, h );
                    }
                }
            }
            left -= take;
        }

        // ---------------------------------------------------------------- one step per lane
        // Compute first (this consumes the loads of the trip before), issue the next loads last.
        // Written with selects instead of nested branches: in a wave of unrelated rays every
        // branch is taken by some lane, so branching saves no arithmetic and costs scalar
        // execution-mask bookkeeping for every level of nesting.
        const bool is_inner = state == MRT_ST_INNER, is_test = state == MRT_ST_TEST;
        bool advance = false, retire = false, pop, want_node;
        uint32_t node;
        {
            // first level: the head of the record picks one of the two children it holds (after
            // a pop the child was picked when it was deferred)
            const uint32_t h_axis = __float_as_uint( L0.y ) & 3u;
            const bool forced     = ( pick != 0u ) | ( h_axis == 3u );
            const bool at_head    = is_inner & !forced;
            const float diff1     = L0.x - comp( o, h_axis );
            const float t1        = diff1 / comp( d, h_axis );
            const bool lnear1     = diff1 > 0.0f;
            const bool near_only1 = ( t1 > b ) | ( t1 < 0.0f );
            const bool far_only1  = !near_only1 & ( t1 < a );
            const bool both1      = at_head & !near_only1 & !far_only1;
            const bool right      = ( forced & ( ( pick & 1u ) != 0u ) ) | ( !forced & ( lnear1 == far_only1 ) ); // (no ?: -- it becomes a branch)
            if ( both1 ) push( cur | ( lnear1 ? 0xc0000000u : 0x80000000u ), b ); // far = right iff near = left
            b    = both1 ? t1 : b;
            pick = 0u;
            // second level: the child
            const float w0f    = right ? L1.x : L0.z;
            const uint32_t w0  = __float_as_uint( w0f );
            const uint32_t w1  = __float_as_uint( right ? L1.y : L0.w );
            const uint32_t r0  = __float_as_uint( right ? L1.w : L1.z );
            const bool at_node = is_inner & ( ( w1 & 3u ) != 3u ), at_leaf = is_inner & ( ( w1 & 3u ) == 3u );
            const uint32_t axis = w1 & 3u;
            const uint32_t left = w1 >> 2;
            const float diff    = w0f - comp( o, axis );
            const float t       = diff / comp( d, axis );
            const bool lnear    = diff > 0.0f;
            const uint32_t nr   = lnear ? left : left + 1;
            const uint32_t fr   = lnear ? left + 1 : left;
            const bool near_only = ( t > b ) | ( t < 0.0f );
            const bool far_only  = !near_only & ( t < a );
            const bool both      = at_node & !near_only & !far_only;
            node                 = far_only ? fr : nr;
            want_node            = at_node;
            if ( both ) push( fr, b );
            // leaf: the ray interval becomes [a, b]
            tmax = at_leaf ? b : tmax;
            b    = both ? t : b;
            has  = has & !at_leaf;
            ptr  = at_leaf ? w0 : ptr;
            end  = at_leaf ? w0 + ( w1 >> 2 ) : end;
            pop       = at_leaf & ( ptr == end );
            advance   = at_leaf & ( ptr != end ); // the first record goes in flight in this trip
            nq0       = at_leaf ? r0 : nq0;
            qhi       = at_leaf ? w0 + 1u : qhi;
            if ( STATS ) {
                st.inner += ( at_head ? 1u : 0u ) + ( at_node ? 1u : 0u );
                st.leaves += at_leaf ? 1u : 0u;
            }
        }
        const bool in_flight = is_test & ( MB == 0 || obj != MRT_NO_PRIM );
        // (a lane in TEST without a record in flight was waiting for references only)
        advance = advance | ( is_test & !in_flight );
#ifndef MRT_TEST_EVERY_LANE
#define MRT_TEST_EVERY_LANE 0 // 1: also in the kernels that keep a hit record (they need more than 80 registers for it)
#endif
#ifndef MRT_TEST_EVERY_LANE_OCCLUSION
#define MRT_TEST_EVERY_LANE_OCCLUSION 1
#endif
        if constexpr ( ( BOOL_ONLY && MRT_TEST_EVERY_LANE_OCCLUSION ) || MRT_TEST_EVERY_LANE )
        {
            // The primitive test of the occlusion kernels, computed in every lane, without a branch, and
            // applied where a record is in flight.  Per-lane flags live in scalar masks, and a flag assigned
            // under a divergent branch costs three scalar instructions to merge after it -- which this loop
            // feels three times as much as vector instructions.  (The kernels that keep a hit record have no
            // registers to spare for it: there the branch, which shortens live ranges, stays.)
            bool hit = false, outside = true, is_tri = true;
            float t = 0.f, u = 0.f, v = 0.f;
            uint32_t flags = 0u;
            {
                if ( STATS && in_flight ) {
                    st.refs++;
                    st.tests++;
                }
                flags             = __float_as_uint( L0.w );
                const uint32_t xf = flags >> MRT_PRIM_XF_SHIFT;
                if ( in_flight & ( xf != xr_id ) ) {
                    V3 oo = o, od = d;
                    if ( xf != 0 ) transform_ray( s, xf_lds, xf, o, d, oo, od );
                    put_xr( oo, od );
                    xr_id = xf;
                }
                const V3 origin = xro, dir = xrd;
                float det = 0.f;
                bool apart;
                is_tri             = ( flags & 3u ) == MRT_PRIM_TRI;
                const bool hit_tri = tri_test( origin, dir, L0, L1, L2, a, tmax, t, u, v, det, apart );
                uint32_t sph = 0u; // a sphere's outcome as bits of a vector register: 1 = hit, 2 = outside
                if ( in_flight & !is_tri ) {
                    float ts     = 0.f;
                    bool outs    = true;
                    const bool h = sphere_test( s, xf_lds, xf, origin, dir, o, d, a, tmax, ts, outs );
                    sph          = ( h ? 1u : 0u ) | ( outs ? 2u : 0u );
                    t            = ts;
                }
                hit     = in_flight & ( ( is_tri & hit_tri ) | ( ( sph & 1u ) != 0u ) );
                outside = is_tri ? !( det < 0.f ) : ( sph & 2u ) != 0u;
                if ( MB != 0 && ( in_flight & is_tri & apart ) ) {
#pragma unroll
                    for ( int i = MB - 1; i > 0; i-- )
                        mb[i] = mb[i - 1];
                    mb[0] = obj;
                }
            }
            // Accepted hits shrink the interval whether or not they are kept.  For a closest-hit ray
            // an accepted hit that is not kept has a distance that compares EQUAL to the kept one
            // (t <= tmax and !(t < tmax)), so tmax stays the kept hit's own distance, which is what the
            // reference reports (kdtree.cpp:547): the two differ when one is -0 and the other +0.
            const bool keep = hit & ( ANY ? !( flags & MRT_PRIM_EMISSIVE ) : ( !has | ( t < tmax ) ) );
            tmax            = ( ANY ? hit : keep ) ? t : tmax;
            if ( !BOOL_ONLY && keep ) {
                v2f m;
                m.x = __uint_as_float( obj | ( outside ? 0x80000000u : 0u ) ), m.y = is_tri ? u : 0.f; // (a sphere reports no u, v)
                *out_lds  = m;
                *outv_lds = is_tri ? v : 0.f;
            }
            has = has | keep;
            ptr += in_flight ? 1u : 0u;
            // the search ends at the first leaf that produced a hit (a shadow ray: at the first hit)
            const bool leaf_done = ptr >= end;
            const bool ends      = ANY ? ( keep | ( leaf_done & has ) ) : ( leaf_done & has );
            retire               = in_flight & ends;
            advance              = advance | ( in_flight & !ends & !leaf_done );
            pop                  = pop | ( in_flight & !ends & leaf_done );
        }
        else if ( in_flight ) {
            if ( STATS ) {
                st.refs++;
                st.tests++;
            }
            const uint32_t flags = __float_as_uint( L0.w );
            const uint32_t xf    = flags >> MRT_PRIM_XF_SHIFT;
            if ( xf != xr_id ) {
                V3 oo = o, od = d;
                if ( xf != 0 ) transform_ray( s, xf_lds, xf, o, d, oo, od );
                put_xr( oo, od );
                xr_id = xf;
            }
            const V3 origin = xro, dir = xrd;
            float t = 0.f, u = 0.f, v = 0.f, det = 0.f;
            bool outside = true, hit;
            if ( ( flags & 3u ) == MRT_PRIM_TRI ) {
                bool apart;
                hit     = tri_test( origin, dir, L0, L1, L2, a, tmax, t, u, v, det, apart );
                outside = !( det < 0.f );
                if ( MB != 0 && apart ) {
#pragma unroll
                    for ( int i = MB - 1; i > 0; i-- )
                        mb[i] = mb[i - 1];
                    mb[0] = obj;
                }
            }
            else { hit = sphere_test( s, xf_lds, xf, origin, dir, o, d, a, tmax, t, outside ); }
            // Accepted hits shrink the interval whether or not they are kept.  For a closest-hit ray
            // an accepted hit that is not kept has a distance that compares EQUAL to the kept one
            // (t <= tmax and !(t < tmax)), so tmax stays the kept hit's own distance, which is what the
            // reference reports (kdtree.cpp:547): the two differ when one is -0 and the other +0.
            const bool keep = hit & ( ANY ? !( flags & MRT_PRIM_EMISSIVE ) : ( !has | ( t < tmax ) ) );
            tmax            = ( ANY ? hit : keep ) ? t : tmax;
            if ( !BOOL_ONLY && keep ) {
                v2f m;
                m.x = __uint_as_float( obj | ( outside ? 0x80000000u : 0u ) ), m.y = u;
                *out_lds  = m;
                *outv_lds = v;
            }
            has = has | keep;
            ptr++;
            // the search ends at the first leaf that produced a hit (a shadow ray: at the first hit)
            const bool leaf_done = ptr >= end;
            retire               = ANY ? ( keep | ( leaf_done & has ) ) : ( leaf_done & has );
            advance              = !retire & !leaf_done;
            pop                  = !retire & leaf_done;
        }
        // the primitive the lane is about to fetch (nq0, at position ptr) and up to MRT_MAILBOX_SKIPS - 1
        // after it are passed over when the ray is known to miss them
        bool starve = false; // the leaf goes on, but its next reference has not been fetched yet
        if ( MB != 0 ) {
#pragma unroll
            for ( int k = 0; k < MRT_MAILBOX_SKIPS; k++ ) {
                const bool skip = advance & ( s.mailbox_on != 0u ) & mb_has( nq0 );
                if ( STATS && skip ) {
                    st.refs++;
                    st.tests++;
                }
                ptr += skip ? 1u : 0u;
                nq0 = skip ? nq1 : nq0, nq1 = skip ? nq2 : nq1, nq2 = skip ? nq3 : nq2;
                const bool leaf_done = skip & ( ptr >= end );
                const bool dry       = skip & !leaf_done & ( ptr >= qhi );
                advance              = advance & !leaf_done & !dry;
                retire               = retire | ( leaf_done & has );
                pop                  = pop | ( leaf_done & !has );
                starve               = starve | dry;
            }
        }
        state = ( advance | starve ) ? MRT_ST_TEST : state;
        if ( TAIL && tail_trips > relay_after ) {
            // (wave-uniform branch, taken only in the tail of the launch) hand over the rays that have
            // just finished a leaf without a hit and have deferred nodes left: finite rays with no zero
            // direction component, none of whose split distances can be NaN
            const float lo = fminf( fminf( fabsf( d.x ), fabsf( d.y ) ), fabsf( d.z ) ), hi = fmaxf( fmaxf( fabsf( d.x ), fabsf( d.y ) ), fabsf( d.z ) );
            const float om = fmaxf( fmaxf( fabsf( o.x ), fabsf( o.y ) ), fabsf( o.z ) );
            if ( pop && sp != 0 && lo >= 1e-30f && hi < 1e30f && om < 1e30f && fabsf( b ) < 1e30f ) {
                const unsigned long long m = __ballot( true );
                const uint32_t rank        = __builtin_amdgcn_mbcnt_hi( (uint32_t)( m >> 32 ), __builtin_amdgcn_mbcnt_lo( (uint32_t)m, 0u ) );
                uint32_t base              = 0;
                if ( rank == 0 ) base = atomicAdd( relay.count, (uint32_t)__popcll( m ) );
                base = (uint32_t)__builtin_amdgcn_readfirstlane( (int)base );
                if ( base + rank < relay.cap ) {
                    relay.list[base + rank] = make_uint2( tag, __float_as_uint( b ) );
                    pop                     = false;
                    state                   = MRT_ST_IDLE;
                }
            }
        }
        const bool do_pop = pop & ( sp != 0 );
#ifdef MRT_COUNT_DEEP_POPS
        // measurement aid (with MRT_TRIP_HISTOGRAM): bucket 28 counts the trips of all waves, 29 those in which some
        // lane takes a deferred node back from the scratch part of its stack, 30 those with a push into it
        if ( __lane_id() == 0 ) atomicAdd( &g_trip_hist[28], 1ull );
        if ( __ballot( do_pop && sp > NST ) != 0ull && __lane_id() == 0 ) atomicAdd( &g_trip_hist[29], 1ull );
        if ( __ballot( sp > NST ) != 0ull && __lane_id() == 0 ) atomicAdd( &g_trip_hist[30], 1ull );
#endif
        retire            = retire | ( pop & ( sp == 0 ) );
        if ( do_pop ) {
            --sp;
            v2u e;
            if ( sp < NST ) e = stack_lds[sp * MRT_TRACE_BLOCK];
            else e = stack_deep[sp - NST];
            a    = b;
            b    = __uint_as_float( e.y );
            node = e.x & 0x3fffffffu;
            pick = e.x >> 30;
        }
        want_node = want_node | do_pop;
        state     = do_pop ? MRT_ST_INNER : state;
        if ( retire ) {
            // a shadow ray that found an occluder in the tree never reaches the planes
            if ( STATS && !( ANY && has ) ) st.tests += pl_tests;
            DHit h;
            h.prim = has ? 0 : -1, h.t = tmax, h.u = h.v = 0.f, h.outside = 0;
            bool found = has;
            if ( !BOOL_ONLY ) {
                const v2f rec = *out_lds;
                h.prim    = has ? (int32_t)prim_index( __float_as_uint( rec.x ) & 0x7fffffffu ) : -1;
                h.t       = has ? tmax : rec.y;
                h.u       = has ? rec.y : 0.f;
                h.v       = has ? *outv_lds : 0.f;
                h.outside = has ? __float_as_uint( rec.x ) >> 31 : 0u;
                if ( has_pl ) {
                    const v4f r = *pl_lds;
                    if ( ANY ? !has : ( has ? r.y < h.t : !( r.y > h.t ) ) ) {
                        h.prim = (int32_t)__float_as_uint( r.x ), h.t = r.y, h.u = r.z, h.v = r.w, h.outside = 1;
                        found  = true;
                    }
                }
            }
            io.store( tag, found, h );
            state = MRT_ST_IDLE;
#ifdef MRT_TRIP_HISTOGRAM
#ifdef MRT_TRIP_HISTOGRAM_RELAY_ONLY
            if ( IO::is_segment_io )
#endif
            atomicAdd( &g_trip_hist[31 - __clz( (int)( trips | 1u ) )], 1ull );
#endif
        }
        // ---------------------------------------------------------------- issue
        // (Two blocks, and between them the compiler makes the wave wait for the first block's data,
        // because the second block's addresses share registers with what those loads are to deliver and it
        // cannot know that the blocks serve different lanes.  Fetching both kinds of record with one pair
        // of instructions from one base removes that wait -- and made the connection kernels 9 % slower:
        // with six waves per SIMD the memory system is oversubscribed, and the wait spaces the requests.)
        if ( advance ) {
            // the record of the next primitive goes in flight, and the next four references when the
            // queue has run out (one cache look-up per four primitives instead of one per primitive: the
            // loop's pace follows the number of look-ups)
            obj   = nq0;
            L0    = ld_prim( s, obj, 0 );
            L1    = ld_prim( s, obj, 1 );
            L2    = ld_prim_tail( s, obj );
            nq0 = nq1, nq1 = nq2, nq2 = nq3;
        }
        if ( MB != 0 && starve ) obj = MRT_NO_PRIM;
        {
            const uint32_t rpos = ptr + ( starve ? 0u : 1u ); // the position the queue must reach next
            if ( ( advance | starve ) && rpos >= qhi && rpos < end ) {
                const v4u_a4 r = ld_ref4( s, rpos );
                nq0 = r.x, nq1 = r.y, nq2 = r.z, nq3 = r.w;
                qhi = rpos + 4u;
            }
        }
        if ( want_node ) {
            cur = node;
#if MRT_TOP_LDS
            if ( node - 1u < (uint32_t)MRT_TOP_LDS ) { // node 0 is never asked for: every ray starts in it
                const v4f a0 = top_lds[( node - 1u ) * 2u], a1 = top_lds[( node - 1u ) * 2u + 1u];
                L0 = make_float4( a0.x, a0.y, a0.z, a0.w );
                L1 = make_float4( a1.x, a1.y, a1.z, a1.w );
            }
            else
#endif
            {
                L0 = ld_node( s, node, 0 );
                L1 = ld_node( s, node, 1 );
            }
        }
        return true;
    };
    while ( more )
        (void)trip( std::false_type {} );
    while ( trip( std::true_type {} ) ) {}
}

// ------------------------------------------------------------------------------------------------
// Packet traversal: the 64 rays a wave holds walk the tree TOGETHER.
//
// For rays that start at (nearly) one point and differ little in direction -- the samples of one
// pixel -- every lane of the streaming traversal above fetches the same records, keeps the same stack
// and pays the state machine's bookkeeping for it, and the kernel is bound by instruction issue, not by
// memory.  Here the node a wave is at, the stack of deferred nodes and the primitive under test are
// wave-uniform: records, references and primitives arrive through scalar loads, the per-lane work is
// the reference's arithmetic and nothing else (the split distance, the primitive test), and a mask
// says which lanes take part.
//
// Every lane still makes exactly the decisions of its own traversal, on its own interval, in its own
// front-to-back order:
//   - at an inner node every participating lane classifies itself (near only / far only / both) from
//     its own t = (split - o[axis]) / d[axis] and interval; the wave goes to the near child with the
//     lanes that want it and defers the far child with the lanes that want that (those that want both
//     shorten their interval to [a, t] and get [t, b] back when the deferred node is taken up: the b
//     they had is kept per lane with the deferred node, the new a is the b they come back with, as in
//     the single-ray traversal);
//   - which child is the near one depends on the lane's side of the split plane.  Lanes that disagree
//     (origins on both sides: only possible where the plane cuts through the lens) are split: the wave
//     goes on with those for which the left child is near and defers THE SAME NODE for the others;
//   - at a leaf the participating lanes test the leaf's primitives in stored order against their own
//     [a, b]; a lane that has accepted a hit in a leaf is finished and leaves every mask.
// A lane that does not take part in a subtree keeps its interval untouched, deferred nodes are taken up
// last-in first-out, so each lane meets its leaves in the order and with the intervals of its own
// traversal: same hit, same distance, bit for bit, for ANY 64 rays -- coherence only decides the speed.
// Closest-hit rays only.  The deferred-node stack lives in per-lane scratch arrays (a deferred node's
// words in lanes 0-5 of one array, every lane's old b in another).
#define MRT_PACKET_STACK ( 2 * MRT_STACK_MAX ) // per tree level at most one deferred child and one deferred "other side"
typedef uint32_t v4u __attribute__( ( ext_vector_type( 4 ) ) );
typedef __attribute__( ( address_space( 4 ) ) ) const v4u const_u4;         // constant address space:
typedef __attribute__( ( address_space( 4 ) ) ) const uint32_t const_u32;   // loads at wave-uniform addresses are scalar loads
typedef __attribute__( ( address_space( 4 ) ) ) const unsigned char const_u8;
typedef v4u v4u_a8 __attribute__( ( aligned( 8 ) ) );
typedef __attribute__( ( address_space( 4 ) ) ) const v4u_a8 const_u4_a8;
typedef __attribute__( ( address_space( 4 ) ) ) const v2u const_u2;

template <class IO>
MRT_DEV void trace_packets( const DTrace& s, lds_u32* lds, const IO& io, uint32_t n, uint32_t* deal ) {
    const lds_f32* xf_lds = (const lds_f32*)lds;
    stage_transforms( s, (lds_f32*)lds );
    const_u4* const cnodes  = (const_u4*)(uintptr_t)s.nodes;
    const_u32* const crefs  = (const_u32*)(uintptr_t)s.refs;
    const_u8* const cprims  = (const_u8*)(uintptr_t)s.prims;
    const uint32_t lane     = __lane_id();
    const unsigned long long my_bit = 1ull << lane;
    uint32_t st_words[MRT_PACKET_STACK]; // deferred node e: lane 0 its reference, 1-2 the lanes that enter it, 3-4 those of them whose interval is restored, 5 its kind
    uint32_t st_b[MRT_PACKET_STACK];     // every lane's b at the time node e was deferred

    const uint32_t n_chunks = ( n + ( MRT_CHUNK - 1u ) ) / MRT_CHUNK;
    for ( uint32_t part_try = 0; part_try < 8u; ) {
        // the next chunk of this XCD's part of the list, then of the other parts (as in trace_stream)
        const uint32_t part = ( ( blockIdx.x & s.xcd_mask ) + part_try ) & 7u;
        uint32_t got        = 0;
        if ( lane == 0 ) got = atomicAdd( deal + part * MRT_DEAL_STRIDE, 1u );
        got              = (uint32_t)__builtin_amdgcn_readfirstlane( (int)got );
        const uint32_t c = ( ( got / MRT_DEAL_RUN ) * 8u + part ) * MRT_DEAL_RUN + got % MRT_DEAL_RUN;
        if ( c >= n_chunks ) {
            part_try++;
            continue;
        }
        for ( uint32_t base = c * MRT_CHUNK; base < ( c + 1u ) * MRT_CHUNK && base < n; base += 64u ) {
            // ------------------------------------------------------------ fetch
            const uint32_t idx = base + lane;
            const bool valid   = idx < n;
            uint32_t tag       = 0;
            V3 o = mk3( 0.f ), d = mk3( 0.f );
            float a = 0.f, b = 0.f, ray_tmax = 0.f;
            bool in = false; // this lane takes part in the subtree the wave is in
            if ( valid ) {
                io.load( idx, tag, o, d, a, ray_tmax );
                b  = ray_tmax;
                // (a NaN ray fails every test of the reference after walking the whole tree: same result without the walk)
                in = !( isnan( o.x ) | isnan( o.y ) | isnan( o.z ) | isnan( d.x ) | isnan( d.y ) | isnan( d.z ) ) && clip_root( s, o, d, a, b );
            }
            unsigned long long act = __ballot( in ), done = 0ull;
            float tmax = 0.f, hu = 0.f, hv = 0.f;
            uint32_t hprim = 0;
            bool has       = false;
            V3 xro = o, xrd = d; // object-space ray under transform xr_id (wave-uniform)
            uint32_t xr_id = 0;
            uint32_t ref   = 0; // where the wave is: bits 0-29 a record, bit 31 set = one of its children (bit 30: the right one), else its head
            uint32_t sp    = 0;
            uint32_t cur_rec = 0xffffffffu;
            v4u R0 = { 0u, 0u, 0u, 0u }, R1 = R0;
            // ------------------------------------------------------------ walk
            for ( ;; ) {
                if ( act == 0ull ) { // take up the last deferred node some unfinished lane enters
                    bool found_one = false;
                    while ( sp > 0u ) {
                        --sp;
                        const uint32_t w  = st_words[sp];
                        const unsigned long long m_enter = ( (unsigned long long)(uint32_t)__builtin_amdgcn_readlane( (int)w, 2 ) << 32 | (uint32_t)__builtin_amdgcn_readlane( (int)w, 1 ) ) & ~done;
                        if ( m_enter == 0ull ) continue;
                        const unsigned long long m_restore = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane( (int)w, 4 ) << 32 | (uint32_t)__builtin_amdgcn_readlane( (int)w, 3 );
                        ref = (uint32_t)__builtin_amdgcn_readlane( (int)w, 0 );
                        act = m_enter;
                        in  = ( act & my_bit ) != 0ull;
                        if ( in && ( m_restore & my_bit ) ) {
                            a = b;
                            b = __uint_as_float( st_b[sp] );
                        }
                        found_one = true;
                        break;
                    }
                    if ( !found_one ) break;
                }
                const uint32_t rec = (uint32_t)__builtin_amdgcn_readfirstlane( (int)( ref & 0x3fffffffu ) );
                if ( rec != cur_rec ) {
                    R0      = cnodes[2u * rec];
                    R1      = cnodes[2u * rec + 1u];
                    cur_rec = rec;
                }
                // the node the wave is at: an inner node (split, axis, its two children as references) or a leaf
                float split;
                uint32_t axis, ref_l, ref_r;
                bool leaf = false;
                uint32_t w0 = 0, w1 = 0, r0 = 0;
                if ( !( ref & 0x80000000u ) && ( R0.y & 3u ) != 3u ) { // the head of the record
                    split = __uint_as_float( R0.x ), axis = R0.y & 3u;
                    ref_l = rec | 0x80000000u, ref_r = rec | 0xc0000000u;
                }
                else { // a child (a head with "axis" 3 is no node: the left slot is the record's node)
                    const bool right = ( ref & 0x80000000u ) && ( ref & 0x40000000u );
                    w0 = right ? R1.x : R0.z, w1 = right ? R1.y : R0.w, r0 = right ? R1.w : R1.z;
                    leaf  = ( w1 & 3u ) == 3u;
                    split = __uint_as_float( w0 ), axis = w1 & 3u;
                    ref_l = w1 >> 2, ref_r = ( w1 >> 2 ) + 1u;
                }
                if ( !leaf ) {
                    const float diff = split - comp( o, (int)axis );
                    const float t    = diff / comp( d, (int)axis );
                    const bool lnear = diff > 0.0f;
                    const unsigned long long m_l = __ballot( in & lnear ), m_r = __ballot( in & !lnear );
                    bool near_left = m_l != 0ull;
                    if ( m_l != 0ull && m_r != 0ull ) { // origins on both sides of the plane: this node again, later, for the other side
                        st_words[sp] = lane == 0u ? ref : lane == 1u ? (uint32_t)m_r : lane == 2u ? (uint32_t)( m_r >> 32 ) : 0u;
                        sp++;
                        act = m_l;
                        in  = in & lnear;
                    }
                    const bool near_only = ( t > b ) | ( t < 0.0f );
                    const bool far_only  = !near_only & ( t < a );
                    const bool both      = !near_only & !far_only;
                    const unsigned long long m_n = __ballot( in & !far_only ), m_f = __ballot( in & !near_only ), m_b = __ballot( in & both );
                    const uint32_t ref_n = near_left ? ref_l : ref_r, ref_f = near_left ? ref_r : ref_l;
                    if ( m_n == 0ull ) {
                        act = m_f;
                        ref = ref_f;
                    }
                    else {
                        if ( m_f != 0ull ) {
                            st_words[sp] = lane == 0u ? ref_f : lane == 1u ? (uint32_t)m_f : lane == 2u ? (uint32_t)( m_f >> 32 ) : lane == 3u ? (uint32_t)m_b : lane == 4u ? (uint32_t)( m_b >> 32 ) : 0u;
                            st_b[sp]     = __float_as_uint( b );
                            sp++;
                        }
                        b   = ( in & both ) ? t : b;
                        act = m_n;
                        in  = in & !far_only;
                        ref = ref_n;
                    }
                    continue;
                }
                // ------------------------------------------------------------ leaf: the interval becomes [a, b]
                if ( in ) {
                    tmax = b;
                    has  = false;
                }
                const uint32_t cnt = w1 >> 2;
                for ( uint32_t k = 0; k < cnt; k++ ) {
                    const uint32_t obj = k == 0u ? r0 : crefs[w0 + k];
                    const uint32_t off = prim_offset( obj );
                    const v4u q0 = *(const_u4_a8*)( cprims + off ), q1 = *(const_u4_a8*)( cprims + off + 16u );
                    const v2u q2 = *(const_u2*)( cprims + off + 32u );
                    const uint32_t flags = q0.w, xf = flags >> MRT_PRIM_XF_SHIFT;
                    if ( xf != xr_id ) {
                        xro = o, xrd = d;
                        if ( xf != 0u ) transform_ray( s, xf_lds, xf, o, d, xro, xrd );
                        xr_id = xf;
                    }
                    float t = 0.f, u = 0.f, v = 0.f, det = 0.f;
                    bool outside = true, hit = false;
                    if ( ( flags & 3u ) == MRT_PRIM_TRI ) {
                        hit     = tri_test( xro, xrd, make_float4( __uint_as_float( q0.x ), __uint_as_float( q0.y ), __uint_as_float( q0.z ), 0.f ),
                                            make_float4( __uint_as_float( q1.x ), __uint_as_float( q1.y ), __uint_as_float( q1.z ), __uint_as_float( q1.w ) ),
                                            make_float2( __uint_as_float( q2.x ), __uint_as_float( q2.y ) ), a, tmax, t, u, v, det );
                        outside = !( det < 0.f );
                    }
                    else { hit = sphere_test( s, xf_lds, xf, xro, xrd, o, d, a, tmax, t, outside ); }
                    // a later hit replaces the current one only if strictly nearer (see trace_stream)
                    const bool keep = in & hit & ( !has | ( t < tmax ) );
                    tmax            = keep ? t : tmax;
                    hprim           = keep ? ( obj | ( outside ? 0x80000000u : 0u ) ) : hprim;
                    hu              = keep ? u : hu;
                    hv              = keep ? v : hv;
                    has             = has | keep;
                }
                // the search ends at the first leaf that produced a hit
                done |= __ballot( in & has );
                act = 0ull;
                in  = false;
            }
            // ------------------------------------------------------------ retire
            if ( valid ) {
                DHit h;
                h.prim    = has ? (int32_t)prim_index( hprim & 0x7fffffffu ) : -1;
                h.t       = has ? tmax : ray_tmax;
                h.u       = has ? hu : 0.f;
                h.v       = has ? hv : 0.f;
                h.outside = has ? hprim >> 31 : 0u;
                bool found = has;
                if ( s.n_out != 0u ) {
                    v4f rec4;
                    uint32_t tested;
                    const bool has_pl = scan_planes<false>( s, xf_lds, o, d, ray_tmax, rec4, tested );
                    found             = merge_plane<false>( has, h, has_pl, rec4 );
                }
                io.store( tag, found, h );
            }
        }
    }
}

// A hit as one 16-byte record: (prim | outside << 31 as bits, or 0xffffffff for a miss; t; u; v).
MRT_DEV float4 pack_hit( const DHit& h ) {
    const uint32_t bits = h.prim < 0 ? 0xffffffffu : ( (uint32_t)h.prim | ( h.outside ? 0x80000000u : 0u ) );
    return make_float4( __uint_as_float( bits ), h.t, h.u, h.v );
}
MRT_DEV void unpack_hit( float4 r, DHit& h ) {
    const uint32_t bits = __float_as_uint( r.x );
    h.prim              = bits == 0xffffffffu ? -1 : (int32_t)( bits & 0x7fffffffu );
    h.outside           = bits == 0xffffffffu ? 0u : bits >> 31;
    h.t = r.y, h.u = r.z, h.v = r.w;
}

// The segments of the relayed rays as a ray list for trace_stream (see Relay): ray j is segment
// j % MRT_RELAY_K of entry j / MRT_RELAY_K.  RAYS::reload( tag, o, d, tmax ) returns the ray an entry
// names.  The interval that remains -- from the distance the entry carries to where the ray leaves
// the root box or ends -- is cut at lo + (hi - lo) * k / K; neighbouring segments share their end
// point, the first starts exactly at lo and the last ends exactly at hi.  The tree passed to the
// kernel has no out-of-tree planes: they are the business of relay_finish_kernel.
template <class RAYS, bool BOOL_ONLY>
struct SegmentIO {
    static constexpr bool is_segment_io = true;
    RAYS rays;
    const uint2* list;
    float root_min[3], root_max[3];
    float4* hits;    // !BOOL_ONLY: one hit record per segment
    uint32_t* flags; // BOOL_ONLY: one word per entry, zero before the launch, set when a segment found something
    MRT_DEV void load( uint32_t j, uint32_t& tag, V3& o, V3& d, float& t0, float& t1 ) const {
        tag            = j;
        const uint2 e  = list[j / MRT_RELAY_K];
        const uint32_t k = j % MRT_RELAY_K;
        float hi;
        rays.reload( e.x, o, d, hi );
        float lo = __uint_as_float( e.y );
        (void)clip_box( root_min, root_max, o, d, lo, hi );
        const float w = hi - lo;
        t0 = k == 0u ? lo : fminf( lo + w * ( (float)k * ( 1.f / MRT_RELAY_K ) ), hi );
        t1 = k + 1u == MRT_RELAY_K ? hi : fminf( lo + w * ( (float)( k + 1u ) * ( 1.f / MRT_RELAY_K ) ), hi );
    }
    MRT_DEV void store( uint32_t j, bool found, const DHit& h ) const {
        if ( BOOL_ONLY ) {
            if ( found ) flags[j / MRT_RELAY_K] = 1u;
        }
        else {
            DHit r = h;
            if ( !found ) r.prim = -1;
            hits[j] = pack_hit( r );
        }
    }
};

template <bool ANY, bool BOOL_ONLY, class RAYS>
__global__ void __launch_bounds__( MRT_TRACE_BLOCK, MRT_TRACE_WAVES ) relay_trace_kernel( DTrace tr, SegmentIO<RAYS, BOOL_ONLY> io, const uint32_t* __restrict__ count, uint32_t cap, uint32_t* __restrict__ deal ) {
    __shared__ uint32_t lds[MRT_TRACE_LDS_WORDS( BOOL_ONLY )];
    lds_u32* lp = (lds_u32*)lds;
    const uint32_t n = *count < cap ? *count : cap;
    DStats st { 0, 0, 0, 0 };
    trace_stream<ANY, BOOL_ONLY, false>( tr, lp, io, n * MRT_RELAY_K, deal, st );
}

// Completes the relayed rays: the first segment that found something is the ray's tree result, the
// planes are merged as at any retirement, and the ray's own IO stores the outcome.  IO::reload as in
// SegmentIO.
template <bool ANY, bool BOOL_ONLY, class IO>
__global__ void __launch_bounds__( 256 ) relay_finish_kernel( DTrace tr, IO io, Relay relay, const float4* __restrict__ hits, uint32_t* __restrict__ flags, unsigned long long* __restrict__ total ) {
    __shared__ uint32_t lds[MRT_XF_LDS * 12];
    lds_u32* lp = (lds_u32*)lds;
    stage_transforms( tr, (lds_f32*)lp );
    const uint32_t n = *relay.count < relay.cap ? *relay.count : relay.cap;
    if ( blockIdx.x == 0 && threadIdx.x == 0 && n != 0u ) atomicAdd( total, (unsigned long long)n ); // diagnostics: rays relayed on this handle so far
    for ( uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x ) {
        const uint2 e = relay.list[i];
        V3 o, d;
        float far;
        io.reload( e.x, o, d, far );
        DHit h;
        h.prim = -1, h.t = far, h.u = h.v = 0.f, h.outside = 0;
        bool has = false;
        if ( BOOL_ONLY ) {
            has      = flags[i] != 0u;
            flags[i] = 0u; // zero again for the next launch that uses the buffer
            h.prim   = has ? 0 : -1;
        }
        else {
            for ( uint32_t k = 0; k < MRT_RELAY_K && !has; k++ ) {
                const float4 r = hits[(size_t)i * MRT_RELAY_K + k];
                if ( __float_as_uint( r.x ) != 0xffffffffu ) {
                    unpack_hit( r, h );
                    has = true;
                }
            }
        }
        bool found = has;
        if ( !BOOL_ONLY && tr.n_out != 0u ) {
            v4f rec;
            uint32_t tested;
            const bool has_pl = scan_planes<ANY>( tr, (const lds_f32*)lp, o, d, far, rec, tested );
            found             = merge_plane<ANY>( has, h, has_pl, rec );
        }
        io.store( e.x, found, h );
    }
}

// Adds the lanes' counters to stats[0..3]: one atomic per wave and counter.
MRT_DEV void flush_stats( const DStats& st, unsigned long long* stats ) {
    unsigned long long vals[4] = { st.inner, st.leaves, st.refs, st.tests };
    for ( int k = 0; k < 4; k++ ) {
        unsigned long long x = vals[k];
        for ( int off = 32; off > 0; off >>= 1 )
            x += __shfl_down( x, off );
        if ( __lane_id() == 0 ) atomicAdd( &stats[k], x );
    }
}

// Grid of a streaming traversal over n rays (n = upper bound when the count lives on the device):
// as many blocks as the chip keeps resident for this kernel (the ray list is dealt statically to
// the waves of the grid, so a block that had to wait for a slot would serialise its share), fewer
// when the list is short.
template <class Kernel>
inline uint32_t trace_grid_blocks( Kernel kernel, uint64_t n, int n_cus ) {
    // resident blocks per CU of this kernel instantiation on each device (0 = not asked yet); several
    // host threads may render on several devices at once
    static std::mutex mu;
    static int per_cu_of[64] = {};
    int dev = 0;
    (void)hipGetDevice( &dev );
    int per_cu = 0;
    {
        std::lock_guard<std::mutex> lock( mu );
        int& slot = per_cu_of[dev >= 0 && dev < 64 ? dev : 0];
        if ( slot == 0 ) {
            int v = 0;
            if ( hipOccupancyMaxActiveBlocksPerMultiprocessor( &v, kernel, MRT_TRACE_BLOCK, 0 ) != hipSuccess || v < 1 ) v = 1;
            if ( const char* e = std::getenv( "MRT_TRACE_BLOCKS_PER_CU" ) ) { // measurements: fewer resident blocks than fit
                const int want = std::atoi( e );
                if ( want >= 1 && want < v ) v = want;
            }
            slot = v;
        }
        per_cu = slot;
    }
    const uint64_t full = (uint64_t)n_cus * (uint64_t)per_cu;
    const uint64_t want = ( n + MRT_TRACE_BLOCK - 1 ) / MRT_TRACE_BLOCK;
    return (uint32_t)( want < full ? ( want ? want : 1u ) : full );
}
