// TEST INFRASTRUCTURE ONLY -- the CPU oracle's C entry points.  See mrt_oracle.h.
#include "mrt_oracle.h"

#include "orc_build.h"
#include "orc_math.h"
#include "orc_scene.h"
#include "orc_shading.h"

#include <cstring>
#include <omp.h>

using namespace orc;

namespace {

void put3( float* dst, V3 v ) {
    dst[0] = v.x;
    dst[1] = v.y;
    dst[2] = v.z;
}
void put_m3( float* dst, const M3& m ) {
    for ( int c = 0; c < 3; c++ )
        for ( int r = 0; r < 3; r++ )
            dst[3 * c + r] = m.c[c][r];
}

struct Walk {
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
    int64_t n_nodes = 0, n_refs = 0;
    int depth = 0;
};

int64_t walk( const Node* node, Walk& c, int depth ) {
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

// ABI layout: the two children of an inner node are adjacent; pairs are laid out pre-order.
struct AbiEmit {
    mrt_kdnode* nodes;
    uint32_t* refs;
    uint32_t next_node = 1, next_ref = 0;
};
void emit( const Node* n, uint32_t at, AbiEmit& e ) {
    if ( n->leaf ) {
        e.nodes[at].w0 = e.next_ref;
        e.nodes[at].w1 = 3u | ( (uint32_t)n->objects.size() << 2 );
        for ( int o : n->objects )
            e.refs[e.next_ref++] = (uint32_t)o;
        return;
    }
    uint32_t l = e.next_node;
    e.next_node += 2;
    std::memcpy( &e.nodes[at].w0, &n->split, 4 );
    e.nodes[at].w1 = (uint32_t)n->axis | ( l << 2 );
    emit( n->left, l, e );
    emit( n->right, l + 1, e );
}

} // namespace

struct orc_tree {
    Tree* t;
};

extern "C" {

void orc_transform( int n_ops, const float* ops, mrt_transform* out ) {
    Xf x;
    for ( int i = 0; i < n_ops; i++ ) {
        const float* o = ops + 5 * i;
        int kind       = (int)o[0];
        if ( kind == 0 ) x.scale( o[1], o[2], o[3] );
        else if ( kind == 1 ) x.rotate( v3( o[1], o[2], o[3] ), o[4] );
        else x.translate( v3( o[1], o[2], o[3] ) );
    }
    put_m3( out->m, x.m );
    put_m3( out->inv, x.inv );
    put3( out->t, x.t );
}

void orc_sphere_from_transform( const mrt_transform* xf, float* cr ) { // scene.cpp:33-44
    Xf x     = xf_from( *xf );
    V3 c     = x.from( v3( 0.f ) );
    float s0 = dot( column( x.m, 0 ), column( x.m, 0 ) );
    float s1 = dot( column( x.m, 1 ), column( x.m, 1 ) );
    float s2 = dot( column( x.m, 2 ), column( x.m, 2 ) );
    float mx = std::max( s0, std::max( s1, s2 ) );
    put3( cr, c );
    cr[3] = 1.f * std::sqrt( mx );
}

orc_tree* orc_build( const mrt_scene_desc* desc ) {
    return new orc_tree { init_kdtree( desc ) };
}
void orc_tree_free( orc_tree* t ) {
    if ( !t ) return;
    delete t->t;
    delete t;
}
void orc_tree_counts( const orc_tree* t, int64_t* n_nodes, int64_t* n_refs, int32_t* depth ) {
    Walk c {};
    if ( t->t->root ) walk( t->t->root, c, 0 );
    *n_nodes = c.n_nodes;
    *n_refs  = c.n_refs;
    *depth   = c.depth;
}
void orc_tree_flatten( const orc_tree* t,
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
    Walk c { leaf, axis, split, left, right, bmin, bmax, ref_off, ref_cnt, refs };
    if ( t->t->root ) walk( t->t->root, c, 0 );
}
void orc_tree_to_abi( const orc_tree* t,
                      mrt_kdnode* nodes,
                      uint32_t* refs,
                      float* root_min3,
                      float* root_max3,
                      uint32_t* out_of_tree,
                      uint32_t* n_out_of_tree ) {
    if ( t->t->root ) {
        AbiEmit e { nodes, refs };
        emit( t->t->root, 0, e );
        put3( root_min3, t->t->root->min );
        put3( root_max3, t->t->root->max );
    }
    *n_out_of_tree = (uint32_t)t->t->outOfTree.size();
    for ( size_t i = 0; i < t->t->outOfTree.size(); i++ )
        out_of_tree[i] = (uint32_t)t->t->outOfTree[i];
}

void orc_intersect( const mrt_scene_desc* desc,
                    int64_t n,
                    const float* o3,
                    const float* d3,
                    const float* tmin,
                    const float* tmax,
                    int shadow,
                    int32_t* prim,
                    float* t,
                    float* u,
                    float* v,
                    float* pos3,
                    float* nrm3,
                    uint8_t* outside,
                    mrt_trace_stats* stats,
                    int n_threads ) {
    TraceScene sc( desc );
    uint64_t s_inner = 0, s_leaves = 0, s_refs = 0, s_tests = 0;
    if ( n_threads < 1 ) n_threads = 1;
#pragma omp parallel for num_threads( n_threads ) schedule( dynamic, 4096 ) \
    reduction( + : s_inner, s_leaves, s_refs, s_tests )
    for ( int64_t i = 0; i < n; i++ ) {
        Ray ray( v3( o3 + 3 * i ), v3( d3 + 3 * i ), tmin[i], tmax[i] );
        ray.shadow           = shadow != 0;
        ray.hasDifferentials = false;
        Isect isect;
        mrt_trace_stats st {};
        bool h = intersect_kdtree( sc, &ray, &isect, &st );
        if ( prim ) prim[i] = h ? isect.prim : -1;
        if ( t ) t[i] = ray.tmax;
        if ( u ) u[i] = isect.u;
        if ( v ) v[i] = isect.v;
        if ( pos3 ) put3( pos3 + 3 * i, isect.position );
        if ( nrm3 ) put3( nrm3 + 3 * i, isect.normal );
        if ( outside ) outside[i] = isect.isOutside ? 1 : 0;
        s_inner += st.inner;
        s_leaves += st.leaves;
        s_refs += st.refs;
        s_tests += st.prim_tests;
    }
    if ( stats ) {
        stats->inner      = s_inner;
        stats->leaves     = s_leaves;
        stats->refs       = s_refs;
        stats->prim_tests = s_tests;
    }
}

void orc_object_intersect( const mrt_scene_desc* desc,
                           int obj,
                           int64_t n,
                           const float* o3,
                           const float* d3,
                           const float* tmin,
                           const float* tmax,
                           int shadow,
                           uint8_t* hit,
                           float* t,
                           float* u,
                           float* v,
                           float* pos3,
                           float* nrm3,
                           uint8_t* outside ) {
    Shape s = object_shape( desc, obj );
    for ( int64_t i = 0; i < n; i++ ) {
        Ray ray( v3( o3 + 3 * i ), v3( d3 + 3 * i ), tmin[i], tmax[i] );
        ray.shadow = shadow != 0;
        Isect isect;
        bool h = shape_intersect( s, &ray, &isect );
        hit[i] = h ? 1 : 0;
        t[i]   = ray.tmax;
        u[i]   = isect.u;
        v[i]   = isect.v;
        put3( pos3 + 3 * i, isect.position );
        put3( nrm3 + 3 * i, isect.normal );
        outside[i] = isect.isOutside ? 1 : 0;
    }
}

void orc_intersect_aabb( int64_t n,
                         const float* o3,
                         const float* d3,
                         const float* tmin,
                         const float* tmax,
                         const float* bmin3,
                         const float* bmax3,
                         uint8_t* hit,
                         float* otmin,
                         float* otmax ) {
    for ( int64_t i = 0; i < n; i++ ) {
        Ray ray( v3( o3 + 3 * i ), v3( d3 + 3 * i ), tmin[i], tmax[i] );
        hit[i]   = intersect_aabb( &ray, v3( bmin3 ), v3( bmax3 ) ) ? 1 : 0;
        otmin[i] = ray.tmin;
        otmax[i] = ray.tmax;
    }
}

} // extern "C"

#include "orc_api_shading.inc"
