// TEST INFRASTRUCTURE ONLY (CPU oracle) -- see mrt_oracle.h.
//
// Literal restatement of the reference's SAH kd-tree build, kdtree.cpp:29-464, including the
// behaviours that decide the shape of the tree:
//   - events sorted with std::sort on `b` alone (kdtree.cpp:311-313,355), so the order among
//     equal keys is whatever libstdc++'s introsort produces for that input order;
//   - the sweep advances `i` inside the three while loops AND in the for header
//     (kdtree.cpp:358-373), so the event after each group is skipped;
//   - candidates must lie strictly inside the node (kdtree.cpp:380);
//   - termination test `c == INFINITY || c > 1.5 * N` in double (kdtree.cpp:412);
//   - children receive objects by closed bounding-box overlap (kdtree.cpp:182-195,437-449);
//   - no depth limit.
#pragma once
#include "orc_scene.h"

#include <algorithm>
#include <cmath>
#include <vector>

namespace orc {

struct Node { // KdTreeNode (kdtree.h:11-21)
    bool leaf  = false;
    int axis   = 0;
    float split = 0;
    int depth  = 0;
    std::vector<int> objects;
    Node* left  = nullptr;
    Node* right = nullptr;
    V3 min { 0, 0, 0 }, max { 0, 0, 0 };
    ~Node() {
        delete left;
        delete right;
    }
};

struct Event { // kdtree.cpp:45-50
    int s;
    float b;
    int k;
    int type;
};
enum { STARTING = 0, LYING = 1, ENDING = 2 };

inline bool comp_events( Event i, Event j ) { // kdtree.cpp:311-313
    return i.b < j.b;
}

struct Builder {
    const mrt_scene_desc* d;
    std::vector<Shape> shapes;

    explicit Builder( const mrt_scene_desc* desc ) : d( desc ) {
        for ( uint32_t i = 0; i < desc->n_objects; i++ )
            shapes.push_back( object_shape( desc, i ) );
    }

    static float surfaceArea( V3 mn, V3 mx ) { // kdtree.cpp:212-217
        float dx = mx.x - mn.x;
        float dy = mx.y - mn.y;
        float dz = mx.z - mn.z;
        return 2 * dx * dy + 2 * dx * dz + 2 * dy * dz;
    }
    static float p_VSub_V( V3 min1, V3 max1, V3 min2, V3 max2 ) { // :219-221
        return surfaceArea( min1, max1 ) / surfaceArea( min2, max2 );
    }
    static float lambda( int nl, int nr ) { // :223-226
        if ( nl == 0 || nr == 0 ) return 0.8f;
        return 1.f;
    }
    static float cost( int nl, int nr, float pl, float pr ) { // :228-233
        float Kt       = 1.0;
        float Ki       = 1.5;
        float lambda_p = lambda( nl, nr );
        return ( lambda_p * ( Kt + Ki * ( pl * nl + pr * nr ) ) );
    }
    static void sah( V3 mn, V3 mx, float p, int nl, int nr, int np, int k, float& c ) { // :251-266
        c         = INFINITY;
        V3 min_vl = mn, max_vl = mx, min_vr = mn, max_vr = mx;
        max_vl.at( k ) = p;
        min_vr.at( k ) = p;
        float pl  = p_VSub_V( min_vl, max_vl, mn, mx );
        float pr  = p_VSub_V( min_vr, max_vr, mn, mx );
        float cpl = cost( nl + np, nr, pl, pr );
        float cpr = cost( nl, nr + np, pl, pr );
        if ( cpl < cpr ) { c = cpl; }
        else { c = cpr; }
    }
    void clipSphereToBox( int sphere, V3 mn, V3 mx, V3& minb, V3& maxb ) const { // :268-278
        const Shape& s = shapes[sphere];
        float radius   = s.radius;
        minb           = v3( s.center.x - radius, s.center.y - radius, s.center.z - radius );
        maxb           = v3( s.center.x + radius, s.center.y + radius, s.center.z + radius );
        for ( int k = 0; k < 3; k++ ) {
            if ( mn[k] > minb[k] ) { minb.at( k ) = mn[k]; }
            if ( mx[k] < maxb[k] ) { maxb.at( k ) = mx[k]; }
        }
    }
    void clipTriangleToBox( int tri, V3 mn, V3 mx, V3& minb, V3& maxb ) const { // :280-302
        const Shape& s = shapes[tri];
        V3 v0          = s.xf.from( s.p1 );
        V3 v1          = s.xf.from( s.p2 );
        V3 v2          = s.xf.from( s.p3 );
        const float minx = std::min( v0.x, std::min( v1.x, v2.x ) );
        const float maxx = std::max( v0.x, std::max( v1.x, v2.x ) );
        const float miny = std::min( v0.y, std::min( v1.y, v2.y ) );
        const float maxy = std::max( v0.y, std::max( v1.y, v2.y ) );
        const float minz = std::min( v0.z, std::min( v1.z, v2.z ) );
        const float maxz = std::max( v0.z, std::max( v1.z, v2.z ) );
        minb             = v3( minx, miny, minz );
        maxb             = v3( maxx, maxy, maxz );
        for ( int k = 0; k < 3; k++ ) {
            if ( mn[k] > minb[k] ) { minb.at( k ) = mn[k]; }
            if ( mx[k] < maxb[k] ) { maxb.at( k ) = mx[k]; }
        }
    }
    static bool isPlanar( V3 mn, V3 mx ) { // :304-309
        float dx = mx.x - mn.x;
        float dy = mx.y - mn.y;
        float dz = mx.z - mn.z;
        return dx <= 0.001 || dy <= 0.001 || dz <= 0.001;
    }

    void findPlane( Node* node, float& p_, float& k_, float& c_ ) const { // :316-390
        c_ = INFINITY;
        p_ = 0;
        for ( int k = 0; k < 3; k++ ) {
            std::vector<Event> events;
            for ( size_t i = 0; i < node->objects.size(); i++ ) {
                V3 minb, maxb;
                if ( shapes[node->objects[i]].kind == MRT_SPHERE ) {
                    clipSphereToBox( node->objects[i], node->min, node->max, minb, maxb );
                }
                else { clipTriangleToBox( node->objects[i], node->min, node->max, minb, maxb ); }
                if ( isPlanar( minb, maxb ) ) {
                    events.push_back( Event { node->objects[i], minb[k], k, LYING } );
                }
                else {
                    events.push_back( Event { node->objects[i], minb[k], k, STARTING } );
                    events.push_back( Event { node->objects[i], maxb[k], k, ENDING } );
                }
            }
            std::sort( events.begin(), events.end(), comp_events );

            int nl = 0, np = 0, nr = (int)node->objects.size();
            for ( size_t i = 0; i < events.size(); i++ ) {
                float p       = events[i].b;
                int pStarting = 0, pEnding = 0, pLying = 0;
                while ( i < events.size() && events[i].b == p && events[i].type == ENDING ) {
                    pEnding++;
                    i++;
                }
                while ( i < events.size() && events[i].b == p && events[i].type == LYING ) {
                    pLying++;
                    i++;
                }
                while ( i < events.size() && events[i].b == p && events[i].type == STARTING ) {
                    pStarting++;
                    i++;
                }
                np = pLying;
                nr -= pLying;
                nr -= pEnding;

                float c;
                sah( node->min, node->max, p, nl, nr, np, k, c );
                if ( c < c_ && !( p <= node->min[k] ) && !( p >= node->max[k] ) ) {
                    c_ = c;
                    p_ = p;
                    k_ = k;
                }
                nl += pStarting;
                nl += pLying;
                np = 0;
            }
        }
    }

    static bool boxOverlap( V3 mn, V3 mx, V3 aabbMin, V3 aabbMax ) { // :193-194
        return !( ( mn.x > aabbMax.x ) || ( mx.x < aabbMin.x ) || ( mn.y > aabbMax.y ) ||
                  ( mx.y < aabbMin.y ) || ( mn.z > aabbMax.z ) || ( mx.z < aabbMin.z ) );
    }
    static bool intersectTriangleAabb( V3 p1, V3 p2, V3 p3, V3 aabbMin, V3 aabbMax ) { // :182-195
        float minx = fmin_glm( p1.x, fmin_glm( p2.x, p3.x ) );
        float miny = fmin_glm( p1.y, fmin_glm( p2.y, p3.y ) );
        float minz = fmin_glm( p1.z, fmin_glm( p2.z, p3.z ) );
        float maxx = fmax_glm( p1.x, fmax_glm( p2.x, p3.x ) );
        float maxy = fmax_glm( p1.y, fmax_glm( p2.y, p3.y ) );
        float maxz = fmax_glm( p1.z, fmax_glm( p2.z, p3.z ) );
        return boxOverlap( v3( minx, miny, minz ), v3( maxx, maxy, maxz ), aabbMin, aabbMax );
    }
    static bool intersectSphereAabb( V3 c, float r, V3 aabbMin, V3 aabbMax ) { // :203-210
        V3 closest = v3( fmin_glm( fmax_glm( c.x, aabbMin.x ), aabbMax.x ),
                         fmin_glm( fmax_glm( c.y, aabbMin.y ), aabbMax.y ),
                         fmin_glm( fmax_glm( c.z, aabbMin.z ), aabbMax.z ) );
        V3 seg     = closest - c;
        float d2   = dot( seg, seg );
        return d2 < ( r * r );
    }

    void subdivide( Node* node ) const { // :392-464
        int d            = ( node->depth ) % 3;
        Node* node_left  = new Node();
        Node* node_right = new Node();
        node_left->axis = node_right->axis = d;
        node_left->depth = node_right->depth = node->depth + 1;
        node_left->min  = node->min;
        node_left->max  = node->max;
        node_right->min = node->min;
        node_right->max = node->max;

        float p    = 0;
        float axis = 0;
        float c    = 0;
        findPlane( node, p, axis, c );

        if ( c == INFINITY || c > 1.5 * node->objects.size() ) {
            node->leaf = true;
            delete node_left; // the reference leaks these two
            delete node_right;
            return;
        }

        node->split                   = p;
        node->axis                    = (int)axis;
        node_left->max.at( (int)axis )  = p;
        node_right->min.at( (int)axis ) = p;
        node->left                    = node_left;
        node->right                   = node_right;

        for ( size_t i = 0; i < node->objects.size(); i++ ) {
            const Shape& s = shapes[node->objects[i]];
            bool left, right;
            if ( s.kind == MRT_SPHERE ) {
                left  = intersectSphereAabb( s.center, s.radius, node_left->min, node_left->max );
                right = intersectSphereAabb( s.center, s.radius, node_right->min, node_right->max );
            }
            else {
                V3 p1 = s.xf.from( s.p1 );
                V3 p2 = s.xf.from( s.p2 );
                V3 p3 = s.xf.from( s.p3 );
                left  = intersectTriangleAabb( p1, p2, p3, node_left->min, node_left->max );
                right = intersectTriangleAabb( p1, p2, p3, node_right->min, node_right->max );
            }
            if ( left ) { node_left->objects.push_back( node->objects[i] ); }
            if ( right ) { node_right->objects.push_back( node->objects[i] ); }
        }
        node->objects.clear();
        node->objects.shrink_to_fit();

        subdivide( node_left );
        subdivide( node_right );
    }
};

struct Tree { // KdTree (kdtree.h:23-32)
    Node* root = nullptr;
    std::vector<int> outOfTree, inTree;
    ~Tree() { delete root; }
};

inline Tree* init_kdtree( const mrt_scene_desc* desc ) { // kdtree.cpp:54-158
    Tree* tree = new Tree();
    Builder b( desc );
    for ( uint32_t i = 0; i < desc->n_objects; i++ ) {
        int kind = desc->obj_kind[i];
        if ( kind == MRT_PLANE ) { tree->outOfTree.push_back( i ); }
        else if ( kind == MRT_SPHERE ) { tree->inTree.push_back( i ); }
        else if ( kind == MRT_TRIANGLE ) { tree->inTree.push_back( i ); }
    }
    if ( tree->inTree.size() == 0 ) return tree;
    Node* root = new Node();

    std::vector<float> x_vector, y_vector, z_vector;
    for ( size_t i = 0; i < tree->inTree.size(); i++ ) {
        const Shape& s = b.shapes[tree->inTree[i]];
        if ( s.kind == MRT_SPHERE ) {
            float rad = s.radius;
            V3 pos    = s.xf.from( v3( 0.f ) );
            x_vector.push_back( pos.x + rad );
            x_vector.push_back( pos.x - rad );
            y_vector.push_back( pos.y + rad );
            y_vector.push_back( pos.y - rad );
            z_vector.push_back( pos.z + rad );
            z_vector.push_back( pos.z - rad );
        }
        else {
            V3 p[3] = { s.xf.from( s.p1 ), s.xf.from( s.p2 ), s.xf.from( s.p3 ) };
            for ( int j = 0; j < 3; j++ ) {
                x_vector.push_back( p[j].x );
                y_vector.push_back( p[j].y );
                z_vector.push_back( p[j].z );
            }
        }
    }
    float xmin = *std::min_element( x_vector.begin(), x_vector.end() );
    float ymin = *std::min_element( y_vector.begin(), y_vector.end() );
    float zmin = *std::min_element( z_vector.begin(), z_vector.end() );
    float xmax = *std::max_element( x_vector.begin(), x_vector.end() );
    float ymax = *std::max_element( y_vector.begin(), y_vector.end() );
    float zmax = *std::max_element( z_vector.begin(), z_vector.end() );

    root->min = v3( xmin, ymin, zmin ) - 1.f;
    root->max = v3( xmax, ymax, zmax ) + 1.f;
    for ( size_t i = 0; i < tree->inTree.size(); i++ )
        root->objects.push_back( tree->inTree[i] );
    tree->root = root;
    b.subdivide( root );
    return tree;
}

} // namespace orc
