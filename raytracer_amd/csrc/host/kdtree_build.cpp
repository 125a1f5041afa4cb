// initKdTree for the GPU backend: the reference's SAH kd-tree construction (kdtree.cpp:54-464)
// producing the SAME tree -- node for node, split for split, leaf list for leaf list -- because
// the order of primitives inside leaves and the position of every split decide which primitive
// wins a tie, and hit ids must match the reference bit for bit.
//
// What is kept exactly: the event generation order, std::sort on the event position alone (the
// permutation among equal positions is whatever libstdc++'s introsort yields), the sweep whose
// index is advanced both inside the three while loops and by the for header (so one event after
// every group is skipped), the strict-interior candidate test, the float cost model
// lambda * (1 + 1.5 * (pl*nl + pr*nr)), termination at cost > 1.5 * N in double, distribution to
// children by closed bounding-box overlap, no depth limit.
//
// What is different from the reference: world-space boxes are computed once per object instead
// of once per node, axis and object; the three per-axis event lists are generated in one pass;
// subtrees are built by a pool of threads that always takes the largest node waiting (the SAH cuts
// off small pieces for many levels, so the order in which nodes are taken decides how long the
// threads stay busy), and in the large nodes near the root, where there are fewer nodes than
// threads, the three axes are sorted and swept as three jobs (each axis keeps its own best
// candidate; they are merged in axis order with the same strict comparison, so the outcome is that
// of the sequential loop).
#include "host_math.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <exception>
#include <mutex>
#include <omp.h>
#include <queue>
#include <thread>

namespace mrt {

KdTreeNode::~KdTreeNode() {
    delete left;
    delete right;
}
KdTree::~KdTree() {
    delete root;
    if ( device_scene ) mrt_scene_destroy( device_scene ); // created lazily by intersectKdTree / Pathtracer::preprocess
    delete flat;
}

namespace {

struct Event {
    float b;
    int type;
};
enum { STARTING = 0, LYING = 1, ENDING = 2 };
struct CompEvents { // the reference's comp_events (kdtree.cpp:311-313) as a functor the sort can inline
    bool operator()( const Event& i, const Event& j ) const { return i.b < j.b; }
};

struct Prim {
    int kind;
    vec3 bmin, bmax; // triangle: bounds of the world-space vertices; sphere: centre -/+ radius
    vec3 center;
    float radius;
};

float surfaceArea( const vec3& mn, const vec3& mx ) {
    float dx = mx.x - mn.x;
    float dy = mx.y - mn.y;
    float dz = mx.z - mn.z;
    return 2 * dx * dy + 2 * dx * dz + 2 * dy * dz;
}
float cost( int nl, int nr, float pl, float pr ) {
    float Kt = 1.0, Ki = 1.5;
    float lambda_p = ( nl == 0 || nr == 0 ) ? 0.8f : 1.f;
    return ( lambda_p * ( Kt + Ki * ( pl * nl + pr * nr ) ) );
}
float sah( const vec3& mn, const vec3& mx, float area, float p, int nl, int nr, int np, int k ) {
    vec3 max_vl = mx, min_vr = mn;
    max_vl[k] = p;
    min_vr[k] = p;
    float pl  = surfaceArea( mn, max_vl ) / area;
    float pr  = surfaceArea( min_vr, mx ) / area;
    float cpl = cost( nl + np, nr, pl, pr );
    float cpr = cost( nl, nr + np, pl, pr );
    return ( cpl < cpr ) ? cpl : cpr;
}

struct Builder {
    std::vector<Prim> prims;
    size_t node_cutoff = 4096;  // a node with fewer objects is built, subtree and all, by the thread that made it
    size_t axis_cutoff = 32768; // nodes from this size on sort and sweep their three axes as three jobs

    void clipped( const Prim& p, const vec3& mn, const vec3& mx, vec3& minb, vec3& maxb ) const {
        minb = p.bmin;
        maxb = p.bmax;
        for ( int k = 0; k < 3; k++ ) {
            if ( mn[k] > minb[k] ) minb[k] = mn[k];
            if ( mx[k] < maxb[k] ) maxb[k] = mx[k];
        }
    }

    // kdtree.cpp:316-390
    // events of the node's objects, per axis, in the reference's generation order
    void generate( const KdTreeNode* node, std::vector<Event>* ev ) const {
        const size_t n_obj = node->objects.size();
        for ( int k = 0; k < 3; k++ )
            ev[k].resize( 2 * n_obj );
        Event *e0 = ev[0].data(), *e1 = ev[1].data(), *e2 = ev[2].data();
        size_t m = 0;
        for ( int obj : node->objects ) {
            vec3 minb, maxb;
            clipped( prims[obj], node->min, node->max, minb, maxb );
            float dx = maxb.x - minb.x, dy = maxb.y - minb.y, dz = maxb.z - minb.z;
            bool planar = dx <= 0.001 || dy <= 0.001 || dz <= 0.001;
            if ( planar ) {
                e0[m] = Event { minb.x, LYING }, e1[m] = Event { minb.y, LYING }, e2[m] = Event { minb.z, LYING };
                m++;
            }
            else {
                e0[m] = Event { minb.x, STARTING }, e1[m] = Event { minb.y, STARTING }, e2[m] = Event { minb.z, STARTING };
                e0[m + 1] = Event { maxb.x, ENDING }, e1[m + 1] = Event { maxb.y, ENDING }, e2[m + 1] = Event { maxb.z, ENDING };
                m += 2;
            }
        }
        for ( int k = 0; k < 3; k++ )
            ev[k].resize( m );
    }
    // sort + sweep of one axis: the first candidate of lowest cost on that axis
    void sweep( const KdTreeNode* node, std::vector<Event>& events, int k, float area, float& p_, float& c_ ) const {
        c_ = INFINITY;
        p_ = 0;
        std::sort( events.begin(), events.end(), CompEvents() );
        const size_t n = events.size();
        int nl = 0, np = 0, nr = (int)node->objects.size();
        for ( size_t i = 0; i < n; i++ ) {
            float p       = events[i].b;
            int pStarting = 0, pEnding = 0, pLying = 0;
            while ( i < n && events[i].b == p && events[i].type == ENDING ) {
                pEnding++;
                i++;
            }
            while ( i < n && events[i].b == p && events[i].type == LYING ) {
                pLying++;
                i++;
            }
            while ( i < n && events[i].b == p && events[i].type == STARTING ) {
                pStarting++;
                i++;
            }
            np = pLying;
            nr -= pLying;
            nr -= pEnding;
            float c = sah( node->min, node->max, area, p, nl, nr, np, k );
            if ( c < c_ && !( p <= node->min[k] ) && !( p >= node->max[k] ) ) {
                c_ = c;
                p_ = p;
            }
            nl += pStarting;
            nl += pLying;
            np = 0;
        }
    }

    static bool boxOverlap( const vec3& mn, const vec3& mx, const vec3& aMin, const vec3& aMax ) {
        return !( ( mn.x > aMax.x ) || ( mx.x < aMin.x ) || ( mn.y > aMax.y ) || ( mx.y < aMin.y ) || ( mn.z > aMax.z ) || ( mx.z < aMin.z ) );
    }
    static bool sphereOverlap( const vec3& c, float r, const vec3& aMin, const vec3& aMax ) {
        vec3 q;
        for ( int k = 0; k < 3; k++ ) {
            float v = ( c[k] < aMin[k] ) ? aMin[k] : c[k]; // max(c, aMin)
            q[k]    = ( aMax[k] < v ) ? aMax[k] : v;       // min(., aMax)
        }
        vec3 seg = q - c;
        return dot( seg, seg ) < ( r * r );
    }

    // ---------------------------------------------------------------- the pool
    // Two kinds of work: nodes to split (largest first) and single axes of a large node (before any
    // node: another thread is waiting for them).  `open` counts the nodes queued or being split;
    // the build is over when it reaches zero.
    struct AxisJob {
        const KdTreeNode* node;
        std::vector<Event>* events;
        int k;
        float area;
        float *p, *c;
        std::atomic<int>* left;
    };
    struct BySize {
        bool operator()( const KdTreeNode* a, const KdTreeNode* b ) const { return a->objects.size() < b->objects.size(); }
    };
    std::mutex m;
    std::condition_variable wake;
    std::priority_queue<KdTreeNode*, std::vector<KdTreeNode*>, BySize> nodes_waiting;
    std::vector<AxisJob> axes_waiting;
    size_t open = 0;
    bool parallel = false;
    std::exception_ptr failure;

    void run_axis( const AxisJob& j ) const {
        sweep( j.node, *j.events, j.k, j.area, *j.p, *j.c );
        j.left->fetch_sub( 1, std::memory_order_release );
    }

    void findPlane( const KdTreeNode* node, float& p_, int& k_, float& c_ ) {
        static thread_local std::vector<Event> ev_small[3];
        const float area = surfaceArea( node->min, node->max );
        float p[3], c[3];
        if ( parallel && node->objects.size() >= axis_cutoff ) {
            // its own event lists: while this thread waits for the other two axes it runs axis jobs
            // of other nodes, and they must not share its lists
            std::vector<Event> ev[3];
            generate( node, ev );
            std::atomic<int> left( 2 );
            {
                std::lock_guard<std::mutex> g( m );
                axes_waiting.push_back( AxisJob { node, &ev[1], 1, area, &p[1], &c[1], &left } );
                axes_waiting.push_back( AxisJob { node, &ev[2], 2, area, &p[2], &c[2], &left } );
            }
            wake.notify_all();
            sweep( node, ev[0], 0, area, p[0], c[0] );
            // Help until both are done.  Its own two jobs are in the list until somebody takes
            // them, so a thread that finds the list empty only waits for jobs already running.
            while ( left.load( std::memory_order_acquire ) != 0 ) {
                AxisJob j {};
                bool got = false;
                {
                    std::lock_guard<std::mutex> g( m );
                    if ( !axes_waiting.empty() ) {
                        j = axes_waiting.back();
                        axes_waiting.pop_back();
                        got = true;
                    }
                }
                if ( got ) run_axis( j );
                else std::this_thread::yield();
            }
        }
        else {
            generate( node, ev_small );
            for ( int k = 0; k < 3; k++ )
                sweep( node, ev_small[k], k, area, p[k], c[k] );
        }
        c_ = INFINITY;
        p_ = 0;
        for ( int k = 0; k < 3; k++ )
            if ( c[k] < c_ ) c_ = c[k], p_ = p[k], k_ = k;
    }

    // kdtree.cpp:392-464: one node; false when it became a leaf
    bool split( KdTreeNode* node ) {
        float p = 0, c = 0;
        int axis = 0;
        findPlane( node, p, axis, c );
        if ( c == INFINITY || c > 1.5 * node->objects.size() ) {
            node->leaf = true;
            return false;
        }
        KdTreeNode* l = new KdTreeNode();
        KdTreeNode* r = new KdTreeNode();
        l->depth = r->depth = node->depth + 1;
        l->min = r->min = node->min;
        l->max = r->max = node->max;
        node->split     = p;
        node->axis      = axis;
        l->max[axis]    = p;
        r->min[axis]    = p;
        node->left      = l;
        node->right     = r;
        // two passes, so that the children's lists are allocated once at their final size
        static thread_local std::vector<unsigned char> side;
        const size_t n_obj = node->objects.size();
        side.resize( n_obj );
        size_t n_l = 0, n_r = 0;
        for ( size_t i = 0; i < n_obj; i++ ) {
            const Prim& pr = prims[node->objects[i]];
            bool inl, inr;
            if ( pr.kind == SPHERE ) {
                inl = sphereOverlap( pr.center, pr.radius, l->min, l->max );
                inr = sphereOverlap( pr.center, pr.radius, r->min, r->max );
            }
            else {
                inl = boxOverlap( pr.bmin, pr.bmax, l->min, l->max );
                inr = boxOverlap( pr.bmin, pr.bmax, r->min, r->max );
            }
            side[i] = (unsigned char)( ( inl ? 1 : 0 ) | ( inr ? 2 : 0 ) );
            n_l += inl, n_r += inr;
        }
        l->objects.resize( n_l );
        r->objects.resize( n_r );
        {
            int *pl = l->objects.data(), *pr = r->objects.data();
            const int* src = node->objects.data();
            for ( size_t i = 0; i < n_obj; i++ ) {
                if ( side[i] & 1 ) *pl++ = src[i];
                if ( side[i] & 2 ) *pr++ = src[i];
            }
        }
        std::vector<int>().swap( node->objects );
        return true;
    }
    void subtree( KdTreeNode* node ) { // by this thread alone
        if ( !split( node ) ) return;
        subtree( node->left );
        subtree( node->right );
    }

    void worker() {
        std::unique_lock<std::mutex> lock( m );
        for ( ;; ) {
            if ( failure ) return;
            if ( !axes_waiting.empty() ) {
                AxisJob j = axes_waiting.back();
                axes_waiting.pop_back();
                lock.unlock();
                run_axis( j ); // sorts and compares: nothing in it throws
                lock.lock();
                continue;
            }
            if ( !nodes_waiting.empty() ) {
                KdTreeNode* node = nodes_waiting.top();
                nodes_waiting.pop();
                lock.unlock();
                KdTreeNode* kids[2] = { nullptr, nullptr };
                try {
                    if ( node->objects.size() < node_cutoff ) subtree( node );
                    else if ( split( node ) ) kids[0] = node->left, kids[1] = node->right;
                }
                catch ( ... ) {
                    lock.lock();
                    if ( !failure ) failure = std::current_exception();
                    wake.notify_all();
                    return;
                }
                lock.lock();
                for ( KdTreeNode* k : kids )
                    if ( k ) {
                        nodes_waiting.push( k );
                        open++;
                    }
                open--;
                if ( open == 0 || kids[0] ) wake.notify_all();
                continue;
            }
            if ( open == 0 ) return;
            wake.wait( lock );
        }
    }

    void build( KdTreeNode* root, int n_threads ) {
        if ( n_threads <= 1 ) {
            parallel = false;
            subtree( root );
            return;
        }
        parallel = true;
        nodes_waiting.push( root );
        open = 1;
        std::vector<std::thread> pool;
        for ( int t = 1; t < n_threads; t++ )
            pool.emplace_back( [this] { worker(); } );
        worker();
        for ( std::thread& t : pool )
            t.join();
        if ( failure ) std::rethrow_exception( failure );
    }
};

} // namespace

KdTree* initKdTree( Scene* scene, int n_threads ) {
    KdTree* tree = new KdTree();
    Builder b;
    b.prims.resize( scene->objects.size() );
    for ( size_t i = 0; i < scene->objects.size(); i++ ) {
        const Object* o = scene->objects[i];
        Prim& p         = b.prims[i];
        p.kind          = o->geom.type;
        if ( o->geom.type == PLANE ) { tree->outOfTree.push_back( (int)i ); }
        else if ( o->geom.type == SPHERE ) {
            tree->inTree.push_back( (int)i );
            p.center = o->geom.sphere.center;
            p.radius = o->geom.sphere.radius;
            p.bmin   = vec3( p.center.x - p.radius, p.center.y - p.radius, p.center.z - p.radius );
            p.bmax   = vec3( p.center.x + p.radius, p.center.y + p.radius, p.center.z + p.radius );
        }
        else if ( o->geom.type == TRIANGLE ) {
            tree->inTree.push_back( (int)i );
            vec3 v0 = o->transform.transformFrom( o->geom.triangle.p1 );
            vec3 v1 = o->transform.transformFrom( o->geom.triangle.p2 );
            vec3 v2 = o->transform.transformFrom( o->geom.triangle.p3 );
            p.bmin  = vec3( std::min( v0.x, std::min( v1.x, v2.x ) ), std::min( v0.y, std::min( v1.y, v2.y ) ), std::min( v0.z, std::min( v1.z, v2.z ) ) );
            p.bmax  = vec3( std::max( v0.x, std::max( v1.x, v2.x ) ), std::max( v0.y, std::max( v1.y, v2.y ) ), std::max( v0.z, std::max( v1.z, v2.z ) ) );
        }
    }
    tree->objLimit = scene->objects.size() * sizeof( Object );
    if ( tree->inTree.empty() ) return tree;

    // Root box (kdtree.cpp:79-146).  A sphere contributes the centre recomputed through its
    // transform +/- the radius; a triangle its three world-space vertices.
    bool first = true;
    vec3 mn, mx;
    for ( int i : tree->inTree ) {
        const Object* o = scene->objects[i];
        vec3 lo, hi;
        if ( o->geom.type == SPHERE ) {
            vec3 c  = o->transform.transformFrom( vec3( 0.f ) );
            float r = o->geom.sphere.radius;
            lo      = vec3( c.x - r, c.y - r, c.z - r );
            hi      = vec3( c.x + r, c.y + r, c.z + r );
        }
        else {
            lo = b.prims[i].bmin;
            hi = b.prims[i].bmax;
        }
        if ( first ) {
            mn    = lo;
            mx    = hi;
            first = false;
        }
        else {
            for ( int k = 0; k < 3; k++ ) {
                if ( lo[k] < mn[k] ) mn[k] = lo[k];
                if ( mx[k] < hi[k] ) mx[k] = hi[k];
            }
        }
    }
    KdTreeNode* root = new KdTreeNode();
    root->min        = mn - 1.f;
    root->max        = mx + 1.f;
    root->objects.assign( tree->inTree.begin(), tree->inTree.end() );
    tree->root = root;

    if ( n_threads < 1 ) n_threads = omp_get_max_threads(); // honours OMP_NUM_THREADS, like the rest of the host library
    b.build( root, n_threads );
    return tree;
}

KdTree* initKdTree( Scene* scene ) {
    return initKdTree( scene, 0 );
}

} // namespace mrt
