// libmrt_hip.so -- the C-ABI of include/mrt.h on one MI355X (gfx950).
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared (see __graft_entry__.py).
// -ffp-contract=off is part of the contract, not a tuning flag: hit ids are bit-identical to the
// CPU reference only if no multiply-add is fused.
#include "../../include/mrt.h"
#include "device_render.hpp"
#include "device_trace.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <memory>
#include <mutex>
#include <new>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
namespace {

thread_local std::string g_err;

mrt_status fail( mrt_status st, const std::string& msg ) {
    g_err = msg;
    return st;
}

#define HIP_TRY( expr )                                                                                        \
    do {                                                                                                       \
        hipError_t e_ = ( expr );                                                                              \
        if ( e_ != hipSuccess )                                                                                \
            return fail( e_ == hipErrorOutOfMemory ? MRT_ERR_NOMEM : MRT_ERR_DEVICE,                            \
                         std::string( #expr ) + ": " + hipGetErrorString( e_ ) );                             \
    } while ( 0 )

template <typename T>
struct DevBuf {
    T* p     = nullptr;
    size_t n = 0;
    hipError_t alloc( size_t count ) {
        release();
        n = count;
        if ( count == 0 ) return hipSuccess;
        return hipMalloc( (void**)&p, count * sizeof( T ) );
    }
    hipError_t upload( const T* v, size_t count ) {
        hipError_t e = alloc( count );
        if ( e != hipSuccess || count == 0 ) return e;
        return hipMemcpy( p, v, count * sizeof( T ), hipMemcpyHostToDevice );
    }
    hipError_t upload( const std::vector<T>& v ) { return upload( v.data(), v.size() ); }
    void release() {
        if ( p ) (void)hipFree( p );
        p = nullptr;
        n = 0;
    }
    ~DevBuf() { release(); }
};

// Host staging array whose elements start uninitialised: the tables made at scene creation run to
// hundreds of megabytes and every element is written exactly once, so value-initialising them first
// (what std::vector does) would touch all that memory twice.
template <typename T>
struct HostBuf {
    T* p     = nullptr;
    size_t n = 0;
    HostBuf() = default;
    explicit HostBuf( size_t count ) : p( count ? (T*)std::malloc( count * sizeof( T ) ) : nullptr ), n( count ) {
        if ( count && !p ) throw std::bad_alloc();
    }
    HostBuf( const HostBuf& )            = delete;
    HostBuf& operator=( const HostBuf& ) = delete;
    HostBuf& operator=( HostBuf&& o ) noexcept {
        std::swap( p, o.p ), std::swap( n, o.n );
        return *this;
    }
    ~HostBuf() { std::free( p ); }
    T& operator[]( size_t i ) { return p[i]; }
    const T& operator[]( size_t i ) const { return p[i]; }
};

// body( begin, end ) over [0, n) cut into one contiguous range per thread (up to 8; one for small n)
template <class Body>
void parallel_ranges( size_t n, const Body& body ) {
    const size_t n_threads = n < ( 1u << 18 ) ? 1u : std::min<size_t>( 8u, std::max( 1u, std::thread::hardware_concurrency() ) );
    const size_t per       = ( n + n_threads - 1 ) / n_threads;
    std::vector<std::thread> workers;
    for ( size_t t = 1; t < n_threads; t++ )
        workers.emplace_back( [&body, n, per, t]() { body( std::min( n, t * per ), std::min( n, ( t + 1 ) * per ) ); } );
    body( 0, std::min( n, per ) );
    for ( std::thread& w : workers )
        w.join();
}

} // namespace

// ------------------------------------------------------------------------------------------------
// scene
// ------------------------------------------------------------------------------------------------
// Buffers of one relay (device_trace.hpp: Relay): the list of handed-over rays, one hit record per
// segment (traversals that report a hit) or one flag per ray (traversals that only answer "occluded").
#define MRT_RELAY_CAP ( 1u << 17 )
#define MRT_RELAY_AFTER_DEFAULT 256u
// mrt_intersect's per-scene scratch: 8 x u64 traversal counters, MRT_DEAL_WORDS chunk counters, the relay's
// counter (MRT_DEAL_STRIDE words apart from what follows), the relay launch's chunk counters
#define MRT_INTERSECT_SCRATCH_BYTES ( 8 * sizeof( uint64_t ) + ( 2 * MRT_DEAL_WORDS + MRT_DEAL_STRIDE ) * sizeof( uint32_t ) )
struct RelaySlot {
    DevBuf<uint2> list;
    DevBuf<float4> hits;
    DevBuf<uint32_t> flags;
};

struct mrt_scene {
    int device = 0;
    DevBuf<uint4> nodes;
    DevBuf<uint32_t> refs;
    DevBuf<float4> prims;
    DevBuf<uint2> prim_info;
    DevBuf<DXf> xforms;
    DevBuf<uint32_t> out_of_tree;
    DevBuf<uint32_t> object_of; // the caller's index of the object the library keeps at a given place (see mrt_scene_create)
    // shading tables
    DevBuf<float> shade_n;  // [n][9]
    DevBuf<float> shade_uv; // [n][6]
    DevBuf<DMaterial> materials;
    DevBuf<DTexture> textures;
    DevBuf<float> texels;
    DevBuf<DLight> lights;
    DevBuf<float> light_cdf;
    std::vector<int32_t> shade_light_kind; // host copy of mrt_light::kind (argument checks)
    DevBuf<float> env_rgb;
    DevBuf<float> env_func_int; // per-row integrals of the Distribution2D (funcInt of each conditional)
    DTrace trace {};
    DShade shade {};
    RenderWorkspace ws;
    // Calls on one handle share its scratch (d_stats, the render workspace).  Every call that uses
    // it waits on `order` when it starts and records it when it has queued its work, so calls made
    // one after another on DIFFERENT streams are still executed in call order.
    hipEvent_t order = nullptr;
    // mrt_render: side streams for the connection kernels, with the events that fork them from and
    // join them to the caller's stream
    hipStream_t side[2] = { nullptr, nullptr };
    hipEvent_t fork = nullptr, join[2] = { nullptr, nullptr };
    uint64_t* d_stats = nullptr; // mrt_intersect: 8 x u64 traversal counters, the MRT_DEAL_WORDS chunk counters, then the relay's counter and chunk counters
    // relay of long rays (device_trace.hpp: Relay): one set of buffers per traversal launch that can be
    // in flight at a time -- [0] continuation rays and mrt_intersect, [1] shadow rays, [2] MIS rays
    RelaySlot relay[3];
    uint64_t* d_relayed = nullptr; // rays relayed on this handle so far (mrt_debug_relayed)
    DevBuf<float> reduce_staging;  // mrt_reduce: where the other handles' images land on this (the root's) device
    uint32_t relay_after = MRT_RELAY_AFTER_DEFAULT;
    bool relay_on        = true;
    // How the first bounce's rays are traced: -1 = whichever of the two traversals is faster for this scene
    // and frame (timed once per image size on three bands of scanlines, see mrt_render), 0 = the streaming
    // traversal, 1 = as packets (trace_packets).  MRT_CAMERA_PACKETS=0 / 1 in the environment fixes it.
    int camera_mode      = -1;
    struct {
        uint32_t width = 0, height = 0, n_samples = 0;
        int packets = 0;
        float ms_stream = 0.f, ms_packets = 0.f;
    } camera_choice; // what was measured last
    bool intersect_packets = false; // tests: mrt_intersect's closest-hit rays go through trace_packets as well
    uint32_t tree_depth = 0;
    int n_cus           = 256;
    bool no_emitters    = false; // no object has a material with a non-zero emission component
    bool mixed_kinds    = false; // the objects' materials are of more than one shade kind (plain / textured / transparent)
    bool has_camera     = false;
    ~mrt_scene();
};

namespace {

// Walks the flattened tree on the host: every index must be in range, children must be stored
// after their parent (which also rules out cycles), no node may have two parents, and the depth must
// fit the traversal stack.  level[n] receives the parity of node n's depth (2 = not part of the tree).
// The top of the tree is walked breadth-first by the caller's thread until some hundreds of subtrees
// are pending; those are then walked depth-first by up to 8 threads.  A node is claimed by an atomic
// exchange on its level byte, so a second parent is noticed whichever threads the two parents are
// walked by (which also keeps a malformed description from making the walk exponential).
mrt_status validate_tree( const mrt_scene_desc* d, uint32_t* depth_out, std::vector<uint8_t>& level ) {
    *depth_out = 0;
    level.clear();
    if ( d->n_nodes == 0 ) return MRT_OK;
    if ( !d->nodes ) return fail( MRT_ERR_INVALID, "nodes is NULL" );
    if ( d->n_refs && !d->refs ) return fail( MRT_ERR_INVALID, "refs is NULL" );
    level.assign( d->n_nodes, 2 );
    level[0] = 0;
    std::mutex err_mu;
    const char* err_msg = nullptr; // the first problem found (string literals only)
    std::atomic<bool> stop { false };
    auto problem = [&]( const char* msg ) {
        std::lock_guard<std::mutex> lock( err_mu );
        if ( !err_msg ) err_msg = msg;
        stop.store( true, std::memory_order_relaxed );
        return false;
    };
    // checks one node; for an inner node claims its two children and returns the left one in *left
    auto visit = [&]( uint32_t node, uint32_t depth, uint32_t* left_out ) {
        *left_out           = 0;
        const mrt_kdnode& n = d->nodes[node];
        if ( ( n.w1 & 3u ) == 3u ) {
            const uint64_t off = n.w0, cnt = n.w1 >> 2;
            if ( off + cnt > d->n_refs ) return problem( "leaf reference range out of bounds" );
            for ( uint64_t i = 0; i < cnt; i++ ) {
                const uint32_t obj = d->refs[off + i];
                if ( obj >= d->n_objects ) return problem( "leaf references an object out of range" );
                if ( d->obj_kind[obj] == MRT_PLANE ) return problem( "a plane must not be inside the tree" );
            }
            return true;
        }
        const uint32_t left = n.w1 >> 2;
        if ( left <= node || (uint64_t)left + 1 >= d->n_nodes ) return problem( "child index out of range or not after its parent" );
        const uint8_t mark = (uint8_t)( ( depth + 1 ) & 1u );
        if ( __atomic_exchange_n( &level[left], mark, __ATOMIC_RELAXED ) != 2 || __atomic_exchange_n( &level[left + 1], mark, __ATOMIC_RELAXED ) != 2 )
            return problem( "a tree node has two parents" );
        *left_out = left;
        return true;
    };
    typedef std::pair<uint32_t, uint32_t> Entry; // node, depth
    std::vector<Entry> frontier;
    frontier.emplace_back( 0u, 0u );
    uint32_t depth_top = 0;
    size_t head = 0;
    while ( head < frontier.size() && frontier.size() - head < 1024 && !stop.load( std::memory_order_relaxed ) ) {
        const Entry e = frontier[head++];
        if ( e.second > depth_top ) depth_top = e.second;
        uint32_t left;
        if ( visit( e.first, e.second, &left ) && left ) {
            frontier.emplace_back( left, e.second + 1 );
            frontier.emplace_back( left + 1, e.second + 1 );
        }
    }
    frontier.erase( frontier.begin(), frontier.begin() + (std::ptrdiff_t)head ); // what is left are the pending subtrees
    std::atomic<size_t> next_subtree { 0 };
    std::atomic<uint32_t> depth_max { depth_top };
    auto walk = [&]() {
        std::vector<Entry> stack;
        uint32_t deepest = 0;
        for ( size_t k; ( k = next_subtree.fetch_add( 1 ) ) < frontier.size() && !stop.load( std::memory_order_relaxed ); ) {
            stack.push_back( frontier[k] );
            while ( !stack.empty() && !stop.load( std::memory_order_relaxed ) ) {
                const Entry e = stack.back();
                stack.pop_back();
                if ( e.second > deepest ) deepest = e.second;
                uint32_t left;
                if ( visit( e.first, e.second, &left ) && left ) {
                    stack.emplace_back( left, e.second + 1 );
                    stack.emplace_back( left + 1, e.second + 1 );
                }
            }
            stack.clear();
        }
        uint32_t seen = depth_max.load();
        while ( deepest > seen && !depth_max.compare_exchange_weak( seen, deepest ) ) {}
    };
    if ( !stop.load() && !frontier.empty() ) {
        const size_t n_threads = d->n_nodes < ( 1u << 18 ) ? 1u : std::min<size_t>( 8u, std::max( 1u, std::thread::hardware_concurrency() ) );
        std::vector<std::thread> workers;
        for ( size_t t = 1; t < n_threads; t++ )
            workers.emplace_back( walk );
        walk();
        for ( std::thread& w : workers )
            w.join();
    }
    if ( err_msg ) return fail( MRT_ERR_INVALID, err_msg );
    *depth_out = depth_max.load();
    if ( *depth_out + 2 > MRT_STACK_MAX ) return fail( MRT_ERR_UNSUPPORTED, "kd-tree deeper than the traversal stack (" + std::to_string( *depth_out ) + ")" );
    return MRT_OK;
}

// The tree as the kernels read it: one 32-byte record (two uint4) per node of even depth, holding
// the node and its two children (layout in device_trace.hpp); records follow the order of the
// caller's nodes.  Expects the depth parities validate_tree made of a tree that passed it.
//
// With `inline_lists` the records and the leaves' reference lists share one pool of 32-byte slots: the
// two records of a sibling pair are followed by the lists of the (up to four) leaves they hold -- every
// reference but a leaf's first, which rides in the record -- padded to a whole slot, then by the pair's
// subtrees.  A ray that has just fetched a record finds the list of the leaf it enters in the same or the
// next cache line instead of somewhere in a separate array of 4-byte indices that is as large as the tree.
// A leaf's w0 is then the pool word before its list (the kernels read words w0 + 1 .. w0 + count - 1 of
// DTrace::refs, which points at the pool), and DTrace::n_records counts records and list slots together.
mrt_status build_tree_records( const mrt_scene_desc* d, const std::vector<uint8_t>& level, bool inline_lists, bool align_groups, HostBuf<uint4>& nodes ) {
    nodes = HostBuf<uint4>();
    if ( d->n_nodes ) {
        HostBuf<uint32_t> record( d->n_nodes ); // the slot of the record of a node of even depth
        HostBuf<uint32_t> list_at;              // inline_lists: the pool word where the lists of that record's leaves start
        uint64_t n_slots = 0;
        // words the lists of the leaves held by the record of even-depth node i take
        auto list_words = [&]( uint32_t i ) {
            auto of_leaf = [&]( uint32_t n ) {
                const uint32_t w1 = d->nodes[n].w1;
                return ( w1 & 3u ) == 3u && ( w1 >> 2 ) > 1u ? (uint64_t)( w1 >> 2 ) - 1u : (uint64_t)0;
            };
            const uint32_t w1 = d->nodes[i].w1;
            if ( ( w1 & 3u ) == 3u ) return of_leaf( i );
            return of_leaf( w1 >> 2 ) + of_leaf( ( w1 >> 2 ) + 1u );
        };
        if ( !inline_lists ) {
            for ( uint32_t i = 0; i < d->n_nodes; i++ )
                if ( level[i] == 0 ) record[i] = (uint32_t)n_slots++;
        }
        else {
            list_at = HostBuf<uint32_t>( d->n_nodes );
            std::vector<uint8_t> first( d->n_nodes, 0 ); // the left one of a sibling pair of even depth
            first[0] = 1;
            parallel_ranges( d->n_nodes, [&]( size_t begin, size_t end ) {
                for ( size_t i = begin; i < end; i++ )
                    if ( level[i] == 1 && ( d->nodes[i].w1 & 3u ) != 3u ) first[d->nodes[i].w1 >> 2] = 1;
            } );
            for ( uint32_t i = 0; i < d->n_nodes; i++ ) {
                if ( level[i] != 0 || !first[i] ) continue;
                const bool pair = i != 0;
                if ( align_groups ) {
                    // a pair and its lists that fit one 128-byte line do not straddle two
                    const uint64_t group = ( pair ? 2u : 1u ) + ( list_words( i ) + ( pair ? list_words( i + 1 ) : 0u ) + 7u ) / 8u;
                    if ( group <= 4u && ( n_slots & 3u ) + group > 4u ) n_slots = ( n_slots + 3u ) & ~(uint64_t)3u;
                }
                record[i]       = (uint32_t)n_slots++;
                if ( pair ) record[i + 1] = (uint32_t)n_slots++;
                uint64_t words = n_slots * 8u;
                list_at[i]     = (uint32_t)words;
                words += list_words( i );
                if ( pair ) {
                    list_at[i + 1] = (uint32_t)words;
                    words += list_words( i + 1 );
                }
                n_slots = ( words + 7u ) / 8u;
                if ( n_slots >= ( 1u << 27 ) ) break;
            }
        }
        // a child as the traversal reads it: (w0, w1, first reference); `at`: where its list goes.  References name a
        // primitive by where its record lies (prim_loc)
        auto child = [&]( uint32_t n, uint32_t out[3], uint32_t& at ) {
            const uint32_t w0 = d->nodes[n].w0, w1 = d->nodes[n].w1;
            out[0] = w0, out[1] = w1, out[2] = 0;
            if ( ( w1 & 3u ) == 3u ) {
                const uint32_t cnt = w1 >> 2;
                if ( cnt > 0 ) out[2] = prim_loc( d->refs[w0] );
                if ( inline_lists ) {
                    out[0] = cnt > 1 ? at - 1u : 0u;
                    uint32_t* pool = (uint32_t*)nodes.p;
                    for ( uint32_t k = 1; k < cnt; k++ )
                        pool[at++] = prim_loc( d->refs[w0 + k] );
                }
            }
            else out[1] = ( w1 & 3u ) | ( record[w1 >> 2] << 2 ); // its children are adjacent nodes of even depth: adjacent records
        };
        // the kernels form a slot's byte offset (slot << 5) in 32 bits
        if ( n_slots >= ( 1u << 27 ) ) return fail( MRT_ERR_UNSUPPORTED, "kd-tree too large (limit: 2^27 nodes of even depth, 4 GB of records and reference lists)" );
        if ( inline_lists ) n_slots++; // the kernels fetch a list four references at a time: spare words behind the last list
        nodes = HostBuf<uint4>( 2 * (size_t)n_slots );
        if ( inline_lists ) std::memset( nodes.p, 0, 2 * (size_t)n_slots * sizeof( uint4 ) ); // the padding
        // every record (and its lists) is written by the one thread whose range holds its node
        parallel_ranges( d->n_nodes, [&]( size_t begin, size_t end ) {
            for ( size_t i = begin; i < end; i++ ) {
                if ( level[i] != 0 ) continue;
                const uint32_t w0 = d->nodes[i].w0, w1 = d->nodes[i].w1;
                uint32_t l[3] = { 0, 0, 0 }, r[3] = { 0, 0, 0 };
                uint32_t h0 = 0, h1 = 3;
                uint32_t at = inline_lists ? list_at[i] : 0u;
                if ( ( w1 & 3u ) == 3u ) child( (uint32_t)i, l, at ); // a leaf of even depth: no head step
                else {
                    h0 = w0, h1 = w1 & 3u;
                    child( w1 >> 2, l, at );
                    child( ( w1 >> 2 ) + 1, r, at );
                }
                nodes[2 * (size_t)record[i]]     = make_uint4( h0, h1, l[0], l[1] );
                nodes[2 * (size_t)record[i] + 1] = make_uint4( r[0], r[1], l[2], r[2] );
            }
        } );
    }
    return MRT_OK;
}

} // namespace

// One lane per block reports the XCC_ID hardware register (bits 3:0 of hardware register 20 on
// gfx942 / gfx950): which XCD the block was placed on.
__global__ void xcd_probe_kernel( uint32_t* __restrict__ out ) {
    if ( threadIdx.x == 0 ) out[blockIdx.x] = (uint32_t)__builtin_amdgcn_s_getreg( 20 | ( 0 << 6 ) | ( 3 << 11 ) );
}

namespace {
// blockIdx.x & mask = the XCD of a block, for the current device: 2^k - 1 when the first 64 blocks
// of a grid land on 2^k XCDs round-robin (7 on an unpartitioned MI355X, 3 or 1 or 0 in the
// partitioned modes), 0 when the placement follows no such rule.  The traversal kernels use it only
// to choose which part of a ray list a wave draws from first, so a wrong mask costs locality, never
// correctness.  Probed once per device.
mrt_status xcd_mask_for_device( int device, uint32_t* mask_out ) {
    static int cached[64];
    static bool have[64] = {};
    static std::mutex mu;
    std::lock_guard<std::mutex> lock( mu );
    if ( device >= 0 && device < 64 && have[device] ) {
        *mask_out = (uint32_t)cached[device];
        return MRT_OK;
    }
    constexpr int N = 64;
    DevBuf<uint32_t> buf;
    HIP_TRY( buf.alloc( N ) );
    HIP_TRY( hipMemset( buf.p, 0xff, N * sizeof( uint32_t ) ) );
    hipLaunchKernelGGL( xcd_probe_kernel, dim3( N ), dim3( 64 ), 0, 0, buf.p );
    HIP_TRY( hipGetLastError() );
    uint32_t id[N];
    HIP_TRY( hipMemcpy( id, buf.p, sizeof( id ), hipMemcpyDeviceToHost ) );
    uint32_t mask = 0;
    for ( int n = 8; n >= 2; n /= 2 ) {
        bool ok = true;
        for ( int a = 0; a < n && ok; a++ )
            for ( int b = 0; b < a && ok; b++ )
                ok = id[a] != id[b]; // n distinct XCDs ...
        for ( int b = n; b < N && ok; b++ )
            ok = id[b] == id[b % n]; // ... visited round-robin
        if ( ok ) {
            mask = (uint32_t)n - 1u;
            break;
        }
    }
    if ( device >= 0 && device < 64 ) cached[device] = (int)mask, have[device] = true;
    *mask_out = mask;
    return MRT_OK;
}
} // namespace

static mrt_status upload_shading( const mrt_scene_desc* d, mrt_scene* s ); // render_host.inc
static mrt_status validate_shading( const mrt_scene_desc* d );
namespace {
void free_workspace( RenderWorkspace& w );
mrt_status scratch_for_device( int device, uint32_t** out ); // with mrt_intersect_noalloc
}
mrt_scene::~mrt_scene() {
    if ( d_stats ) (void)hipFree( d_stats );
    if ( d_relayed ) (void)hipFree( d_relayed );
    if ( order ) (void)hipEventDestroy( order );
    if ( fork ) (void)hipEventDestroy( fork );
    for ( int k = 0; k < 2; k++ ) {
        if ( join[k] ) (void)hipEventDestroy( join[k] );
        if ( side[k] ) (void)hipStreamDestroy( side[k] );
    }
    free_workspace( ws );
}

extern "C" {

const char* mrt_last_error( void ) {
    return g_err.c_str();
}
int32_t mrt_version( void ) {
    return MRT_VERSION;
}

static mrt_status scene_create( const mrt_scene_desc* d, int device, mrt_scene** out );
mrt_status mrt_scene_create( const mrt_scene_desc* d, int device, mrt_scene** out ) {
    try { // the host-side tables are large: running out of memory is a status like any other
        return scene_create( d, device, out );
    }
    catch ( const std::bad_alloc& ) {
        if ( out ) *out = nullptr;
        return fail( MRT_ERR_NOMEM, "out of host memory while preparing the scene tables" );
    }
}
static mrt_status scene_create( const mrt_scene_desc* d, int device, mrt_scene** out ) {
    if ( !d || !out ) return fail( MRT_ERR_INVALID, "NULL argument" );
    *out = nullptr;
    if ( d->struct_size != sizeof( mrt_scene_desc ) ) return fail( MRT_ERR_INVALID, "mrt_scene_desc.struct_size mismatch" );
    if ( d->n_objects && ( !d->obj_kind || !d->obj_material || !d->obj_transform || !d->tri_p || !d->tri_n || !d->tri_uv || !d->sphere ) )
        return fail( MRT_ERR_INVALID, "object arrays are NULL" );
    if ( d->n_transforms == 0 || !d->transforms ) return fail( MRT_ERR_INVALID, "at least one transform (the identity, index 0) is required" );
    if ( d->n_materials == 0 || !d->materials ) return fail( MRT_ERR_INVALID, "at least one material is required" );
    // the kernels address the scene tables with 32-bit byte offsets, and a hit record keeps the
    // object index in 31 bits
    if ( d->n_objects >= ( 1u << 26 ) || d->n_transforms >= ( 1u << 26 ) ) return fail( MRT_ERR_UNSUPPORTED, "too many objects or transforms (limit 2^26)" );
    if ( d->n_nodes >= ( 1u << 28 ) || d->n_refs >= ( 1u << 30 ) ) return fail( MRT_ERR_UNSUPPORTED, "kd-tree too large (limits: 2^28 nodes, 2^30 leaf references)" );
    {
        static const float ident[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
        static const float zero[3]  = { 0, 0, 0 };
        const mrt_transform& t0     = d->transforms[0];
        if ( std::memcmp( t0.m, ident, 36 ) || std::memcmp( t0.inv, ident, 36 ) || std::memcmp( t0.t, zero, 12 ) )
            return fail( MRT_ERR_INVALID, "transform 0 must be the identity" );
    }
    if ( d->n_out_of_tree && !d->out_of_tree ) return fail( MRT_ERR_INVALID, "out_of_tree is NULL" );
    if ( d->n_textures && !d->textures ) return fail( MRT_ERR_INVALID, "textures is NULL" );
    if ( d->n_texels && !d->texels ) return fail( MRT_ERR_INVALID, "texels is NULL" );
    if ( d->n_lights && !d->lights ) return fail( MRT_ERR_INVALID, "lights is NULL" );
    uint32_t depth = 0;
    std::vector<uint8_t> level;
    mrt_status vs = validate_tree( d, &depth, level );
    if ( vs != MRT_OK ) return vs;
    for ( uint32_t i = 0; i < d->n_out_of_tree; i++ ) {
        if ( d->out_of_tree[i] >= d->n_objects ) return fail( MRT_ERR_INVALID, "out_of_tree index out of range" );
        // kdtree.cpp:64: planes, and only planes, stay outside the tree
        if ( d->obj_kind[d->out_of_tree[i]] != MRT_PLANE ) return fail( MRT_ERR_INVALID, "out_of_tree may list planes only" );
    }
    for ( uint32_t i = 0; i < d->n_objects; i++ ) {
        if ( d->obj_material[i] >= d->n_materials ) return fail( MRT_ERR_INVALID, "object material out of range" );
        if ( d->obj_transform[i] >= d->n_transforms ) return fail( MRT_ERR_INVALID, "object transform out of range" );
        if ( d->obj_kind[i] != MRT_TRIANGLE && d->obj_kind[i] != MRT_SPHERE && d->obj_kind[i] != MRT_PLANE ) return fail( MRT_ERR_UNSUPPORTED, "unsupported object kind" );
    }
    vs = validate_shading( d );
    if ( vs != MRT_OK ) return vs;

    // The library keeps the objects in the order in which the leaves first reference them (leaves in
    // node order, i.e. depth-first below the top of the tree), so that the primitive records a ray
    // pulls in one region of the tree lie together in memory whatever order the caller's mesh came
    // in; objects no leaf references (the planes) follow.  Every per-object table and every object
    // index of the description is rewritten accordingly; the order inside a leaf's list, which decides
    // ties, is untouched, and mrt_intersect translates hit indices back (object_of).
    // MRT_OBJECT_ORDER=input in the environment keeps the caller's order (for measurements).
    const uint32_t n_obj = d->n_objects;
    std::vector<uint32_t> place( n_obj ), object_of( n_obj );
    {
        const char* keep = std::getenv( "MRT_OBJECT_ORDER" );
        uint32_t next    = 0;
        if ( keep && std::strcmp( keep, "input" ) == 0 ) {
            for ( uint32_t o = 0; o < n_obj; o++ )
                place[o] = o;
        }
        else {
            std::fill( place.begin(), place.end(), 0xffffffffu );
            // every thread lists, in order, the objects first referenced within its range of nodes;
            // the lists are then merged in range order, which is the order one walk would have found
            std::vector<std::vector<uint32_t>> firsts;
            std::mutex firsts_mu;
            parallel_ranges( d->n_nodes, [&]( size_t begin, size_t end ) {
                std::vector<uint32_t> mine;
                std::vector<bool> seen( n_obj, false );
                for ( size_t i = begin; i < end; i++ ) {
                    const uint32_t w1 = d->nodes[i].w1;
                    if ( level[i] == 2 || ( w1 & 3u ) != 3u ) continue; // not part of the tree / not a leaf
                    for ( uint32_t k = 0; k < ( w1 >> 2 ); k++ ) {
                        const uint32_t o = d->refs[d->nodes[i].w0 + k];
                        if ( !seen[o] ) seen[o] = true, mine.push_back( o );
                    }
                }
                std::lock_guard<std::mutex> lock( firsts_mu );
                firsts.emplace_back( std::move( mine ) );
                firsts.back().push_back( (uint32_t)begin ); // the range's first node, to put the lists in order
            } );
            std::sort( firsts.begin(), firsts.end(), []( const std::vector<uint32_t>& a, const std::vector<uint32_t>& b ) { return a.back() < b.back(); } );
            for ( std::vector<uint32_t>& list : firsts ) {
                list.pop_back();
                for ( uint32_t o : list )
                    if ( place[o] == 0xffffffffu ) place[o] = next++;
            }
            for ( uint32_t o = 0; o < n_obj; o++ )
                if ( place[o] == 0xffffffffu ) place[o] = next++;
        }
        for ( uint32_t o = 0; o < n_obj; o++ )
            object_of[place[o]] = o;
    }
    HostBuf<uint8_t> r_kind( n_obj );
    HostBuf<uint32_t> r_material( n_obj ), r_transform( n_obj ), r_refs( d->n_refs ), r_out( d->n_out_of_tree );
    HostBuf<float> r_p( 9 * (size_t)n_obj ), r_n( 9 * (size_t)n_obj ), r_uv( 6 * (size_t)n_obj ), r_sphere( 4 * (size_t)n_obj );
    parallel_ranges( n_obj, [&]( size_t begin, size_t end ) {
        for ( size_t o = begin; o < end; o++ ) {
            const size_t t = place[o];
            r_kind[t] = d->obj_kind[o], r_material[t] = d->obj_material[o], r_transform[t] = d->obj_transform[o];
            std::memcpy( &r_p[9 * t], d->tri_p + 9 * o, 36 );
            std::memcpy( &r_n[9 * t], d->tri_n + 9 * o, 36 );
            std::memcpy( &r_uv[6 * t], d->tri_uv + 6 * o, 24 );
            std::memcpy( &r_sphere[4 * t], d->sphere + 4 * o, 16 );
        }
    } );
    parallel_ranges( d->n_refs, [&]( size_t begin, size_t end ) {
        for ( size_t i = begin; i < end; i++ )
            r_refs[i] = d->refs[i] < n_obj ? place[d->refs[i]] : 0u; // references outside every leaf's list are not validated (nor read)
    } );
    for ( uint32_t i = 0; i < d->n_out_of_tree; i++ )
        r_out[i] = place[d->out_of_tree[i]];
    mrt_scene_desc reordered = *d;
    reordered.obj_kind = r_kind.p, reordered.obj_material = r_material.p, reordered.obj_transform = r_transform.p;
    reordered.tri_p = r_p.p, reordered.tri_n = r_n.p, reordered.tri_uv = r_uv.p, reordered.sphere = r_sphere.p;
    reordered.refs = r_refs.p, reordered.out_of_tree = r_out.p;
    d = &reordered;

    // MRT_REFS_INLINE=0 keeps the leaves' reference lists in an array of their own, =1 packs the groups of records and lists
    // without regard to cache lines (measurements and tests)
    const char* inl = std::getenv( "MRT_REFS_INLINE" );
#if MRT_TOP_LDS
    const bool inline_lists = false; // (the staged top of the tree is addressed by record number)
#else
    const bool inline_lists = !( inl && std::strcmp( inl, "0" ) == 0 );
#endif
    const bool align_groups = inline_lists && !( inl && std::strcmp( inl, "1" ) == 0 ); // MRT_REFS_INLINE=1: groups packed without regard to cache lines
    HostBuf<uint4> nodes;
    vs = build_tree_records( d, level, inline_lists, align_groups, nodes );
    if ( vs != MRT_OK ) return vs;
    level = std::vector<uint8_t>();

    int n_dev = 0;
    HIP_TRY( hipGetDeviceCount( &n_dev ) );
    if ( device < 0 || device >= n_dev ) return fail( MRT_ERR_DEVICE, "no such HIP device" );
    HIP_TRY( hipSetDevice( device ) );
    int n_cus = 0;
    HIP_TRY( hipDeviceGetAttribute( &n_cus, hipDeviceAttributeMultiprocessorCount, device ) );
    {
        // set-up of mrt_intersect_noalloc: this device's address of the static scratch array
        uint32_t* unused = nullptr;
        mrt_status ss    = scratch_for_device( device, &unused );
        if ( ss != MRT_OK ) return ss;
    }

    bool no_emitters = true;
    uint32_t kinds_seen = 0; // bit k: some object's material is of shade kind k
    // records of the traversal (layout in device_trace.hpp) and (flags, material) per object for shading
    const size_t prim_bytes = (size_t)prim_offset( prim_loc( d->n_objects ? d->n_objects - 1u : 0u ) ) + 128u;
    std::vector<float4> prims( prim_bytes / 16u, make_float4( 0.f, 0.f, 0.f, 0.f ) );
    std::vector<uint2> prim_info( d->n_objects );
    for ( uint32_t i = 0; i < d->n_objects; i++ ) {
        uint32_t kind = d->obj_kind[i], mat = d->obj_material[i], xf = d->obj_transform[i];
        if ( mat >= d->n_materials ) return fail( MRT_ERR_INVALID, "object material out of range" );
        if ( xf >= d->n_transforms ) return fail( MRT_ERR_INVALID, "object transform out of range" );
        uint32_t k;
        if ( kind == MRT_TRIANGLE ) k = MRT_PRIM_TRI;
        else if ( kind == MRT_SPHERE ) k = MRT_PRIM_SPHERE;
        else if ( kind == MRT_PLANE ) k = MRT_PRIM_PLANE;
        else return fail( MRT_ERR_UNSUPPORTED, "unsupported object kind" );
        const float* e = d->materials[mat].emission;
        // kdtree.cpp:533-534: a shadow ray ignores a hit iff all three emission components are non-zero-tested
        uint32_t emissive = !( e[0] == 0.f && e[1] == 0.f && e[2] == 0.f );
        uint32_t em_sum   = !( e[0] + e[1] + e[2] == 0.f );
        if ( emissive || em_sum ) no_emitters = false;
        const mrt_material& mm = d->materials[mat];
        const uint32_t shade_kind = mm.type == MRT_MAT_TRANSPARENT ? MRT_SHADE_TRANSPARENT : ( mm.texture >= 0 ? MRT_SHADE_TEXTURED : MRT_SHADE_PLAIN );
        kinds_seen |= 1u << shade_kind;
        uint32_t flags    = k | ( emissive ? MRT_PRIM_EMISSIVE : 0u ) | ( em_sum ? MRT_PRIM_EMISSIVE_SUM : 0u ) | ( shade_kind << MRT_PRIM_SHADE_SHIFT ) | ( xf << MRT_PRIM_XF_SHIFT );
        const float* p    = d->tri_p + 9 * (size_t)i;
        // e1 = p2 - p1, e2 = p3 - p1: one fp32 subtraction each, as triangle.cpp:9-10 does per test
        const float rec[10] = { p[0], p[1], p[2], 0.f, p[3] - p[0], p[4] - p[1], p[5] - p[2], p[6] - p[0], p[7] - p[1], p[8] - p[2] };
        char* at = (char*)prims.data() + prim_offset( prim_loc( i ) );
        std::memcpy( at, rec, sizeof( rec ) );
        std::memcpy( at + 12, &flags, 4 );
        prim_info[i] = make_uint2( flags, mat );
    }
    std::vector<DXf> xforms( d->n_transforms );
    for ( uint32_t i = 0; i < d->n_transforms; i++ ) {
        std::memcpy( xforms[i].inv, d->transforms[i].inv, 36 );
        std::memcpy( xforms[i].t, d->transforms[i].t, 12 );
        std::memcpy( xforms[i].m, d->transforms[i].m, 36 );
        xforms[i].pad[0] = xforms[i].pad[1] = xforms[i].pad[2] = 0.f;
    }
    std::unique_ptr<mrt_scene> s( new mrt_scene );
    s->device     = device;
    s->tree_depth = depth;
    s->n_cus      = n_cus > 0 ? n_cus : 256;
    s->no_emitters = no_emitters;
    s->mixed_kinds = ( kinds_seen & ( kinds_seen - 1u ) ) != 0u;
    HIP_TRY( s->nodes.upload( nodes.p, nodes.n ) );
    if ( !inline_lists ) {
        parallel_ranges( d->n_refs, [&]( size_t begin, size_t end ) {
            for ( size_t i = begin; i < end; i++ )
                r_refs[i] = prim_loc( r_refs[i] );
        } );
        // (four spare words: the kernels fetch a list four references at a time)
        HIP_TRY( s->refs.alloc( (size_t)d->n_refs + 4u ) );
        HIP_TRY( hipMemset( s->refs.p, 0, ( (size_t)d->n_refs + 4u ) * sizeof( uint32_t ) ) );
        if ( d->n_refs ) HIP_TRY( hipMemcpy( s->refs.p, r_refs.p, (size_t)d->n_refs * sizeof( uint32_t ), hipMemcpyHostToDevice ) );
    }
    HIP_TRY( s->prims.upload( prims ) );
    HIP_TRY( s->prim_info.upload( prim_info ) );
    HIP_TRY( s->xforms.upload( xforms ) );
    HIP_TRY( s->out_of_tree.upload( d->out_of_tree, d->n_out_of_tree ) );
    HIP_TRY( s->object_of.upload( object_of ) );
    HIP_TRY( hipMalloc( (void**)&s->d_stats, MRT_INTERSECT_SCRATCH_BYTES ) );
    HIP_TRY( hipMalloc( (void**)&s->d_relayed, sizeof( uint64_t ) ) );
    HIP_TRY( hipMemset( s->d_relayed, 0, sizeof( uint64_t ) ) );
    {
        // MRT_RELAY=0 switches the relay of long rays off, MRT_RELAY_AFTER=<trips> sets how long a wave
        // goes on after its list ran out before it hands its rays over (measurements and tests)
        const char* e = std::getenv( "MRT_RELAY" );
        s->relay_on   = !( e && std::strcmp( e, "0" ) == 0 );
        if ( const char* a = std::getenv( "MRT_RELAY_AFTER" ) ) s->relay_after = (uint32_t)std::strtoul( a, nullptr, 10 );
        // MRT_CAMERA_PACKETS=1 / 0: the camera rays of a pixel always / never walk the tree together
        // (trace_packets; by default mrt_render times both traversals once and keeps the faster: packets win
        // where a triangle covers many pixels, the streaming traversal where triangles are about a pixel wide);
        // MRT_INTERSECT_PACKETS=1: mrt_intersect's closest-hit batches through the packet traversal (tests)
        e                    = std::getenv( "MRT_CAMERA_PACKETS" );
        s->camera_mode       = e && std::strcmp( e, "1" ) == 0 ? 1 : ( e && std::strcmp( e, "0" ) == 0 ? 0 : -1 );
        e                    = std::getenv( "MRT_INTERSECT_PACKETS" );
        s->intersect_packets = e && std::strcmp( e, "1" ) == 0;
        if ( s->relay_on )
            for ( int k = 0; k < 3; k++ ) {
                HIP_TRY( s->relay[k].list.alloc( MRT_RELAY_CAP ) );
                // the flags start at zero and every launch that used them leaves them at zero
                HIP_TRY( s->relay[k].flags.alloc( MRT_RELAY_CAP ) );
                HIP_TRY( hipMemset( s->relay[k].flags.p, 0, MRT_RELAY_CAP * sizeof( uint32_t ) ) );
                // hit records: continuation rays and mrt_intersect; MIS rays only where they report hits
                if ( k == 0 || ( k == 2 && !no_emitters ) ) HIP_TRY( s->relay[k].hits.alloc( (size_t)MRT_RELAY_CAP * MRT_RELAY_K ) );
            }
    }
    HIP_TRY( hipEventCreateWithFlags( &s->order, hipEventDisableTiming ) );
    HIP_TRY( hipEventRecord( s->order, 0 ) );
    HIP_TRY( hipEventCreateWithFlags( &s->fork, hipEventDisableTiming ) );
    for ( int k = 0; k < 2; k++ ) {
        HIP_TRY( hipStreamCreateWithFlags( &s->side[k], hipStreamNonBlocking ) );
        HIP_TRY( hipEventCreateWithFlags( &s->join[k], hipEventDisableTiming ) );
    }

    DTrace& t     = s->trace;
    t.nodes       = s->nodes.p;
    t.refs        = inline_lists ? (const uint32_t*)s->nodes.p : s->refs.p;
    t.prims       = s->prims.p;
    t.prim_info   = s->prim_info.p;
    t.xforms      = s->xforms.p;
    t.out_of_tree = s->out_of_tree.p;
    t.n_nodes     = d->n_nodes;
    t.n_records   = (uint32_t)( nodes.n / 2 );
    t.n_out       = d->n_out_of_tree;
    t.n_xforms    = d->n_transforms;
    {
        // MRT_MAILBOX=0: every primitive of every visited leaf is fetched and tested (measurements and tests)
        const char* e = std::getenv( "MRT_MAILBOX" );
        t.mailbox_on  = e && std::strcmp( e, "0" ) == 0 ? 0u : 1u;
    }
    {
        mrt_status xs = xcd_mask_for_device( device, &t.xcd_mask );
        if ( xs != MRT_OK ) return xs;
    }
    for ( int k = 0; k < 8; k++ )
        t.root[k] = 0;
    if ( nodes.n ) {
        const uint32_t r[8] = { nodes[0].x, nodes[0].y, nodes[0].z, nodes[0].w, nodes[1].x, nodes[1].y, nodes[1].z, nodes[1].w };
        std::memcpy( t.root, r, sizeof( r ) );
    }
    for ( int k = 0; k < 3; k++ ) {
        t.root_min[k] = d->root_min[k];
        t.root_max[k] = d->root_max[k];
    }

    mrt_status ss = upload_shading( d, s.get() );
    if ( ss != MRT_OK ) return ss;
    *out = s.release();
    return MRT_OK;
}

mrt_status mrt_scene_device_info( const mrt_scene* scene, mrt_device_info* out ) {
    if ( !scene || !out ) return fail( MRT_ERR_INVALID, "NULL argument" );
    out->device        = scene->device;
    out->compute_units = scene->n_cus;
    out->xcd_mask      = scene->trace.xcd_mask;
    out->kd_tree_depth = scene->tree_depth;
    return MRT_OK;
}

mrt_status mrt_debug_camera_choice( const mrt_scene* scene, int32_t* packets, float* ms_stream, float* ms_packets ) {
    if ( !scene || !packets ) return fail( MRT_ERR_INVALID, "NULL argument" );
    *packets = scene->camera_mode >= 0 ? scene->camera_mode : scene->camera_choice.packets;
    if ( ms_stream ) *ms_stream = scene->camera_choice.ms_stream;
    if ( ms_packets ) *ms_packets = scene->camera_choice.ms_packets;
    return MRT_OK;
}

mrt_status mrt_debug_relayed( mrt_scene* scene, uint64_t* out ) {
    if ( !scene || !out ) return fail( MRT_ERR_INVALID, "NULL argument" );
    HIP_TRY( hipSetDevice( scene->device ) );
    HIP_TRY( hipDeviceSynchronize() );
    HIP_TRY( hipMemcpy( out, scene->d_relayed, sizeof( uint64_t ), hipMemcpyDeviceToHost ) );
    return MRT_OK;
}

void mrt_scene_destroy( mrt_scene* scene ) {
    if ( !scene ) return;
    (void)hipSetDevice( scene->device );
    delete scene;
}

} // extern "C"

// ------------------------------------------------------------------------------------------------
// mrt_intersect
// ------------------------------------------------------------------------------------------------
struct BatchIO {
    static constexpr bool is_segment_io = false;
    const float *ox, *oy, *oz, *dx, *dy, *dz, *tmin, *tmax;
    int32_t* prim;
    float *t, *u, *v;
    const uint32_t* object_of; // the library's object order -> the caller's indices
    MRT_DEV void load( uint32_t i, uint32_t& tag, V3& o, V3& d, float& t0, float& t1 ) const {
        tag = i;
        o   = mk3( ox[i], oy[i], oz[i] );
        d   = mk3( dx[i], dy[i], dz[i] );
        t0  = tmin[i];
        t1  = tmax[i];
    }
    MRT_DEV void store( uint32_t i, bool, const DHit& h ) const {
        prim[i] = h.prim >= 0 ? (int32_t)object_of[h.prim] : h.prim;
        t[i]    = h.t;
        u[i]    = h.u;
        v[i]    = h.v;
    }
    MRT_DEV void reload( uint32_t i, V3& o, V3& d, float& t1 ) const {
        o  = mk3( ox[i], oy[i], oz[i] );
        d  = mk3( dx[i], dy[i], dz[i] );
        t1 = tmax[i];
    }
    struct Rays {
        const float *ox, *oy, *oz, *dx, *dy, *dz, *tmax;
        MRT_DEV void reload( uint32_t i, V3& o, V3& d, float& t1 ) const {
            o  = mk3( ox[i], oy[i], oz[i] );
            d  = mk3( dx[i], dy[i], dz[i] );
            t1 = tmax[i];
        }
    };
    Rays rays() const { return Rays { ox, oy, oz, dx, dy, dz, tmax }; }
};

template <bool ANY, bool STATS>
__global__ void __launch_bounds__( MRT_TRACE_BLOCK, MRT_TRACE_WAVES ) trace_batch_kernel( DTrace s, BatchIO io, uint32_t n, uint32_t* __restrict__ deal, unsigned long long* __restrict__ stats, Relay relay ) {
    __shared__ uint32_t lds[MRT_TRACE_LDS_WORDS( false )];
    lds_u32* lp = (lds_u32*)lds;
    DStats st { 0, 0, 0, 0 };
    trace_stream<ANY, false, STATS>( s, lp, io, n, deal, st, relay );
    if ( STATS ) flush_stats( st, stats );
}

__global__ void __launch_bounds__( MRT_TRACE_BLOCK, MRT_PACKET_WAVES ) trace_batch_packet_kernel( DTrace s, BatchIO io, uint32_t n, uint32_t* __restrict__ deal ) {
    __shared__ uint32_t lds[MRT_XF_LDS * 12];
    trace_packets( s, (lds_u32*)lds, io, n, deal );
}

// The follow-up launches of a traversal launch that ran with the relay `r` (device_trace.hpp: Relay)
// on `stream`: the segments of the handed-over rays, then their completion through the launch's own
// IO.  `deal`: MRT_DEAL_WORDS chunk counters, zero.  Both kernels read the number of entries from the
// device and end at once when there is none.
template <bool ANY, bool BOOL_ONLY, class IO>
static void launch_relay( const mrt_scene* sc, hipStream_t stream, const IO& io, const Relay& r, const RelaySlot& slot, uint32_t* deal ) {
    if ( !r.list ) return;
    typedef decltype( io.rays() ) RAYS;
    DTrace tr = sc->trace;
    tr.n_out  = 0; // the planes are merged by relay_finish_kernel
    SegmentIO<RAYS, BOOL_ONLY> sio;
    sio.rays  = io.rays();
    sio.list  = r.list;
    sio.hits  = slot.hits.p;
    sio.flags = slot.flags.p;
    for ( int k = 0; k < 3; k++ )
        sio.root_min[k] = tr.root_min[k], sio.root_max[k] = tr.root_max[k];
    auto kernel = relay_trace_kernel<ANY, BOOL_ONLY, RAYS>;
    hipLaunchKernelGGL( kernel, dim3( trace_grid_blocks( kernel, (uint64_t)r.cap * MRT_RELAY_K, sc->n_cus ) ), dim3( MRT_TRACE_BLOCK ), 0, stream, tr, sio, r.count, r.cap, deal );
    hipLaunchKernelGGL( ( relay_finish_kernel<ANY, BOOL_ONLY, IO> ), dim3( ( r.cap + 255u ) / 256u ), dim3( 256 ), 0, stream, sc->trace, io, r, slot.hits.p, slot.flags.p, (unsigned long long*)sc->d_relayed );
}

extern "C" mrt_status mrt_intersect( mrt_scene* sc,
                                     uint64_t n,
                                     const mrt_rays* rays,
                                     const mrt_hits* hits,
                                     uint32_t flags,
                                     uint32_t placement,
                                     void* stream_,
                                     mrt_trace_stats* stats ) {
    if ( !sc || !rays || !hits ) return fail( MRT_ERR_INVALID, "NULL argument" );
    if ( flags != MRT_CLOSEST_HIT && flags != MRT_ANY_HIT ) return fail( MRT_ERR_INVALID, "flags must be MRT_CLOSEST_HIT or MRT_ANY_HIT" );
    if ( placement != MRT_BUF_HOST && placement != MRT_BUF_DEVICE ) return fail( MRT_ERR_INVALID, "bad placement" );
    if ( n == 0 ) {
        if ( stats ) *stats = mrt_trace_stats {};
        return MRT_OK;
    }
    if ( n >= ( 1ull << 31 ) ) return fail( MRT_ERR_UNSUPPORTED, "batch larger than 2^31 rays" );
    const float* in[8] = { rays->ox, rays->oy, rays->oz, rays->dx, rays->dy, rays->dz, rays->tmin, rays->tmax };
    for ( const float* p : in )
        if ( !p ) return fail( MRT_ERR_INVALID, "NULL ray array" );
    if ( !hits->prim || !hits->t || !hits->u || !hits->v ) return fail( MRT_ERR_INVALID, "NULL hit array" );
    HIP_TRY( hipSetDevice( sc->device ) );
    hipStream_t stream = (hipStream_t)stream_;

    DevBuf<float> d_in, d_outf;
    DevBuf<int32_t> d_prim;
    const float* din[8];
    int32_t* dprim;
    float *dt, *du, *dv;
    if ( placement == MRT_BUF_HOST ) {
        HIP_TRY( d_in.alloc( 8 * n ) );
        HIP_TRY( d_outf.alloc( 3 * n ) );
        HIP_TRY( d_prim.alloc( n ) );
        for ( int k = 0; k < 8; k++ ) {
            HIP_TRY( hipMemcpyAsync( d_in.p + k * n, in[k], n * sizeof( float ), hipMemcpyHostToDevice, stream ) );
            din[k] = d_in.p + k * n;
        }
        dprim = d_prim.p;
        dt    = d_outf.p;
        du    = d_outf.p + n;
        dv    = d_outf.p + 2 * n;
    }
    else {
        for ( int k = 0; k < 8; k++ )
            din[k] = in[k];
        dprim = hits->prim;
        dt    = hits->t;
        du    = hits->u;
        dv    = hits->v;
    }
    HIP_TRY( hipStreamWaitEvent( stream, sc->order, 0 ) );
    HIP_TRY( hipMemsetAsync( sc->d_stats, 0, MRT_INTERSECT_SCRATCH_BYTES, stream ) );
    const dim3 block( MRT_TRACE_BLOCK );
    unsigned long long* ds = (unsigned long long*)sc->d_stats;
    uint32_t* const words  = (uint32_t*)( sc->d_stats + 8 ); // deal counters, relay counter (+ padding), the relay's deal counters
    const BatchIO io { din[0], din[1], din[2], din[3], din[4], din[5], din[6], din[7], dprim, dt, du, dv, sc->object_of.p };
    // Long closest-hit rays are relayed unless the visits are being counted (the counters are those of
    // the plain traversal).  Any-hit rays are not: which occluder they report -- the first one accepted
    // in a leaf's list order, not the nearest -- depends on the whole interval the ray has in that leaf
    // (see Relay; the renderer's shadow rays only report "occluded" and are relayed).
    Relay relay { nullptr, nullptr, 0u, 0u };
    if ( sc->relay_on && !stats && flags == MRT_CLOSEST_HIT ) relay = Relay { sc->relay[0].list.p, words + MRT_DEAL_WORDS, MRT_RELAY_CAP, sc->relay_after };
#define LAUNCH( A, S ) \
    hipLaunchKernelGGL( ( trace_batch_kernel<A, S> ), dim3( trace_grid_blocks( trace_batch_kernel<A, S>, n, sc->n_cus ) ), block, 0, stream, sc->trace, io, (uint32_t)n, words, ds, relay )
    if ( flags == MRT_ANY_HIT ) {
        if ( stats ) LAUNCH( true, true );
        else LAUNCH( true, false );
    }
    else if ( sc->intersect_packets && !stats ) {
        hipLaunchKernelGGL( trace_batch_packet_kernel, dim3( trace_grid_blocks( trace_batch_packet_kernel, n, sc->n_cus ) ), block, 0, stream, sc->trace, io, (uint32_t)n, words );
    }
    else {
        if ( stats ) LAUNCH( false, true );
        else LAUNCH( false, false );
        launch_relay<false, false>( sc, stream, io, relay, sc->relay[0], words + MRT_DEAL_WORDS + MRT_DEAL_STRIDE );
    }
#undef LAUNCH
    HIP_TRY( hipGetLastError() );
    if ( !stats ) HIP_TRY( hipEventRecord( sc->order, stream ) ); // with stats the call waits for the stream below
    if ( placement == MRT_BUF_HOST ) {
        HIP_TRY( hipMemcpyAsync( hits->prim, dprim, n * sizeof( int32_t ), hipMemcpyDeviceToHost, stream ) );
        HIP_TRY( hipMemcpyAsync( hits->t, dt, n * sizeof( float ), hipMemcpyDeviceToHost, stream ) );
        HIP_TRY( hipMemcpyAsync( hits->u, du, n * sizeof( float ), hipMemcpyDeviceToHost, stream ) );
        HIP_TRY( hipMemcpyAsync( hits->v, dv, n * sizeof( float ), hipMemcpyDeviceToHost, stream ) );
    }
    if ( stats ) {
        uint64_t h[8];
        HIP_TRY( hipMemcpyAsync( h, sc->d_stats, sizeof( h ), hipMemcpyDeviceToHost, stream ) );
        HIP_TRY( hipEventRecord( sc->order, stream ) );
        HIP_TRY( hipStreamSynchronize( stream ) );
        stats->inner      = h[0];
        stats->leaves     = h[1];
        stats->refs       = h[2];
        stats->prim_tests = h[3];
    }
    else if ( placement == MRT_BUF_HOST ) { HIP_TRY( hipStreamSynchronize( stream ) ); }
    return MRT_OK;
}

// ------------------------------------------------------------------------------------------------
// mrt_intersect_noalloc: mrt_intersect on device buffers for callers that cannot allocate.
//
// The chunk-deal counters of the streaming traversal (scratch that mrt_intersect keeps in a
// hipMalloc'ed per-scene buffer) live in a static device array instead.  Every GPU the library runs
// on holds its own copy of the array at its own address until the process ends; the address is
// looked up once per device (scratch_for_device, called from mrt_scene_create with that device
// current) and only that pointer, never the array's name, is used in host code.
// ------------------------------------------------------------------------------------------------
#define MRT_NOALLOC_SCRATCH_WORDS ( 16u + MRT_DEAL_WORDS ) // 8 x u64 traversal counters (unused here), then the deal counters
static __device__ __align__( 256 ) uint32_t scratch[MRT_NOALLOC_SCRATCH_WORDS];
static_assert( MRT_NOALLOC_SCRATCH_WORDS * sizeof( uint32_t ) == MRT_NOALLOC_SCRATCH_BYTES, "mrt.h states the size of the static scratch array" );

namespace {
constexpr int MRT_MAX_DEVICES = 64;
uint32_t* g_scratch_ptr[MRT_MAX_DEVICES] = {};
std::mutex g_scratch_mutex;

// Address of `scratch` on `device`, which must be the current device.
mrt_status scratch_for_device( int device, uint32_t** out ) {
    if ( device < 0 || device >= MRT_MAX_DEVICES ) return fail( MRT_ERR_UNSUPPORTED, "device index beyond the library's table (64)" );
    std::lock_guard<std::mutex> lock( g_scratch_mutex );
    if ( !g_scratch_ptr[device] ) {
        void* p = nullptr;
        HIP_TRY( hipGetSymbolAddress( &p, HIP_SYMBOL( scratch ) ) );
        if ( !p ) return fail( MRT_ERR_DEVICE, "hipGetSymbolAddress returned NULL for the static scratch array" );
        g_scratch_ptr[device] = (uint32_t*)p;
    }
    *out = g_scratch_ptr[device];
    return MRT_OK;
}
} // namespace

extern "C" mrt_status mrt_intersect_noalloc( mrt_scene* sc, uint64_t n, const mrt_rays* rays, const mrt_hits* hits, uint32_t flags, void* stream_ ) {
    if ( !sc || !rays || !hits ) return fail( MRT_ERR_INVALID, "NULL argument" );
    if ( flags != MRT_CLOSEST_HIT && flags != MRT_ANY_HIT ) return fail( MRT_ERR_INVALID, "flags must be MRT_CLOSEST_HIT or MRT_ANY_HIT" );
    if ( n == 0 ) return MRT_OK;
    if ( n > MRT_NOALLOC_MAX_RAYS ) return fail( MRT_ERR_UNSUPPORTED, "mrt_intersect_noalloc: batch larger than MRT_NOALLOC_MAX_RAYS (2^31 - 1) rays" );
    const float* in[8] = { rays->ox, rays->oy, rays->oz, rays->dx, rays->dy, rays->dz, rays->tmin, rays->tmax };
    for ( const float* p : in )
        if ( !p ) return fail( MRT_ERR_INVALID, "NULL ray array" );
    if ( !hits->prim || !hits->t || !hits->u || !hits->v ) return fail( MRT_ERR_INVALID, "NULL hit array" );
    HIP_TRY( hipSetDevice( sc->device ) );
    uint32_t* sp = nullptr;
    mrt_status ss = scratch_for_device( sc->device, &sp ); // a table look-up after mrt_scene_create
    if ( ss != MRT_OK ) return ss;
    hipStream_t stream = (hipStream_t)stream_;
    // the deal counters must start at zero; the array keeps the last call's contents
    HIP_TRY( hipMemsetAsync( sp, 0, MRT_NOALLOC_SCRATCH_WORDS * sizeof( uint32_t ), stream ) );
    const BatchIO io { in[0], in[1], in[2], in[3], in[4], in[5], in[6], in[7], hits->prim, hits->t, hits->u, hits->v, sc->object_of.p };
    const dim3 block( MRT_TRACE_BLOCK );
    if ( flags == MRT_ANY_HIT )
        hipLaunchKernelGGL( ( trace_batch_kernel<true, false> ), dim3( trace_grid_blocks( trace_batch_kernel<true, false>, n, sc->n_cus ) ), block, 0, stream, sc->trace, io, (uint32_t)n, sp + 16, (unsigned long long*)sp, Relay { nullptr, nullptr, 0u, 0u } );
    else
        hipLaunchKernelGGL( ( trace_batch_kernel<false, false> ), dim3( trace_grid_blocks( trace_batch_kernel<false, false>, n, sc->n_cus ) ), block, 0, stream, sc->trace, io, (uint32_t)n, sp + 16, (unsigned long long*)sp, Relay { nullptr, nullptr, 0u, 0u } );
    HIP_TRY( hipGetLastError() );
    return MRT_OK;
}

#include "render_host.inc"

#ifdef MRT_TRIP_HISTOGRAM
// measurement aid (variant builds only): copies out, and optionally clears, the histogram of trips per ray
extern "C" mrt_status mrt_debug_trip_histogram( unsigned long long* out32, int clear ) {
    HIP_TRY( hipDeviceSynchronize() );
    HIP_TRY( hipMemcpyFromSymbol( out32, HIP_SYMBOL( g_trip_hist ), 32 * sizeof( unsigned long long ) ) );
    if ( clear ) {
        unsigned long long zero[32] = {};
        HIP_TRY( hipMemcpyToSymbol( HIP_SYMBOL( g_trip_hist ), zero, sizeof( zero ) ) );
    }
    return MRT_OK;
}
#endif
