// Wavefront path tracer for gfx950: [extend -> shade -> connect]* -> accumulate (the first extend generates the camera rays).
//
// It computes what CPURenderer::run does for one camera sample (rendering/CPURenderer.cpp:46-62):
// camera sample, SimpleCamera::get_ray, then Pathtracer::trace_ray (integrator.cpp:128-211) with
// directIllumination (integrator.cpp:69-120).  One path = one camera sample; all per-path state is
// struct-of-arrays in HBM, indexed by the path id (pass_path below: pixels tile by tile, the
// samples of a pixel together), so that the 64 lanes of a wave hold 64 samples of one pixel
// (coherent rays, coalesced state access) and consecutive chunks the neighbouring pixels.  Work lists are queues of path ids; the kernels are
//   extend          closest-hit traversal of the continuation ray of every live path; the first
//                   one generates the camera rays as it goes (camera sample + SimpleCamera::get_ray)
//   shade           everything of one loop iteration of trace_ray that needs no further ray:
//                   emission at depth 0, light choice, light sample + BSDF eval, BSDF sample for
//                   MIS, BSDF sample for the continuation, throughput update; emits one shadow
//                   ray and (when the light pdf is non-zero) one MIS ray, and queues survivors
//   connect_shadow  any-hit traversal of the shadow rays; an occluded ray cancels its contribution
//   connect_mis     closest-hit traversal of the MIS rays; adds bsdf * Li * w / pdf
//   accumulate      folds the last pending direct-lighting term of every path and adds the
//                   per-sample radiances of a pixel, in sample order, to the image sum
// Dead paths are retired by wave ballot + one atomic per wave when the next queue is written.
//
// Order of the float additions into a path's radiance follows the reference: depth-0 emission,
// then per bounce (throughput * DI) / lightPdf with DI = (0 + light term) + bsdf term, finally
// throughput * Le on a miss.  A bounce's DI is complete only after both connect kernels, so it is
// kept pending and folded in by the next shade of the path or by accumulate.
#pragma once
#include "device_shading.hpp"

// Rays and hits are 16-byte records (one or two wide loads/stores per path instead of up to
// eight narrow ones, and whole 32-byte sectors used when the path ids of a wave are scattered);
// everything else is one 32-bit array per field.
//   ray records   r0 = (o.x, o.y, o.z, d.x)   r1 = (d.y, d.z, tmax, 0)
//   hit record    (prim | outside << 31 as bits, or 0xffffffff for a miss; t; u; v)
struct PathState {
    // continuation ray (tmin = 0; tmax = 1e5 for camera rays, 1e4 afterwards) and its closest hit
    float4 *ray0, *ray1;
    float4* hit;
    // shadow ray (its contribution is already in di; cancelled if occluded) and MIS ray
    float4 *sray0, *sray1;
    float4 *mray0, *mray1;
    // path
    float *tpx, *tpy, *tpz; // throughput
    float *Lx, *Ly, *Lz;    // radiance so far
    uint64_t* rng_state; // PCG32 state of the sample's engine stream (its increment follows from the sample's key)
    // pending direct lighting of the last shaded bounce: L += (ptp * di) / (the light pick's pdf, a
    // constant of the scene: UniformLightDistribution, LightDistribution.cpp:5-12); ptp is not stored
    // for a path's first bounce, where it is (1, 1, 1) (MRT_PENDING_TP_ONE)
    float *ptx, *pty, *ptz, *dix, *diy, *diz;
    uint32_t* pending;
    // factors of the MIS ray's contribution; in a scene without emissive objects (mb) is the whole
    // contribution the ray makes if it escapes, and mw, mpdf, mlight are unused
    float *mbx, *mby, *mbz, *mw, *mpdf;
    uint32_t* mlight;
};
#define MRT_PATH_WORDS 49

MRT_DEV void put_ray( float4* r0, float4* r1, uint32_t id, V3 o, V3 d, float tmax ) {
    r0[id] = make_float4( o.x, o.y, o.z, d.x );
    r1[id] = make_float4( d.y, d.z, tmax, 0.f );
}
MRT_DEV void get_ray( const float4* r0, const float4* r1, uint32_t id, V3& o, V3& d, float& tmax ) {
    const float4 a = r0[id], b = r1[id];
    o    = mk3( a.x, a.y, a.z );
    d    = mk3( a.w, b.x, b.y );
    tmax = b.z;
}
MRT_DEV void put_hit( float4* hit, uint32_t id, const DHit& h ) {
    hit[id] = pack_hit( h );
}
MRT_DEV void get_hit( const float4* hit, uint32_t id, DHit& h ) {
    unpack_hit( hit[id], h );
}
// The rays of one of the three ray record pairs, by path id (what a relay launch reads them through).
struct PathRays {
    const float4 *r0, *r1;
    MRT_DEV void reload( uint32_t id, V3& o, V3& d, float& tmax ) const { get_ray( r0, r1, id, o, d, tmax ); }
};

struct PassInfo {
    uint32_t width;
    uint32_t row_begin, n_rows;      // pixel_local = (y - row_begin) * width + x
    uint32_t sample_begin, n_samples;
    uint32_t spp_x, spp_y;
    uint32_t seed;
    uint32_t n_pixels;               // n_rows * width / tile_step
    uint32_t n_paths;                // n_pixels * n_samples
    uint32_t max_depth;
    // tile_step > 1: the pass holds only the 8x8 tiles (tx, ty) of its rows with (tx + ty) % tile_step ==
    // tile_phase (ty counted from the top of the IMAGE); width is then a multiple of 8 * tile_step and the
    // pass is made of whole tiles
    uint32_t tile_step, tile_phase;
};

// A pass made of whole 8x8 pixel tiles numbers its pixels and paths tile by tile.
#define MRT_TILE_STRIPS 4u
MRT_DEV bool pass_tiled( const PassInfo& pi ) {
#ifndef MRT_NO_TILES
    return ( ( pi.width | pi.n_rows ) & 7u ) == 0u;
#else
    return false;
#endif
}

// Pixel of the pass that path-local index p_local stands for.  When the pass is made of whole 8x8
// tiles the pixels are numbered tile by tile, so that 64 consecutive paths (a wave's first rays, a
// queue's neighbours) cover a square of the image instead of a 64-pixel line: their rays start
// closer together and walk more of the same subtrees.  Otherwise scanline order.
MRT_DEV void pass_pixel( const PassInfo& pi, uint32_t p_local, uint32_t& x, uint32_t& y ) {
    if ( pi.tile_step > 1u ) {
        // a share of the tiles, row of tiles by row of tiles: every row holds width / 8 / tile_step of them,
        // tile_step tiles apart, starting where (tx + ty) % tile_step == tile_phase
        const uint32_t tile = p_local >> 6, in = p_local & 63u, per_row = ( pi.width >> 3 ) / pi.tile_step;
        const uint32_t ty_local = tile / per_row, j = tile - ty_local * per_row;
        const uint32_t ty = ( pi.row_begin >> 3 ) + ty_local;
        const uint32_t tx = ( pi.tile_phase + pi.tile_step - ty % pi.tile_step ) % pi.tile_step + j * pi.tile_step;
        x = ( tx << 3 ) + ( in & 7u );
        y = ( ty << 3 ) + ( in >> 3 );
        return;
    }
    if ( pass_tiled( pi ) ) {
        const uint32_t tile = p_local >> 6, in = p_local & 63u, tiles_x = pi.width >> 3;
        uint32_t ty = tile / tiles_x, tx = tile - ty * tiles_x;
        if ( ( tiles_x & ( MRT_TILE_STRIPS - 1u ) ) == 0u ) {
            // tiles run down vertical strips MRT_TILE_STRIPS tiles wide: a tile's successors are
            // its neighbours in both directions, not only along the scanline
            const uint32_t tiles_y = pi.n_rows >> 3, per_strip = tiles_y * MRT_TILE_STRIPS;
            const uint32_t strip = tile / per_strip, r = tile - strip * per_strip;
            ty = r / MRT_TILE_STRIPS;
            tx = strip * MRT_TILE_STRIPS + ( r - ty * MRT_TILE_STRIPS );
        }
        x = ( tx << 3 ) + ( in & 7u );
        y = pi.row_begin + ( ty << 3 ) + ( in >> 3 );
        return;
    }
    const uint32_t r = p_local / pi.width;
    x = p_local - r * pi.width;
    y = pi.row_begin + r;
}

// Path id of sample s_local of pixel p_local, and back.  In a tiled pass the samples of one pixel
// are numbered together and the pixels tile by tile: (pixel, sample).  A 64-path chunk is then
// 64 samples of one pixel - rays that start at (nearly) the same point and descend the tree
// together - and the rays in flight at any time come from a few tiles of the image.  Otherwise
// (sample, pixel).
MRT_DEV uint32_t pass_path( const PassInfo& pi, uint32_t s_local, uint32_t p_local ) {
    if ( pass_tiled( pi ) ) return p_local * pi.n_samples + s_local;
    return s_local * pi.n_pixels + p_local;
}
MRT_DEV void pass_split( const PassInfo& pi, uint32_t id, uint32_t& s_local, uint32_t& p_local ) {
    if ( pass_tiled( pi ) ) {
        p_local = id / pi.n_samples;
        s_local = id - p_local * pi.n_samples;
        return;
    }
    s_local = id / pi.n_pixels;
    p_local = id - s_local * pi.n_pixels;
}

struct Queues {
    uint32_t* cur;     // live paths of this bounce
    uint32_t* next;    // survivors
    uint32_t* shadow;  // paths with a shadow ray
    uint32_t* mis;     // paths with a MIS ray
    uint32_t* counts;  // [0] next, [1] shadow, [2] mis
};

struct RenderCounters {
    unsigned long long* c; // [0] closest rays, [1] any-hit rays
};

// Appends `id` to up to three queues for the threads whose `take[k]` holds.  Appends to one queue
// all go through one counter, and same-address atomics are served one at a time by the memory
// side (a frame makes tens of millions of appends), so the block first adds up its waves' counts
// in LDS and reserves its range of each queue with a single atomic.  Every thread of the block
// must call this.
#ifndef MRT_SHADE_BLOCK
#define MRT_SHADE_BLOCK 512
#ifndef MRT_SHADE_WAVES
#define MRT_SHADE_WAVES 4 // waves per SIMD the shade kernel is compiled for
#endif
#ifndef MRT_SHADE_RUN
#define MRT_SHADE_RUN 2u // consecutive blocks of paths a block of a later bounce draws at a time
#endif
#endif
MRT_DEV void queue_push3( uint32_t* const* queue, uint32_t* counters, const bool* take, uint32_t id ) {
    __shared__ uint32_t s_cnt[3][MRT_SHADE_BLOCK / 64];
    __shared__ uint32_t s_base[3];
    const uint32_t lane = __lane_id(), wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    unsigned long long mask[3];
    for ( int k = 0; k < 3; k++ ) {
        mask[k] = __ballot( take[k] );
        if ( lane == 0 ) s_cnt[k][wave] = (uint32_t)__popcll( mask[k] );
    }
    __syncthreads();
    if ( threadIdx.x < 3 ) {
        const uint32_t k = threadIdx.x;
        uint32_t sum     = 0;
        for ( uint32_t w = 0; w < n_waves; w++ ) {
            const uint32_t c = s_cnt[k][w];
            s_cnt[k][w]      = sum;
            sum += c;
        }
        s_base[k] = sum ? atomicAdd( counters + k, sum ) : 0u;
    }
    __syncthreads();
    for ( int k = 0; k < 3; k++ )
        if ( take[k] ) queue[k][s_base[k] + s_cnt[k][wave] + (uint32_t)__popcll( mask[k] & ( ( 1ull << lane ) - 1ull ) )] = id;
}

// ------------------------------------------------------------------------------------------------
// extend: closest hit of the continuation rays.  `queue` == nullptr means the identity queue.
struct ExtendIO {
    static constexpr bool is_segment_io = false;
    PathState ps;
    const uint32_t* queue;
    MRT_DEV void load( uint32_t i, uint32_t& id, V3& o, V3& d, float& t0, float& t1 ) const {
        id = queue ? queue[i] : i;
        get_ray( ps.ray0, ps.ray1, id, o, d, t1 );
        t0 = 0.f;
    }
    MRT_DEV void store( uint32_t id, bool, const DHit& h ) const { put_hit( ps.hit, id, h ); }
    MRT_DEV void reload( uint32_t id, V3& o, V3& d, float& t1 ) const { get_ray( ps.ray0, ps.ray1, id, o, d, t1 ); }
    PathRays rays() const { return PathRays { ps.ray0, ps.ray1 }; }
};

// The first bounce makes its camera rays itself: the ray of path i is computed when a lane takes it
// and traced.  There is no generate pass and no reading back of what it would have written (+1.2 % on
// the frame).  Throughput 1, radiance 0, no pending term and the start of the engine stream are what
// the first shade of a path assumes without reading them.
// The camera ray of path `id` of a pass: Sampler::GetCameraSample (sampler.cpp:43-49: 2D dimension 0,
// 1D dimension 0 = time, unused, 2D dimension 1) and SimpleCamera::get_ray.  A function of the path id
// alone, so nobody stores it: the first extend and the first shade each compute it (64 bytes of path
// state less written and read back per camera sample), and so does a relay launch.
MRT_DEV void camera_path_ray( const PassInfo& pi, const DCamera& cam, uint32_t id, V3& o, V3& d ) {
    uint32_t s_local, p_local;
    pass_split( pi, id, s_local, p_local );
    uint32_t x, y;
    pass_pixel( pi, p_local, x, y );
    const uint32_t pixel  = y * pi.width + x;
    const uint32_t sample = pi.sample_begin + s_local;
    const uint64_t key    = pixel_key( pi.seed, pixel );
    V2 jit  = strat2d( key, 0, sample, pi.spp_x, pi.spp_y );
    V2 lens = strat2d( key, 1, sample, pi.spp_x, pi.spp_y );
    camera_ray( cam, (float)x + jit.x, (float)y + jit.y, lens.x, lens.y, o, d );
}
#define MRT_CAMERA_TMAX 100000.f
struct CameraRays {
    PassInfo pi;
    DCamera cam;
    MRT_DEV void reload( uint32_t id, V3& o, V3& d, float& tmax ) const {
        camera_path_ray( pi, cam, id, o, d );
        tmax = MRT_CAMERA_TMAX;
    }
};
struct CameraIO {
    static constexpr bool is_segment_io = false;
    PathState ps;
    PassInfo pi;
    DCamera cam;
    MRT_DEV void load( uint32_t i, uint32_t& id, V3& o, V3& d, float& t0, float& t1 ) const {
        id = i;
        camera_path_ray( pi, cam, id, o, d );
        t0 = 0.f, t1 = MRT_CAMERA_TMAX;
    }
    MRT_DEV void store( uint32_t id, bool, const DHit& h ) const { put_hit( ps.hit, id, h ); }
    MRT_DEV void reload( uint32_t id, V3& o, V3& d, float& t1 ) const {
        camera_path_ray( pi, cam, id, o, d );
        t1 = MRT_CAMERA_TMAX;
    }
    CameraRays rays() const { return CameraRays { pi, cam }; }
};
// camera rays: the samples of a pixel fetch the same records, which the first-level cache serves; the
// kernel is bound by instruction issue and its registers are all in use, so no mailbox
template <>
struct Mailbox<CameraIO> {
    static constexpr int size = 0;
};
template <bool STATS>
#ifndef MRT_CAMERA_WAVES
#define MRT_CAMERA_WAVES MRT_TRACE_WAVES
#endif
__global__ void __launch_bounds__( MRT_TRACE_BLOCK, MRT_CAMERA_WAVES ) extend_camera_kernel( DTrace tr, CameraIO io, uint32_t n, uint32_t* __restrict__ deal, unsigned long long* __restrict__ stats, Relay relay ) {
    __shared__ uint32_t lds[MRT_TRACE_LDS_WORDS( false )];
    lds_u32* lp = (lds_u32*)lds;
    DStats st { 0, 0, 0, 0 };
    trace_stream<false, false, STATS>( tr, lp, io, n, deal, st, relay );
    if ( STATS ) flush_stats( st, stats );
}

// The same with the 64 rays of a wave walking the tree together (trace_packets): consecutive paths are
// samples of one pixel, whose camera rays differ by a lens offset and a sub-pixel jitter.
#ifndef MRT_PACKET_WAVES
#define MRT_PACKET_WAVES 8
#endif
__global__ void __launch_bounds__( MRT_TRACE_BLOCK, MRT_PACKET_WAVES ) extend_camera_packet_kernel( DTrace tr, CameraIO io, uint32_t n, uint32_t* __restrict__ deal ) {
    __shared__ uint32_t lds[MRT_XF_LDS * 12];
    trace_packets( tr, (lds_u32*)lds, io, n, deal );
}

// STATS (all traversal kernels of this file): also count the visited nodes, leaves, references
// and primitive tests into stats[0..3].
template <bool STATS>
__global__ void __launch_bounds__( MRT_TRACE_BLOCK, MRT_TRACE_WAVES ) extend_kernel( DTrace tr, ExtendIO io, const uint32_t* __restrict__ n_ptr, uint32_t n_max, uint32_t* __restrict__ deal, unsigned long long* __restrict__ stats, Relay relay ) {
    __shared__ uint32_t lds[MRT_TRACE_LDS_WORDS( false )];
    lds_u32* lp = (lds_u32*)lds;
    DStats st { 0, 0, 0, 0 };
    trace_stream<false, false, STATS>( tr, lp, io, n_ptr ? *n_ptr : n_max, deal, st, relay );
    if ( STATS ) flush_stats( st, stats );
}

// ------------------------------------------------------------------------------------------------
// The pending word of a path: byte 0 = a direct-lighting term is pending, byte 2 = its MIS ray
// escaped (written, as a single byte, by connect_mis_occlusion; the term shade stored in mb for that
// case is then part of DI).  shade writes the whole word, which clears byte 2.
// Byte 1 = the term's throughput factor is (1, 1, 1) and was not stored (a path's first bounce).
#define MRT_PENDING 0x000000ffu
#define MRT_PENDING_TP_ONE 0x0000ff00u
#define MRT_PENDING_MIS_ESCAPED 0x00ff0000u
// pick_pdf: the pdf of the light pick, Distribution1D::SampleDiscrete's func[offset] / (funcInt * count)
// (sampling.h:103-111), the same for every light of a UniformLightDistribution
MRT_DEV float light_pick_pdf( float func_int, uint32_t n_lights ) {
    return ( func_int > 0 ) ? 1.f / ( func_int * (int)n_lights ) : 0;
}
MRT_DEV void fold_pending( PathState& ps, uint32_t id, V3& L, float pick_pdf ) {
    const uint32_t w = ps.pending[id];
    if ( !( w & MRT_PENDING ) ) return;
    // ret += throughput * directIllumination(...) / lightPdf   (integrator.cpp:169-177)
    V3 tp = mk3( 1.f ), di = mk3( ps.dix[id], ps.diy[id], ps.diz[id] );
    if ( !( w & MRT_PENDING_TP_ONE ) ) tp = mk3( ps.ptx[id], ps.pty[id], ps.ptz[id] );
    if ( w & MRT_PENDING_MIS_ESCAPED ) di = di + mk3( ps.mbx[id], ps.mby[id], ps.mbz[id] ); // integrator.cpp:114-117
    L     = L + ( tp * di ) / pick_pdf;
    ps.pending[id] = 0;
}

// Regroups the paths of a shade block by kind.  In: this thread's path (`have`: it has one).  Out:
// the path this thread is to shade instead (return value: it has one).  The block's paths are
// partitioned, order kept within a kind, into miss / plain / textured / transparent, and thread t
// takes the t-th of them.  Every thread of the block must call this.
MRT_DEV bool bucket_by_kind( const DTrace& tr, const PathState& ps, uint32_t& id, bool have ) {
    __shared__ uint32_t s_ids[MRT_SHADE_BLOCK];
    __shared__ uint32_t s_cnt[4][MRT_SHADE_BLOCK / 64];
    __shared__ uint32_t s_total;
    uint32_t kind = 4u; // no path
    if ( have ) {
        const uint32_t bits = __float_as_uint( ps.hit[id].x );
        if ( bits == 0xffffffffu ) kind = 0u;
        else {
            const uint32_t flags = tr.prim_info[bits & 0x7fffffffu].x;
            kind                 = 1u + ( ( flags >> MRT_PRIM_SHADE_SHIFT ) & 3u );
            if ( kind > 3u ) kind = 3u;
        }
    }
    const uint32_t lane = __lane_id(), wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    unsigned long long mine = 0;
    for ( uint32_t k = 0; k < 4u; k++ ) {
        const unsigned long long m = __ballot( kind == k );
        if ( kind == k ) mine = m;
        if ( lane == 0 ) s_cnt[k][wave] = (uint32_t)__popcll( m );
    }
    __syncthreads();
    if ( threadIdx.x == 0 ) {
        uint32_t sum = 0;
        for ( uint32_t k = 0; k < 4u; k++ )
            for ( uint32_t w = 0; w < n_waves; w++ ) {
                const uint32_t c = s_cnt[k][w];
                s_cnt[k][w]      = sum;
                sum += c;
            }
        s_total = sum;
    }
    __syncthreads();
    if ( kind < 4u ) s_ids[s_cnt[kind][wave] + (uint32_t)__popcll( mine & ( ( 1ull << lane ) - 1ull ) )] = id;
    __syncthreads();
    const bool out = threadIdx.x < s_total;
    if ( out ) id = s_ids[threadIdx.x];
    __syncthreads(); // the arrays are reused by the block's next run
    return out;
}

// One loop iteration of trace_ray for every live path of the bounce.
// `chunks` == nullptr: one block per MRT_SHADE_BLOCK paths (the first bounce, whose path count the
// host knows).  Otherwise the grid is sized for the chip and its blocks draw runs of
// MRT_SHADE_RUN x MRT_SHADE_BLOCK paths from the counter `chunks` (zero at launch) until the live
// paths, whose number is read from the device, are used up: a late bounce with few paths left costs
// a short launch, and because the runs are handed out in order, the queues this kernel appends to
// keep the order of the queue it reads (a fixed stride would let blocks with cheap paths run ahead,
// and the next traversal kernels would find neighbouring rays far apart in their lists).
// MIS_ESCAPE: the scene has no emissive object, so a MIS ray contributes only when it escapes, and
// what it then contributes -- bsdf * Le(wi) * weight / pdf -- is known here: it is stored in mb and
// connect_mis_occlusion adds it for the rays that escape (no mis_resolve pass; a MIS ray whose
// contribution would be black is not traced at all).
// FIRST: the first bounce (depth 0, identity queue, one block per MRT_SHADE_BLOCK paths).
// BUCKET (later bounces of a scene whose materials are of more than one shade kind): the paths a
// block draws are regrouped by what shade will do with them -- miss, plain, textured, transparent
// (bucket_by_kind) -- so that the lanes of a wave run the same branch of the material code.  In the
// first bounce a wave holds 64 samples of one pixel, which hit one material anyway.
template <bool MIS_ESCAPE, bool FIRST, bool BUCKET = false>
__global__ void __launch_bounds__( MRT_SHADE_BLOCK, MRT_SHADE_WAVES ) shade_kernel( DTrace tr, DShade sh, PathState ps, PassInfo pi, Queues q, const uint32_t* __restrict__ queue_, const uint32_t* __restrict__ n_ptr, uint32_t n_max, uint32_t depth_, uint32_t* __restrict__ chunks_ ) {
    const uint32_t depth           = FIRST ? 0u : depth_;
    const uint32_t* const queue    = FIRST ? nullptr : queue_;
    uint32_t* const chunks         = FIRST ? nullptr : chunks_;
    const uint32_t n = ( !FIRST && n_ptr ) ? *n_ptr : n_max;
    __shared__ uint32_t s_run;
    for ( uint32_t step = 0;; step++ ) {
    uint32_t base;
    if ( chunks ) {
        if ( step % MRT_SHADE_RUN == 0u ) {
            if ( threadIdx.x == 0 ) s_run = atomicAdd( chunks, 1u );
            __syncthreads(); // the next write of s_run comes after the barriers of queue_push3
        }
        base = ( s_run * MRT_SHADE_RUN + step % MRT_SHADE_RUN ) * MRT_SHADE_BLOCK;
    }
    else base = blockIdx.x * MRT_SHADE_BLOCK;
    if ( base >= n || ( !chunks && step != 0u ) ) break;
    const uint32_t i = base + threadIdx.x;
    bool alive = false, has_shadow = false, has_mis = false;
    uint32_t id = 0;
    bool have   = i < n;
    if ( have ) id = queue ? queue[i] : i;
    if ( BUCKET ) have = bucket_by_kind( tr, ps, id, have );
    if ( have ) {
        // a path's first shade starts from radiance 0, throughput 1 and nothing pending
        V3 L = mk3( 0.f ), tp = mk3( 1.f );
#ifdef MRT_EXTRA_VALU_SHADE
        { // measurement aid: that many more vector-ALU instructions per shaded path
            float extra_valu = 1.f;
#pragma unroll
            for ( int i = 0; i < MRT_EXTRA_VALU_SHADE; i++ )
                asm volatile( "v_add_f32 %0, %0, %0" : "+v"( extra_valu ) );
        }
#endif
        if ( depth != 0 ) {
            L = mk3( ps.Lx[id], ps.Ly[id], ps.Lz[id] );
            fold_pending( ps, id, L, light_pick_pdf( sh.light_func_int, sh.n_lights ) );
            tp = mk3( ps.tpx[id], ps.tpy[id], ps.tpz[id] );
        }
        bool pending = false;
        V3 ro, rd;
        if ( FIRST ) camera_path_ray( pi, sh.cam, id, ro, rd ); // nobody stored it
        else {
            float ray_far;
            get_ray( ps.ray0, ps.ray1, id, ro, rd, ray_far );
        }
        DHit h;
        get_hit( ps.hit, id, h );
        if ( h.prim < 0 ) {
            // integrator.cpp:195-198
            if ( sh.env_light >= 0 ) L = L + tp * ibl_Le( sh, rd );
        }
        else {
            Surf sf = surface_from_hit( tr, sh, ro, rd, h );
            const DMaterial mat = sh.materials[sf.material];
            uint32_t s_local, p_local;
            pass_split( pi, id, s_local, p_local );
            uint32_t px, py;
            pass_pixel( pi, p_local, px, py );
            const uint32_t pixel = py * pi.width + px;
            const uint32_t sample = pi.sample_begin + s_local;
            const uint64_t key    = pixel_key( pi.seed, pixel );
            // the engine stream of the sample: its increment is a function of the sequence index,
            // only the state travels with the path
            Pcg32 rng;
            pcg_set_sequence( rng, sample_sequence( key, sample ) );
            if ( depth != 0 ) rng.state = ps.rng_state[id];

            // integrator.cpp:147-150
            if ( !is_black( ld3( mat.emission ) ) && depth == 0 ) {
                V3 col = dot3( sf.normal, -rd ) > 0.f ? ld3( sh.light0_color ) : mk3( 0.f );
                L      = L + tp * col;
            }
            // integrator.cpp:152-178
            if ( sh.n_lights > 0 ) {
                // sampler dimensions consumed so far: 1D 1 + depth, 2D 2 + 3*depth
                const uint32_t d1 = 1 + depth, d2 = 2 + 3 * depth;
                float u1 = d1 < 8 ? strat1d( key, d1, sample, pi.spp_x * pi.spp_y ) : pcg_float( rng );
                float lightPickPdf;
                int li = sample_discrete( sh.light_cdf, (int)sh.n_lights + 1, sh.light_func_int, u1, lightPickPdf );
                if ( lightPickPdf > 0 ) {
                    V2 uLight;
                    if ( d2 < 8 ) uLight = strat2d( key, d2, sample, pi.spp_x, pi.spp_y );
                    else {
                        uLight.x = pcg_float( rng );
                        uLight.y = pcg_float( rng );
                    }
                    // ---- directIllumination (integrator.cpp:69-120)
                    const DLight& light = sh.lights[li];
                    V3 di               = mk3( 0.f );
                    V3 wi               = mk3( 0.f );
                    float lightPdf = 0, scatteringPdf = 0;
                    V3 Li;
                    V3 s_o, s_d;
                    float s_tmax;
                    if ( light.kind == 0 ) { // ShapeLight::sample_Li (Light.cpp:251-271)
                        ShapePoint pShape = light_shape_sample( tr, light, sf, uLight, lightPdf );
                        wi                = normalize3( pShape.position - sf.position );
                        s_o               = sf.position + ( MRT_ACNE_EPS * sf.normal );
                        s_d               = wi;
                        s_tmax            = length3( sf.position - pShape.position ) * 0.99f;
                        Li                = dot3( pShape.normal, -wi ) > 0.f ? ld3( light.color ) : mk3( 0.f );
                    }
                    else { // IBL::sample_Li (Light.cpp:277-315)
                        V3 normal = sf.outside ? sf.normal : -sf.normal;
                        wi        = normalize3( random_dir( normal, rng ) );
                        lightPdf  = fmax_glm( dot3( normal, wi ), 0.f ) / MRT_PI;
                        s_o       = sf.position + wi * MRT_ACNE_EPS;
                        s_d       = wi;
                        s_tmax    = 10000.f;
                        Li        = ibl_sample_texel( sh, wi );
                    }
                    // the reference always traces the shadow ray inside sample_Li; its result is
                    // only read when the contribution below is non-black, so it is traced only then
                    if ( lightPdf > 0 && !is_black( Li ) ) {
                        V3 normal = sf.outside ? sf.normal : -sf.normal;
                        V3 bsdf   = material_eval( sh, mat, rd, sf, wi, scatteringPdf ) * fmax_glm( dot3( normal, wi ), 0.f );
                        if ( !is_black( bsdf ) ) {
                            float weight = power_heuristic( lightPdf, scatteringPdf );
                            di           = di + bsdf * Li * weight / lightPdf;
                            has_shadow   = true;
                            put_ray( ps.sray0, ps.sray1, id, s_o, s_d, s_tmax );
                        }
                    }
                    // integrator.cpp:97-99: `sample(..., &wi, ...) * max(dot(normal, wi), 0)` -- the
                    // reference as built by g++ evaluates the right operand first, i.e. the cosine
                    // uses the light-sample direction still in wi; made explicit here.
                    const float cos_prev_wi = fmax_glm( dot3( sf.normal, wi ), 0.f );
                    V3 bsdf                 = material_sample( sh, mat, rd, sf, rng, wi, scatteringPdf ) * cos_prev_wi;
                    if ( !is_black( bsdf ) && scatteringPdf > 0 ) {
                        float lp = light.kind == 0 ? light_shape_pdf( tr, light, sf, wi ) : ibl_pdf_Li( sh, wi );
                        if ( lp != 0 ) {
                            float weight = power_heuristic( scatteringPdf, lp );
                            V3 m_o       = sf.position + ( wi * MRT_ACNE_EPS );
                            V3 m_d       = normalize3( wi );
                            if ( MIS_ESCAPE ) {
                                // integrator.cpp:108-117 for a ray that hits nothing: Li = light->Le(ray),
                                // black for a shape light
                                V3 Le = light.kind == 1 ? ibl_Le( sh, m_d ) : mk3( 0.f );
                                if ( !is_black( Le ) ) {
                                    V3 c    = bsdf * Le * weight / scatteringPdf;
                                    has_mis = true;
                                    put_ray( ps.mray0, ps.mray1, id, m_o, m_d, 10000.f );
                                    ps.mbx[id] = c.x, ps.mby[id] = c.y, ps.mbz[id] = c.z;
                                }
                            }
                            else {
                                has_mis = true;
                                put_ray( ps.mray0, ps.mray1, id, m_o, m_d, 10000.f );
                                ps.mbx[id] = bsdf.x, ps.mby[id] = bsdf.y, ps.mbz[id] = bsdf.z;
                                ps.mw[id]     = weight;
                                ps.mpdf[id]   = scatteringPdf;
                                ps.mlight[id] = (uint32_t)li;
                            }
                        }
                    }
                    if ( !FIRST ) ps.ptx[id] = tp.x, ps.pty[id] = tp.y, ps.ptz[id] = tp.z; // (1, 1, 1) in a first bounce: MRT_PENDING_TP_ONE
                    ps.dix[id] = di.x, ps.diy[id] = di.y, ps.diz[id] = di.z;
                    pending    = true; // (lightPickPdf is light_pick_pdf(): the fold computes it again)
                }
            }
            // integrator.cpp:180-193
            float pdf = 0.f;
            V3 wi     = mk3( 0.f );
            V3 bsdf   = material_sample( sh, mat, rd, sf, rng, wi, pdf );
            wi        = normalize3( wi );
            if ( !( is_black( bsdf ) || pdf == 0.f ) ) {
                if ( mat.type != 2 ) tp = tp * ( bsdf * fabsf( dot3( wi, sf.normal ) ) / pdf );
                alive = depth < pi.max_depth; // integrator.cpp:201
                if ( alive ) {
                    V3 no = sf.position + MRT_ACNE_EPS * wi;
                    put_ray( ps.ray0, ps.ray1, id, no, wi, 10000.f );
                    ps.tpx[id] = tp.x, ps.tpy[id] = tp.y, ps.tpz[id] = tp.z;
                    ps.rng_state[id] = rng.state;
                }
            }
        }
        // fold_pending cleared the flag of a later bounce; the first one finds it unwritten
        if ( pending || depth == 0 ) ps.pending[id] = pending ? ( FIRST ? 0x0101u : 1u ) : 0u; // the whole word: no MIS ray has escaped yet
        ps.Lx[id] = L.x, ps.Ly[id] = L.y, ps.Lz[id] = L.z;
    }
    uint32_t* const queues[3] = { q.next, q.shadow, q.mis };
    const bool take[3]        = { alive, has_shadow, has_mis };
    queue_push3( queues, q.counts, take, id );
    }
}

// ------------------------------------------------------------------------------------------------
// connect_shadow: any-hit traversal of the shadow rays.
struct ShadowIO {
    static constexpr bool is_segment_io = false;
    PathState ps;
    const uint32_t* queue;
    MRT_DEV void load( uint32_t i, uint32_t& id, V3& o, V3& d, float& t0, float& t1 ) const {
        id = queue[i];
        get_ray( ps.sray0, ps.sray1, id, o, d, t1 );
        t0 = 0.f;
    }
    // integrator.cpp:88: `if (vis) Li = 0` -- the light term was the only term in DI so far
    MRT_DEV void store( uint32_t id, bool occluded, const DHit& ) const {
        if ( occluded ) ps.dix[id] = ps.diy[id] = ps.diz[id] = 0.f;
    }
    MRT_DEV void reload( uint32_t id, V3& o, V3& d, float& t1 ) const { get_ray( ps.sray0, ps.sray1, id, o, d, t1 ); }
    PathRays rays() const { return PathRays { ps.sray0, ps.sray1 }; }
};

template <bool STATS>
__global__ void __launch_bounds__( MRT_TRACE_BLOCK, MRT_OCCLUSION_WAVES ) connect_shadow_kernel( DTrace tr, ShadowIO io, const uint32_t* __restrict__ count, uint32_t* __restrict__ deal, unsigned long long* __restrict__ stats, Relay relay ) {
    __shared__ uint32_t lds[MRT_TRACE_LDS_WORDS( true )];
    lds_u32* lp = (lds_u32*)lds;
    DStats st { 0, 0, 0, 0 };
    trace_stream<true, true, STATS>( tr, lp, io, *count, deal, st, relay );
    if ( STATS ) flush_stats( st, stats );
}

// connect_mis: closest-hit traversal of the MIS rays.  The hit goes to the h_* arrays of the path
// (shade has consumed the continuation hit by now and the next extend has not run yet);
// mis_resolve_kernel turns it into the bsdf-sampling term.
struct MisIO {
    static constexpr bool is_segment_io = false;
    PathState ps;
    const uint32_t* queue;
    MRT_DEV void load( uint32_t i, uint32_t& id, V3& o, V3& d, float& t0, float& t1 ) const {
        id = queue[i];
        get_ray( ps.mray0, ps.mray1, id, o, d, t1 );
        t0 = 0.f;
    }
    MRT_DEV void store( uint32_t id, bool, const DHit& h ) const { put_hit( ps.hit, id, h ); }
    MRT_DEV void reload( uint32_t id, V3& o, V3& d, float& t1 ) const { get_ray( ps.mray0, ps.mray1, id, o, d, t1 ); }
    PathRays rays() const { return PathRays { ps.mray0, ps.mray1 }; }
};

template <bool STATS>
__global__ void __launch_bounds__( MRT_TRACE_BLOCK, MRT_TRACE_WAVES ) connect_mis_kernel( DTrace tr, MisIO io, const uint32_t* __restrict__ count, uint32_t* __restrict__ deal, unsigned long long* __restrict__ stats, Relay relay ) {
    __shared__ uint32_t lds[MRT_TRACE_LDS_WORDS( false )];
    lds_u32* lp = (lds_u32*)lds;
    DStats st { 0, 0, 0, 0 };
    trace_stream<false, false, STATS>( tr, lp, io, *count, deal, st, relay );
    if ( STATS ) flush_stats( st, stats );
}

// The same for a scene without a single emissive object.  Then the MIS ray only decides between
// "hit something" (no radiance) and "escaped" (environment radiance), and whether the closest-hit
// search of the reference finds a hit is exactly whether its shadow-ray search does (both accept a
// primitive on the same closed leaf interval; they differ in what they do after the first accepted
// hit and in skipping emissive primitives, of which there are none).  So the ray is traced with
// shadow-ray semantics: it stops at the first hit and a plane hit spares the tree altogether.
struct MisOcclusionIO {
    static constexpr bool is_segment_io = false;
    PathState ps;
    const uint32_t* queue;
    MRT_DEV void load( uint32_t i, uint32_t& id, V3& o, V3& d, float& t0, float& t1 ) const {
        id = queue[i];
        get_ray( ps.mray0, ps.mray1, id, o, d, t1 );
        t0 = 0.f;
    }
    // an escaped ray makes the contribution shade computed for that case: one byte tells the fold
    MRT_DEV void store( uint32_t id, bool found, const DHit& ) const {
        if ( !found ) ( (uint8_t*)ps.pending )[(size_t)4 * id + 2u] = 1;
    }
    MRT_DEV void reload( uint32_t id, V3& o, V3& d, float& t1 ) const { get_ray( ps.mray0, ps.mray1, id, o, d, t1 ); }
    PathRays rays() const { return PathRays { ps.mray0, ps.mray1 }; }
};
#ifdef MRT_REFILL_MIN_CAMERA // measurement aids: refill thresholds per ray source
template <>
struct RefillMin<CameraIO> {
    static constexpr uint32_t value = MRT_REFILL_MIN_CAMERA;
};
#endif
#ifdef MRT_REFILL_MIN_OCCLUSION
template <>
struct RefillMin<ShadowIO> {
    static constexpr uint32_t value = MRT_REFILL_MIN_OCCLUSION;
};
template <>
struct RefillMin<MisOcclusionIO> {
    static constexpr uint32_t value = MRT_REFILL_MIN_OCCLUSION;
};
#endif

template <bool STATS>
__global__ void __launch_bounds__( MRT_TRACE_BLOCK, MRT_OCCLUSION_WAVES ) connect_mis_occlusion_kernel( DTrace tr, MisOcclusionIO io, const uint32_t* __restrict__ count, uint32_t* __restrict__ deal, unsigned long long* __restrict__ stats, Relay relay ) {
    __shared__ uint32_t lds[MRT_TRACE_LDS_WORDS( true )];
    lds_u32* lp = (lds_u32*)lds;
    DStats st { 0, 0, 0, 0 };
    trace_stream<true, true, STATS>( tr, lp, io, *count, deal, st, relay );
    if ( STATS ) flush_stats( st, stats );
}

__global__ void __launch_bounds__( 256 ) mis_resolve_kernel( DTrace tr, DShade sh, PathState ps, const uint32_t* __restrict__ queue, const uint32_t* __restrict__ count ) {
    const uint32_t n = *count;
    for ( uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x ) {
        const uint32_t id = queue[i];
        DHit h;
        get_hit( ps.hit, id, h );
        V3 o, d;
        float far_unused;
        get_ray( ps.mray0, ps.mray1, id, o, d, far_unused );
        const uint32_t li = ps.mlight[id];
        V3 Li             = mk3( 0.f );
        if ( h.prim >= 0 ) { // integrator.cpp:108-113
            const uint32_t flags = tr.prim_info[h.prim].x;
            if ( flags & MRT_PRIM_EMISSIVE_SUM ) {
                Surf sf = surface_from_hit( tr, sh, o, d, h );
                // `wi` of the reference is the un-normalised sample direction; the ray direction is
                // normalize(wi) and wi is already unit length up to rounding, the sign test is the same
                Li = dot3( sf.normal, -d ) > 0.f ? ld3( sh.lights[li].color ) : mk3( 0.f );
            }
        }
        else if ( sh.lights[li].kind == 1 ) { Li = ibl_Le( sh, d ); } // Light::Le is black for shape lights
        if ( !is_black( Li ) ) {
            V3 bsdf = mk3( ps.mbx[id], ps.mby[id], ps.mbz[id] );
            V3 c    = bsdf * Li * ps.mw[id] / ps.mpdf[id];
            ps.dix[id] += c.x;
            ps.diy[id] += c.y;
            ps.diz[id] += c.z;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Radiances of a pixel's samples are added in sample order (CPURenderer.cpp:60: pixel_color +=
// trace_ray(...)), then added to the image sum.
// Tiled pass (a pixel's samples are consecutive paths): a block takes 64 pixels; its threads load
// MRT_ACC_TILE consecutive samples of each of them (coalesced) into an LDS tile, then one lane per pixel
// and colour channel adds that pixel's values in order.
#ifndef MRT_ACC_TILE
#define MRT_ACC_TILE 32u // samples of a pixel per tile row (32: half the LDS of 64 and twice the resident blocks, +0.2 % on the frame; 16: -0.4 %)
#endif
__global__ void __launch_bounds__( 256 ) accumulate_pixel_kernel( PathState ps, PassInfo pi, float* __restrict__ rgb_sum, float func_int, uint32_t n_lights ) {
    const float pick_pdf = light_pick_pdf( func_int, n_lights );
    __shared__ float tile[3][64][MRT_ACC_TILE + 1u];
    const uint32_t p0 = blockIdx.x * 64u, wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    float sum = 0.f; // of pixel p0 + lane, channel `wave` (waves 0..2)
    for ( uint32_t base = 0; base < pi.n_samples; base += MRT_ACC_TILE ) {
        // thread t loads sample t % TILE of pixel row t / TILE (+ 256 / TILE rows per step)
        for ( uint32_t r = threadIdx.x / MRT_ACC_TILE; r < 64u; r += 256u / MRT_ACC_TILE ) {
            const uint32_t c = threadIdx.x % MRT_ACC_TILE;
            const uint32_t p = p0 + r, s = base + c;
            V3 L             = mk3( 0.f );
            if ( p < pi.n_pixels && s < pi.n_samples ) {
                const uint32_t id = pass_path( pi, s, p );
                L                 = mk3( ps.Lx[id], ps.Ly[id], ps.Lz[id] );
                fold_pending( ps, id, L, pick_pdf );
            }
            tile[0][r][c] = L.x, tile[1][r][c] = L.y, tile[2][r][c] = L.z;
        }
        __syncthreads();
        const uint32_t cnt = pi.n_samples - base < MRT_ACC_TILE ? pi.n_samples - base : MRT_ACC_TILE;
        if ( wave < 3u )
            for ( uint32_t k = 0; k < cnt; k++ )
                sum = sum + tile[wave][lane][k];
        __syncthreads();
    }
    if ( wave >= 3u || p0 + lane >= pi.n_pixels ) return;
    uint32_t px, py;
    pass_pixel( pi, p0 + lane, px, py );
    rgb_sum[3 * (size_t)( py * pi.width + px ) + wave] += sum;
}

// Any other pass (the samples of a pixel are n_pixels apart): one thread per pixel.
__global__ void __launch_bounds__( 256 ) accumulate_kernel( PathState ps, PassInfo pi, float* __restrict__ rgb_sum, float func_int, uint32_t n_lights ) {
    const float pick_pdf = light_pick_pdf( func_int, n_lights );
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if ( p >= pi.n_pixels ) return;
    V3 sum = mk3( 0.f );
    for ( uint32_t s = 0; s < pi.n_samples; s++ ) {
        const uint32_t id = pass_path( pi, s, p );
        V3 L              = mk3( ps.Lx[id], ps.Ly[id], ps.Lz[id] );
        fold_pending( ps, id, L, pick_pdf );
        sum = sum + L;
    }
    uint32_t px, py;
    pass_pixel( pi, p, px, py );
    const uint32_t pixel = py * pi.width + px;
    float* dst           = rgb_sum + 3 * (size_t)pixel;
    dst[0] += sum.x;
    dst[1] += sum.y;
    dst[2] += sum.z;
}

// out = powf(sum / spp, 1.0f / 2.2) (CPURenderer.cpp:64-69)
__global__ void __launch_bounds__( 256 ) resolve_kernel( const float* __restrict__ sum, float* __restrict__ out, uint64_t n, float spp ) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if ( i >= n ) return;
    out[i] = powf( sum[i] / spp, (float)( 1.0f / 2.2 ) );
}

struct RenderWorkspace {
    float* words      = nullptr; // MRT_PATH_WORDS arrays of capacity floats (u64 arrays take two)
    uint32_t* queues  = nullptr; // 4 queues of capacity ids
    uint32_t* counts  = nullptr; // 4 counters
    unsigned long long* ray_counts = nullptr;
    uint32_t capacity = 0;
};
