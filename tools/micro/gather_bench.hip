// Random-gather ceilings of one GPU, measured the way the traversal kernels read memory: every lane
// of every wave reads its own 32-byte record (two 16-byte loads of one 32-byte-aligned record) from
// a table of random words, 64 different cache lines per wave instruction.
//   independent: the next index comes from a hash of the lane's counter; loads of successive steps
//                overlap freely (how many records per second the memory path can deliver at all)
//   chained:     the next index comes out of the record just read, as a kd-tree descent's does (how
//                many dependent steps per second the resident waves can make: lanes in flight / latency)
// for tables that fit the L2s, the memory-side cache, or neither, and 1..8 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o raytracer_amd/gather_bench tools/micro/gather_bench.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK( e )                                                                      \
    do {                                                                                \
        hipError_t err_ = ( e );                                                        \
        if ( err_ != hipSuccess ) {                                                     \
            std::fprintf( stderr, "%s: %s\n", #e, hipGetErrorString( err_ ) );          \
            std::exit( 1 );                                                             \
        }                                                                               \
    } while ( 0 )

__device__ inline uint32_t mix( uint32_t x ) {
    x ^= x >> 16, x *= 0x7feb352du, x ^= x >> 15, x *= 0x846ca68bu, x ^= x >> 16;
    return x;
}

template <bool CHAINED>
__global__ void __launch_bounds__( 256 ) gather_kernel( const uint4* __restrict__ table, uint32_t mask, uint32_t steps, uint32_t* __restrict__ out ) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t idx       = mix( tid * 2654435761u + 12345u ) & mask;
    uint32_t acc       = 0;
    for ( uint32_t s = 0; s < steps; s++ ) { // every lane makes exactly `steps` trips: the grid drains
        const uint4 a = table[2u * idx];
        const uint4 b = table[2u * idx + 1u];
        acc += a.y ^ b.z;
        // chained: the record decides where the lane goes next, salted with the lane's own counter --
        // a function of (record, step) alone would send every lane that ever reads a record another
        // lane reads at that step along the same path from then on, and the walkers would coalesce
        if ( CHAINED ) idx = ( a.x ^ b.w ^ mix( tid + s * 0x9e3779b9u ) ) & mask;
        else idx = mix( tid + s * 0x9e3779b9u ) & mask;
    }
    out[2u * tid]      = acc;
    out[2u * tid + 1u] = idx; // where the lane ended: the host counts the distinct ones
}

// Independent gathers of records of 64 or 128 bytes: GROUP = bytes / 16 neighbouring lanes read the
// 16-byte pieces of one random record, so one wave instruction touches 64 / GROUP random records (one
// 128-byte record = one whole cache line).  Against the 32-byte gathers above this tells how many
// bytes a miss really moves: if a quarter-line gather and a whole-line gather run at the same number
// of LINES per second, the memory moves whole lines either way.
template <uint32_t GROUP>
__global__ void __launch_bounds__( 256 ) gather_wide_kernel( const uint4* __restrict__ table, uint32_t mask /* of records */, uint32_t steps, uint32_t* __restrict__ out ) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, rec = tid / GROUP, piece = tid % GROUP;
    uint32_t acc       = 0;
    for ( uint32_t s = 0; s < steps; s++ ) {
        const uint32_t idx = mix( rec + s * 0x9e3779b9u ) & mask;
        const uint4 a      = table[(size_t)idx * GROUP + piece];
        acc += a.y ^ a.z;
    }
    out[2u * tid]      = acc;
    out[2u * tid + 1u] = 0;
}

__global__ void fill_kernel( uint32_t* p, size_t n ) {
    for ( size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x )
        p[i] = mix( (uint32_t)i * 747796405u + 2891336453u );
}

static double run( int chained, size_t mb, int waves, uint32_t record_bytes, uint32_t steps, const uint4* table, uint32_t* out, int cus, hipEvent_t e0, hipEvent_t e1, size_t* distinct ) {
    const uint32_t blocks = (uint32_t)( cus * waves ); // 256 threads = 4 waves = 1 per SIMD
    const uint32_t mask   = (uint32_t)( ( mb << 20 ) / record_bytes - 1 );
    float best            = 1e30f;
    for ( int rep = 0; rep < 3; rep++ ) {
        CHECK( hipEventRecord( e0, 0 ) );
        if ( record_bytes == 128 ) hipLaunchKernelGGL( gather_wide_kernel<8>, dim3( blocks ), dim3( 256 ), 0, 0, table, mask, steps, out );
        else if ( record_bytes == 64 ) hipLaunchKernelGGL( gather_wide_kernel<4>, dim3( blocks ), dim3( 256 ), 0, 0, table, mask, steps, out );
        else if ( chained ) hipLaunchKernelGGL( gather_kernel<true>, dim3( blocks ), dim3( 256 ), 0, 0, table, mask, steps, out );
        else hipLaunchKernelGGL( gather_kernel<false>, dim3( blocks ), dim3( 256 ), 0, 0, table, mask, steps, out );
        CHECK( hipEventRecord( e1, 0 ) );
        CHECK( hipEventSynchronize( e1 ) );
        float ms = 0;
        CHECK( hipEventElapsedTime( &ms, e0, e1 ) );
        if ( ms < best ) best = ms;
    }
    if ( distinct ) {
        std::vector<uint32_t> h( (size_t)blocks * 256 * 2 );
        CHECK( hipMemcpy( h.data(), out, h.size() * sizeof( uint32_t ), hipMemcpyDeviceToHost ) );
        std::vector<uint32_t> last( (size_t)blocks * 256 );
        for ( size_t i = 0; i < last.size(); i++ )
            last[i] = h[2 * i + 1];
        std::sort( last.begin(), last.end() );
        *distinct = (size_t)( std::unique( last.begin(), last.end() ) - last.begin() );
    }
    return (double)best;
}

// no arguments: the whole table.  One configuration (for counter passes, which want one kernel):
//   gather_bench <independent|chained> <table MB> <waves per SIMD> <record bytes 32|64|128> <steps>
int main( int argc, char** argv ) {
    int cus = 0, clock_khz = 0;
    CHECK( hipDeviceGetAttribute( &cus, hipDeviceAttributeMultiprocessorCount, 0 ) );
    CHECK( hipDeviceGetAttribute( &clock_khz, hipDeviceAttributeClockRate, 0 ) );
    std::printf( "%d CUs, %.2f GHz\n", cus, clock_khz * 1e-6 );
    const size_t max_bytes = (size_t)4096 << 20;
    uint4* table           = nullptr;
    uint32_t* out          = nullptr;
    CHECK( hipMalloc( (void**)&table, max_bytes ) );
    CHECK( hipMalloc( (void**)&out, (size_t)cus * 8 * 256 * 2 * sizeof( uint32_t ) ) );
    hipLaunchKernelGGL( fill_kernel, dim3( cus * 8 ), dim3( 256 ), 0, 0, (uint32_t*)table, max_bytes / 4 );
    CHECK( hipDeviceSynchronize() );
    hipEvent_t e0, e1;
    CHECK( hipEventCreate( &e0 ) );
    CHECK( hipEventCreate( &e1 ) );
    std::printf( "%-12s %6s %8s %10s %12s %12s %12s %14s %16s\n", "mode", "record", "table", "waves/SIMD", "G records/s", "ns per step", "GB/s useful", "G lines/s", "distinct at end" );
    auto line = [&]( int chained, size_t mb, int waves, uint32_t rb, uint32_t steps ) {
        size_t distinct    = 0;
        const double ms    = run( chained, mb, waves, rb, steps, table, out, cus, e0, e1, rb == 32 ? &distinct : nullptr );
        const double lanes = (double)cus * waves * 256.0;
        const double records = lanes * steps / ( rb / ( rb == 32 ? 32.0 : 16.0 ) ); // 32-byte mode: one record per lane and step
        char dist[64] = "-";
        if ( rb == 32 ) std::snprintf( dist, sizeof( dist ), "%zu of %.0f", distinct, lanes );
        // a 32- or 64-byte record lies in one 128-byte line; a 128-byte record is one
        std::printf( "%-12s %4u B %5zu MB %10d %12.1f %12.0f %12.0f %14.1f %16s\n", chained ? "chained" : "independent", rb, mb, waves, records / ms * 1e-6,
                     ms * 1e6 / steps, records * rb / ms * 1e-6, records / ms * 1e-6, dist );
        std::fflush( stdout );
    };
    if ( argc >= 6 ) {
        line( std::string( argv[1] ) == "chained", (size_t)std::atoi( argv[2] ), std::atoi( argv[3] ), (uint32_t)std::atoi( argv[4] ), (uint32_t)std::atoi( argv[5] ) );
    }
    else {
        for ( int chained = 0; chained < 2; chained++ )
            for ( size_t mb : { (size_t)1, (size_t)2, (size_t)8, (size_t)128, (size_t)512, (size_t)4096 } ) // powers of two: the index is masked
                for ( int waves : { 1, 2, 4, 6, 8 } )
                    line( chained, mb, waves, 32u, chained ? 2000u : 4000u );
        for ( uint32_t rb : { 64u, 128u } )
            for ( size_t mb : { (size_t)2, (size_t)8, (size_t)512, (size_t)4096 } )
                for ( int waves : { 2, 6 } )
                    line( 0, mb, waves, rb, 4000u );
    }
    CHECK( hipFree( table ) );
    CHECK( hipFree( out ) );
    return 0;
}
