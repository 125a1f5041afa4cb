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
#include <cstdlib>
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
        if ( CHAINED ) idx = ( a.x ^ b.w ^ s ) & mask;
        else idx = mix( tid + s * 0x9e3779b9u ) & mask;
    }
    out[tid] = acc ^ idx;
}

__global__ void fill_kernel( uint32_t* p, size_t n ) {
    for ( size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x )
        p[i] = mix( (uint32_t)i * 747796405u + 2891336453u );
}

int main() {
    int cus = 0, clock_khz = 0;
    CHECK( hipDeviceGetAttribute( &cus, hipDeviceAttributeMultiprocessorCount, 0 ) );
    CHECK( hipDeviceGetAttribute( &clock_khz, hipDeviceAttributeClockRate, 0 ) );
    std::printf( "%d CUs, %.2f GHz\n", cus, clock_khz * 1e-6 );
    const size_t max_bytes = (size_t)4096 << 20;
    uint4* table           = nullptr;
    uint32_t* out          = nullptr;
    CHECK( hipMalloc( (void**)&table, max_bytes ) );
    CHECK( hipMalloc( (void**)&out, (size_t)cus * 8 * 256 * 4 * sizeof( uint32_t ) ) );
    hipLaunchKernelGGL( fill_kernel, dim3( cus * 8 ), dim3( 256 ), 0, 0, (uint32_t*)table, max_bytes / 4 );
    CHECK( hipDeviceSynchronize() );
    hipEvent_t e0, e1;
    CHECK( hipEventCreate( &e0 ) );
    CHECK( hipEventCreate( &e1 ) );
    std::printf( "%-12s %8s %10s %12s %14s %12s\n", "mode", "table", "waves/SIMD", "G records/s", "ns per step", "GB/s" );
    for ( int chained = 0; chained < 2; chained++ )
        for ( size_t mb : { (size_t)8, (size_t)128, (size_t)512, (size_t)4096 } ) { // powers of two: the index is masked
            const uint32_t mask = (uint32_t)( ( mb << 20 ) / 32 - 1 );
            for ( int waves : { 1, 2, 4, 6, 8 } ) {
                const uint32_t blocks = (uint32_t)( cus * waves ); // 256 threads = 4 waves = 1 per SIMD
                const uint32_t steps  = chained ? 2000u : 4000u;
                float best            = 1e30f;
                for ( int rep = 0; rep < 3; rep++ ) {
                    CHECK( hipEventRecord( e0, 0 ) );
                    if ( chained ) hipLaunchKernelGGL( gather_kernel<true>, dim3( blocks ), dim3( 256 ), 0, 0, table, mask, steps, out );
                    else hipLaunchKernelGGL( gather_kernel<false>, dim3( blocks ), dim3( 256 ), 0, 0, table, mask, steps, out );
                    CHECK( hipEventRecord( e1, 0 ) );
                    CHECK( hipEventSynchronize( e1 ) );
                    float ms = 0;
                    CHECK( hipEventElapsedTime( &ms, e0, e1 ) );
                    if ( ms < best ) best = ms;
                }
                const double records = (double)blocks * 256.0 * steps;
                std::printf( "%-12s %5zu MB %10d %12.1f %14.0f %12.0f\n", chained ? "chained" : "independent", mb, waves, records / best * 1e-6,
                             best * 1e6 / steps, records * 32.0 / best * 1e-6 );
                std::fflush( stdout );
            }
        }
    CHECK( hipFree( table ) );
    CHECK( hipFree( out ) );
    return 0;
}
