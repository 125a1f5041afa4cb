// Sample streams of the gfx950 path tracer.
//
// The reference draws from a pbrt-style StratifiedSampler(spp_x, spp_y, jitter, 8 dimensions)
// (sampling/stratified.cpp:38-68, sampling/sampler.cpp:97-109) whose state is one sequential PCG32
// stream per thread, and from a thread-local `uniform01(engine)` (defines.h:63-64).  Neither can be
// evaluated for sample s of pixel p without replaying every earlier draw, so the kernels use the
// counter-based form of the same sampler:
//   * for each of the first 8 one-dimensional and 8 two-dimensional dimensions of a pixel, sample s
//     takes stratum permute(s, n, hash(pixel, dim)) -- a hash-built random permutation of the
//     n = spp_x*spp_y strata (Kensler, "Correlated Multi-Jittered Sampling", Pixar TM 13-01) -- and
//     a jitter hashed from (stratum, pixel, dim); the value is min((stratum + jitter) / n, 1-eps)
//     (sampling.cpp:36-55);
//   * dimensions beyond the 8th and every `uniform01(engine)` draw come from one PCG32 sequence
//     (sampling/rng.h:57-134) per camera sample, keyed by (seed, pixel, sample).
#pragma once
#include "device_math.hpp"

#define MRT_ONE_MINUS_EPS 0x1.fffffep-1f

MRT_DEV uint64_t mix64( uint64_t z ) {
    z = ( z ^ ( z >> 30 ) ) * 0xbf58476d1ce4e5b9ULL;
    z = ( z ^ ( z >> 27 ) ) * 0x94d049bb133111ebULL;
    return z ^ ( z >> 31 );
}

MRT_DEV uint32_t kensler_permute( uint32_t i, uint32_t l, uint32_t p ) {
    uint32_t w = l - 1;
    w |= w >> 1;
    w |= w >> 2;
    w |= w >> 4;
    w |= w >> 8;
    w |= w >> 16;
    do {
        i ^= p;
        i *= 0xe170893d;
        i ^= p >> 16;
        i ^= ( i & w ) >> 4;
        i ^= p >> 8;
        i *= 0x0929eb3f;
        i ^= p >> 23;
        i ^= ( i & w ) >> 1;
        i *= 1 | p >> 27;
        i *= 0x6935fa69;
        i ^= ( i & w ) >> 11;
        i *= 0x74dcb303;
        i ^= ( i & w ) >> 2;
        i *= 0x9e501cc3;
        i ^= ( i & w ) >> 2;
        i *= 0xc860a3df;
        i &= w;
        i ^= i >> 5;
    } while ( i >= l );
    return ( i + p ) % l;
}

MRT_DEV float u32_to_unit( uint32_t u ) {
    return fmin_glm( MRT_ONE_MINUS_EPS, (float)u * 0x1p-32f );
}

struct Pcg32 {
    uint64_t state, inc;
};
MRT_DEV uint32_t pcg_next( Pcg32& r ) {
    uint64_t oldstate   = r.state;
    r.state             = oldstate * 0x5851f42d4c957f2dULL + r.inc;
    uint32_t xorshifted = (uint32_t)( ( ( oldstate >> 18u ) ^ oldstate ) >> 27u );
    uint32_t rot        = (uint32_t)( oldstate >> 59u );
    return ( xorshifted >> rot ) | ( xorshifted << ( ( ~rot + 1u ) & 31 ) );
}
MRT_DEV void pcg_set_sequence( Pcg32& r, uint64_t initseq ) {
    r.state = 0u;
    r.inc   = ( initseq << 1u ) | 1u;
    pcg_next( r );
    r.state += 0x853c49e6748fea9bULL;
    pcg_next( r );
}
MRT_DEV float pcg_float( Pcg32& r ) {
    return u32_to_unit( pcg_next( r ) );
}

MRT_DEV uint64_t pixel_key( uint32_t seed, uint32_t pixel ) {
    return mix64( ( (uint64_t)seed << 32 ) ^ (uint64_t)pixel );
}
MRT_DEV uint64_t sample_sequence( uint64_t key, uint32_t sample ) {
    return mix64( key + 0xD1B54A32D192ED03ULL * ( (uint64_t)sample + 1 ) );
}
MRT_DEV uint64_t dim_hash( uint64_t key, uint32_t tag ) {
    return mix64( key + 0x9E3779B97F4A7C15ULL * ( (uint64_t)tag + 1 ) );
}
// 1D dimension d (0..7) of sample s
MRT_DEV float strat1d( uint64_t key, uint32_t d, uint32_t s, uint32_t n ) {
    uint64_t h  = dim_hash( key, d );
    uint32_t st = kensler_permute( s, n, (uint32_t)h );
    uint64_t j  = mix64( h ^ ( 0xA0761D6478BD642FULL * ( (uint64_t)st + 1 ) ) );
    float delta = u32_to_unit( (uint32_t)j );
    float inv   = 1.0f / (float)(int)n;
    return fminf( ( (float)(int)st + delta ) * inv, MRT_ONE_MINUS_EPS );
}
// 2D dimension d (0..7) of sample s
MRT_DEV V2 strat2d( uint64_t key, uint32_t d, uint32_t s, uint32_t nx, uint32_t ny ) {
    uint64_t h  = dim_hash( key, 16 + d );
    uint32_t st = kensler_permute( s, nx * ny, (uint32_t)h );
    uint64_t j  = mix64( h ^ ( 0xA0761D6478BD642FULL * ( (uint64_t)st + 1 ) ) );
    float jx    = u32_to_unit( (uint32_t)j );
    float jy    = u32_to_unit( (uint32_t)( j >> 32 ) );
    int x       = (int)( st % nx );
    int y       = (int)( st / nx );
    float dx = 1.0f / (float)(int)nx, dy = 1.0f / (float)(int)ny;
    V2 r;
    r.x = fminf( ( (float)x + jx ) * dx, MRT_ONE_MINUS_EPS );
    r.y = fminf( ( (float)y + jy ) * dy, MRT_ONE_MINUS_EPS );
    return r;
}
