// TEST INFRASTRUCTURE ONLY (CPU oracle) -- see mrt_oracle.h.
//
// (1) PCG32 and the sequential StratifiedSampler of the reference, restated
//     (sampling/rng.h:57-134, sampling/sampler.cpp:43-121, sampling/stratified.cpp:38-68,
//      sampling/sampling.cpp:36-55, sampling/sampling.h:159-167).
// (2) The counter-based form of the same sampler that the HIP kernels use.  The reference
//     sampler is one sequential RNG stream per thread, seeded from time(NULL): sample s of
//     pixel p depends on every draw made before it, which a GPU cannot reproduce and the
//     reference itself does not reproduce from run to run.  The counter-based form keeps what
//     the sampler guarantees -- each of the first 8 1D and 8 2D dimensions of a pixel is a
//     jittered stratified set over the spp_x * spp_y strata visited in a random order, later
//     dimensions are independent uniforms -- but computes sample s directly:
//         stratum  = permute(s, n, hash(pixel, dim))         (Kensler 2013 hash permutation)
//         jitter   = hash(stratum, pixel, dim)
//     The draws the reference takes from `uniform01(engine)` (defines.h:63-64; Material.cpp:1030,
//     1045; sampling.h:199-211) and from dimensions beyond the 8th come from one PCG32 sequence
//     per camera sample, keyed by (seed, pixel, sample).
#pragma once
#include "orc_math.h"

#include <cstdint>
#include <memory>
#include <vector>

namespace orc {

static const float OneMinusEpsilon = 0x1.fffffep-1; // rng.h:41,47

struct RNG { // rng.h:57-134
    uint64_t state = 0x853c49e6748fea9bULL, inc = 0xda3e39cb94b95bdbULL;
    RNG() {}
    explicit RNG( uint64_t seq ) { SetSequence( seq ); }
    void SetSequence( uint64_t initseq ) {
        state = 0u;
        inc   = ( initseq << 1u ) | 1u;
        UniformUInt32();
        state += 0x853c49e6748fea9bULL;
        UniformUInt32();
    }
    uint32_t UniformUInt32() {
        uint64_t oldstate   = state;
        state               = oldstate * 0x5851f42d4c957f2dULL + inc;
        uint32_t xorshifted = (uint32_t)( ( ( oldstate >> 18u ) ^ oldstate ) >> 27u );
        uint32_t rot        = (uint32_t)( oldstate >> 59u );
        return ( xorshifted >> rot ) | ( xorshifted << ( ( ~rot + 1u ) & 31 ) );
    }
    uint32_t UniformUInt32( uint32_t b ) {
        uint32_t threshold = ( ~b + 1u ) % b;
        while ( true ) {
            uint32_t r = UniformUInt32();
            if ( r >= threshold ) return r % b;
        }
    }
    float Uniformfloat() { return fmin_glm( OneMinusEpsilon, float( UniformUInt32() * 0x1p-32f ) ); }
};

// ------------------------------------------------------------------ sequential sampler
struct StratifiedSampler {
    int xs, ys, ndims;
    int64_t spp;
    std::vector<std::vector<float>> samples1D;
    std::vector<std::vector<V2>> samples2D;
    int cur1D = 0, cur2D = 0;
    int64_t index = 0;
    RNG rng;

    StratifiedSampler( int x, int y, int nd, int seed ) : xs( x ), ys( y ), ndims( nd ), spp( (int64_t)x * y ) {
        for ( int i = 0; i < nd; i++ ) {
            samples1D.push_back( std::vector<float>( spp ) );
            samples2D.push_back( std::vector<V2>( spp ) );
        }
        rng.SetSequence( seed ); // StratifiedSampler::Clone (stratified.cpp:64-68)
    }
    static void StratifiedSample1D( float* samp, int n, RNG& rng ) { // sampling.cpp:36-42
        float inv = (float)1 / n;
        for ( int i = 0; i < n; ++i ) {
            float delta = rng.Uniformfloat();
            samp[i]     = std::min( ( i + delta ) * inv, OneMinusEpsilon );
        }
    }
    static void StratifiedSample2D( V2* samp, int nx, int ny, RNG& rng ) { // sampling.cpp:44-55
        float dx = (float)1 / nx, dy = (float)1 / ny;
        for ( int y = 0; y < ny; ++y )
            for ( int x = 0; x < nx; ++x ) {
                float jx = rng.Uniformfloat();
                float jy = rng.Uniformfloat();
                samp->x  = std::min( ( x + jx ) * dx, OneMinusEpsilon );
                samp->y  = std::min( ( y + jy ) * dy, OneMinusEpsilon );
                ++samp;
            }
    }
    template <typename T>
    static void Shuffle( T* samp, int count, RNG& rng ) { // sampling.h:159-167, nDimensions = 1
        for ( int i = 0; i < count; ++i ) {
            int other = i + rng.UniformUInt32( count - i );
            std::swap( samp[i], samp[other] );
        }
    }
    void StartPixel() { // stratified.cpp:38-62 (no sample arrays are ever requested)
        for ( size_t i = 0; i < samples1D.size(); ++i ) {
            StratifiedSample1D( &samples1D[i][0], xs * ys, rng );
            Shuffle( &samples1D[i][0], xs * ys, rng );
        }
        for ( size_t i = 0; i < samples2D.size(); ++i ) {
            StratifiedSample2D( &samples2D[i][0], xs, ys, rng );
            Shuffle( &samples2D[i][0], xs * ys, rng );
        }
        index = 0;
        cur1D = cur2D = 0;
    }
    bool StartNextSample() { // sampler.cpp:86-89, 56-60
        cur1D = cur2D = 0;
        return ++index < spp;
    }
    float Get1D() { // sampler.cpp:97-102
        if ( cur1D < (int)samples1D.size() ) return samples1D[cur1D++][index];
        return rng.Uniformfloat();
    }
    V2 Get2D() { // sampler.cpp:104-109
        if ( cur2D < (int)samples2D.size() ) return samples2D[cur2D++][index];
        // point2( rng.Uniformfloat(), rng.Uniformfloat() ): the two calls are unsequenced; g++
        // evaluates constructor arguments right to left, so y is drawn first (pinned by the golden
        // stream recorded from the reference's own sampler.cpp)
        float b = rng.Uniformfloat();
        float a = rng.Uniformfloat();
        return V2 { a, b };
    }
};

// ------------------------------------------------------------------ counter-based sampler
inline uint64_t mix64( uint64_t z ) { // splitmix64 finaliser
    z = ( z ^ ( z >> 30 ) ) * 0xbf58476d1ce4e5b9ULL;
    z = ( z ^ ( z >> 27 ) ) * 0x94d049bb133111ebULL;
    return z ^ ( z >> 31 );
}

// Kensler, "Correlated Multi-Jittered Sampling", Pixar TM 13-01: a hash-built permutation of
// [0, l) evaluated at i.
inline uint32_t permute( uint32_t i, uint32_t l, uint32_t p ) {
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

inline float u32_to_unit( uint32_t u ) { // RNG::Uniformfloat's mapping (rng.h:72-78)
    return fmin_glm( OneMinusEpsilon, float( u * 0x1p-32f ) );
}

struct CounterSampler {
    uint64_t key;   // mix64(seed, pixel)
    uint32_t nx, ny, n, s;
    RNG rng;        // per camera sample: engine draws + dimensions beyond the 8th
    int cur1D = 0, cur2D = 0;

    CounterSampler( uint32_t seed, uint32_t pixel, uint32_t spp_x, uint32_t spp_y, uint32_t sample ) :
        nx( spp_x ), ny( spp_y ), n( spp_x * spp_y ), s( sample ) {
        key = mix64( ( (uint64_t)seed << 32 ) ^ (uint64_t)pixel );
        rng.SetSequence( mix64( key + 0xD1B54A32D192ED03ULL * ( (uint64_t)sample + 1 ) ) );
    }
    // dimension tag: 1D dimension d -> d, 2D dimension d -> 16 + d
    uint64_t dim_hash( uint32_t tag ) const { return mix64( key + 0x9E3779B97F4A7C15ULL * ( (uint64_t)tag + 1 ) ); }
    float strat1D( uint32_t d ) const {
        uint64_t h    = dim_hash( d );
        uint32_t st   = permute( s, n, (uint32_t)h );
        uint64_t j    = mix64( h ^ ( 0xA0761D6478BD642FULL * ( (uint64_t)st + 1 ) ) );
        float delta   = u32_to_unit( (uint32_t)j );
        float inv     = (float)1 / (int)n;
        return std::min( ( (int)st + delta ) * inv, OneMinusEpsilon );
    }
    V2 strat2D( uint32_t d ) const {
        uint64_t h  = dim_hash( 16 + d );
        uint32_t st = permute( s, n, (uint32_t)h );
        uint64_t j  = mix64( h ^ ( 0xA0761D6478BD642FULL * ( (uint64_t)st + 1 ) ) );
        float jx    = u32_to_unit( (uint32_t)j );
        float jy    = u32_to_unit( (uint32_t)( j >> 32 ) );
        int x       = (int)( st % nx );
        int y       = (int)( st / nx );
        float dx = (float)1 / (int)nx, dy = (float)1 / (int)ny;
        return V2 { std::min( ( x + jx ) * dx, OneMinusEpsilon ), std::min( ( y + jy ) * dy, OneMinusEpsilon ) };
    }
    float Get1D() {
        if ( cur1D < 8 ) return strat1D( cur1D++ );
        return rng.Uniformfloat();
    }
    V2 Get2D() {
        if ( cur2D < 8 ) return strat2D( cur2D++ );
        float a = rng.Uniformfloat();
        float b = rng.Uniformfloat();
        return V2 { a, b };
    }
    // A Get2D whose value the integrator never reads (uScattering and the argument of
    // Material::sample are ignored by the reference, Material.cpp:1045): the dimension is
    // consumed, nothing is drawn.
    void Skip2D() {
        if ( cur2D < 8 ) cur2D++;
    }
    float engine01() { return rng.Uniformfloat(); }
};

} // namespace orc
