// HIPRenderer (the GPU sibling of the reference's CPURenderer) and the single-ray
// intersectKdTree convenience entry; both go through the C-ABI of mrt.h.
#include "host_math.hpp"

#include <chrono>
#include <cmath>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <stdexcept>

namespace mrt {

// The tree's own device scene, on `device`: created on first use, re-created when another device
// is asked for, released by ~KdTree.
static mrt_scene* device_scene_for( Scene* scene, KdTree* tree, int device ) {
    if ( !tree->flat ) tree->flat = flatten( scene, tree );
    if ( tree->device_scene && tree->device_of_scene != device ) {
        mrt_scene_destroy( tree->device_scene );
        tree->device_scene = nullptr;
    }
    if ( !tree->device_scene ) {
        mrt_status st = mrt_scene_create( &tree->flat->desc, device, &tree->device_scene );
        if ( st != MRT_OK ) throw std::runtime_error( std::string( "mrt_scene_create: " ) + mrt_last_error() );
        tree->device_of_scene = device;
    }
    return tree->device_scene;
}

bool intersectKdTree( Scene* scene, KdTree* tree, Ray* ray, Intersection* intersection ) {
    mrt_scene* ds = device_scene_for( scene, tree, tree->device_of_scene >= 0 ? tree->device_of_scene : 0 );
    mrt_rays r { &ray->orig.x, &ray->orig.y, &ray->orig.z, &ray->dir.x, &ray->dir.y, &ray->dir.z, &ray->tmin, &ray->tmax };
    int32_t prim = -1;
    float t = 0, u = 0, v = 0;
    mrt_hits h { &prim, &t, &u, &v };
    mrt_status st = mrt_intersect( ds, 1, &r, &h, ray->shadow ? MRT_ANY_HIT : MRT_CLOSEST_HIT, MRT_BUF_HOST, nullptr, nullptr );
    if ( st != MRT_OK ) throw std::runtime_error( std::string( "mrt_intersect: " ) + mrt_last_error() );
    if ( prim < 0 ) return false;
    ray->tmax                   = t;
    intersection->hit           = true;
    intersection->prim          = prim;
    intersection->u             = u;
    intersection->v             = v;
    intersection->position      = ray->orig + ( t * ray->dir );
    intersection->materialIndex = scene->objects[prim]->m_MaterialIndex;
    return true;
}

// ------------------------------------------------------------------ Pathtracer (integrator.h:40-43)
Pathtracer::~Pathtracer() {
    for ( mrt_scene* s : m_owned )
        mrt_scene_destroy( s );
}

void Pathtracer::preprocess( Scene* scene, KdTree* tree, Sampler* sampler, const std::vector<int>& devices ) {
    if ( sampler ) m_sampler = sampler;
    auto* strat = dynamic_cast<StratifiedSampler*>( m_sampler );
    if ( !strat ) throw std::invalid_argument( "Pathtracer: the GPU integrator draws from a StratifiedSampler" );
    if ( !strat->jitterSamples ) throw std::invalid_argument( "Pathtracer: unjittered stratified samples are not supported" );
    if ( strat->nSampledDimensions != 8 ) throw std::invalid_argument( "Pathtracer: the kernels pre-stratify 8 dimensions (CPURenderer.cpp:15)" );
    if ( devices.empty() ) throw std::invalid_argument( "Pathtracer: no device" );
    for ( mrt_scene* s : m_owned )
        mrt_scene_destroy( s );
    m_owned.clear();
    m_devices.clear();
    m_tree = tree;
    m_devices.push_back( device_scene_for( scene, tree, devices[0] ) );
    for ( size_t i = 1; i < devices.size(); i++ ) {
        mrt_scene* s  = nullptr;
        mrt_status st = mrt_scene_create( &tree->flat->desc, devices[i], &s );
        if ( st != MRT_OK ) throw std::runtime_error( std::string( "mrt_scene_create: " ) + mrt_last_error() );
        m_owned.push_back( s );
        m_devices.push_back( s );
    }
}

void Pathtracer::trace_samples( const SampleRange& range, float* rgb_sum, mrt_render_stats* stats ) {
    if ( m_devices.empty() ) throw std::logic_error( "Pathtracer::trace_samples before preprocess" );
    auto* strat = static_cast<StratifiedSampler*>( m_sampler );
    mrt_render_params p {};
    p.width        = range.width;
    p.height       = range.height;
    p.spp_x        = (uint32_t)strat->xPixelSamples;
    p.spp_y        = (uint32_t)strat->yPixelSamples;
    p.sample_begin = range.sample_begin;
    p.sample_end   = range.sample_end;
    p.max_depth    = (uint32_t)m_maxDepth;
    p.seed         = range.seed;
    p.row_begin    = range.row_begin;
    p.row_end      = range.row_end;
    if ( p.sample_begin == p.sample_end ) return;
    mrt_status st;
    if ( m_devices.size() == 1 ) st = mrt_render( m_devices[0], &p, rgb_sum, MRT_BUF_HOST, nullptr, stats );
    else st = mrt_render_multi( m_devices.data(), (int32_t)m_devices.size(), &p, rgb_sum, stats );
    if ( st != MRT_OK ) throw std::runtime_error( std::string( "mrt_render: " ) + mrt_last_error() );
}

// ------------------------------------------------------------------ HIPRenderer
void HIPRenderer::init( std::string& imageName, vec2 resolution, Scene* scene ) {
    destroy();
    m_imageName = imageName;
    m_scene     = scene;
    m_w         = (uint32_t)resolution.x;
    m_h         = (uint32_t)resolution.y;
    m_sum.assign( (size_t)m_w * m_h * 3, 0.f );
    m_image.assign( (size_t)m_w * m_h * 3, 0.f );
    m_next = 0;
    std::printf( "Output resolution: %ux%u\n", m_w, m_h );
    m_kdTree = initKdTree( m_scene );
    // CPURenderer.cpp:15-19
    m_sampler = new StratifiedSampler( spp_x, spp_y, true, 8 );
    std::printf( "Spp: %d\n", (int)m_sampler->samplesPerPixel );
    m_integrator = new Pathtracer( maxDepth, m_sampler );
    m_integrator->preprocess( m_scene, m_kdTree, m_sampler, devices.empty() ? std::vector<int> { device } : devices );
}

void HIPRenderer::destroy() {
    delete m_integrator; // before the tree: it renders through the tree's device scene
    m_integrator = nullptr;
    delete m_sampler;
    m_sampler = nullptr;
    delete m_kdTree; // releases its flat description and device scene
    m_kdTree = nullptr;
    m_sum.clear();
    m_image.clear();
}

void HIPRenderer::resetAccumulation() {
    std::fill( m_sum.begin(), m_sum.end(), 0.f );
    m_next = 0;
}

bool HIPRenderer::Render( uint32_t n_samples ) {
    const uint32_t total = (uint32_t)( spp_x * spp_y );
    if ( m_next >= total || n_samples == 0 ) return false;
    SampleRange range;
    range.width = m_w, range.height = m_h, range.row_begin = 0, range.row_end = m_h, range.seed = seed;
    range.sample_begin = m_next;
    range.sample_end   = n_samples < total - m_next ? m_next + n_samples : total;
    mrt_render_stats rs {};
    m_integrator->trace_samples( range, m_sum.data(), &rs );
    m_next = range.sample_end;
    stats.paths += rs.paths, stats.rays_closest += rs.rays_closest, stats.rays_any += rs.rays_any, stats.rays_mis += rs.rays_mis, stats.passes += rs.passes;
    stats.generate_ms += rs.generate_ms, stats.extend_ms += rs.extend_ms, stats.shade_ms += rs.shade_ms, stats.connect_ms += rs.connect_ms, stats.connect_mis_ms += rs.connect_mis_ms;
    // average over the samples so far + gamma (CPURenderer.cpp:64-69, realtime/Renderer.cpp:86-92)
    const float spp = (float)m_next;
    for ( size_t i = 0; i < m_sum.size(); i++ )
        m_image[i] = powf( m_sum[i] / spp, 1.0f / 2.2 );
    return m_next < total;
}

void HIPRenderer::run() {
    auto t0 = std::chrono::steady_clock::now();
    resetAccumulation();
    stats = mrt_render_stats {};
    Render( (uint32_t)( spp_x * spp_y ) );
    seconds = std::chrono::duration<double>( std::chrono::steady_clock::now() - t0 ).count();
    std::printf( "Rendering took %gs\n", seconds );
    if ( !m_imageName.empty() ) savePNG( ( m_imageName + ".png" ).c_str(), m_image.data(), m_w, m_h ); // saveImage( m_renderImage, m_imageName ), CPURenderer.cpp:83
}

namespace {
struct CheckpointHeader {
    char magic[8]; // "MRTCKPT1"
    uint32_t width, height, spp_x, spp_y, seed, max_depth, next, reserved;
};
} // namespace

bool HIPRenderer::saveCheckpoint( const char* filename ) const {
    CheckpointHeader h {};
    std::memcpy( h.magic, "MRTCKPT1", 8 );
    h.width = m_w, h.height = m_h, h.spp_x = (uint32_t)spp_x, h.spp_y = (uint32_t)spp_y, h.seed = seed, h.max_depth = (uint32_t)maxDepth, h.next = m_next;
    FILE* f = std::fopen( filename, "wb" );
    if ( !f ) return false;
    bool ok = std::fwrite( &h, sizeof( h ), 1, f ) == 1 && std::fwrite( m_sum.data(), sizeof( float ), m_sum.size(), f ) == m_sum.size();
    return ( std::fclose( f ) == 0 ) && ok;
}

bool HIPRenderer::loadCheckpoint( const char* filename ) {
    FILE* f = std::fopen( filename, "rb" );
    if ( !f ) return false;
    CheckpointHeader h {};
    bool ok = std::fread( &h, sizeof( h ), 1, f ) == 1 && std::memcmp( h.magic, "MRTCKPT1", 8 ) == 0;
    // a checkpoint continues only the frame it was taken from
    ok = ok && h.width == m_w && h.height == m_h && h.spp_x == (uint32_t)spp_x && h.spp_y == (uint32_t)spp_y && h.seed == seed &&
         h.max_depth == (uint32_t)maxDepth && h.next <= (uint32_t)( spp_x * spp_y );
    std::vector<float> sum( m_sum.size() );
    ok = ok && std::fread( sum.data(), sizeof( float ), sum.size(), f ) == sum.size();
    std::fclose( f );
    if ( !ok ) return false;
    m_sum.swap( sum );
    m_next = h.next;
    const float spp = m_next ? (float)m_next : 1.f;
    for ( size_t i = 0; i < m_sum.size(); i++ )
        m_image[i] = powf( m_sum[i] / spp, 1.0f / 2.2 );
    return true;
}

} // namespace mrt
