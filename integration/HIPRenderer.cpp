#include "HIPRenderer.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <iostream>

static void mrtCheck( mrt_status st, const char* what ) {
    if ( st == MRT_OK ) return;
    std::fprintf( stderr, "%s: %s\n", what, mrt_last_error() );
    std::exit( 1 ); // the reference's own error behaviour (scene.cpp:191)
}

void HIPRenderer::init( std::string& imageName, glm::vec2 resolution, Scene* scene ) {
    m_imageName   = imageName;
    m_scene       = scene;
    m_renderImage = initImage( resolution.x, resolution.y );
    std::cout << "Output resolution: " << m_renderImage->width << "x" << m_renderImage->height << std::endl;
    m_kdTree = initKdTree( m_scene ); // the SAH build stays on the CPU (kdtree.cpp:54-158)
    std::cout << "Spp: " << spp_x * spp_y << std::endl;
    mrtFlattenGeometry( m_scene, m_kdTree, m_flat );
    mrtFlattenLights( m_scene, m_flat );
    for ( int d : devices ) {
        mrt_scene* s = nullptr;
        mrtCheck( mrt_scene_create( &m_flat.desc, d, &s ), "mrt_scene_create" );
        m_dev.push_back( s );
    }
}

void HIPRenderer::run() {
    auto startTime      = std::chrono::system_clock::now();
    const uint32_t w    = (uint32_t)m_renderImage->width, h = (uint32_t)m_renderImage->height;
    const size_t floats = (size_t)w * h * 3;
    std::vector<float> sum( floats, 0.f ), img( floats );
    mrt_render_params p {};
    p.width = w, p.height = h;
    p.spp_x = (uint32_t)spp_x, p.spp_y = (uint32_t)spp_y; // StratifiedSampler(32, 32, true, 8), CPURenderer.cpp:15
    p.max_depth = (uint32_t)maxDepth;                    // Pathtracer(5, sampler), CPURenderer.cpp:18
    p.seed      = seed ? seed : (unsigned)time( NULL );  // CPURenderer.cpp:42
    // the pixel and sample loops of CPURenderer.cpp:30-62, summed per pixel
    if ( m_dev.size() == 1 ) mrtCheck( mrt_render( m_dev[0], &p, sum.data(), MRT_BUF_HOST, nullptr, nullptr ), "mrt_render" );
    else mrtCheck( mrt_render_multi( m_dev.data(), (int32_t)m_dev.size(), &p, sum.data(), nullptr ), "mrt_render_multi" );
    // average and gamma of CPURenderer.cpp:64-69
    mrtCheck( mrt_resolve( m_dev[0], (uint64_t)w * h, sum.data(), (uint32_t)( spp_x * spp_y ), img.data(), MRT_BUF_HOST, nullptr ), "mrt_resolve" );
    for ( uint32_t j = 0; j < h; j++ )
        for ( uint32_t i = 0; i < w; i++ ) {
            const float* c                     = &img[3 * ( (size_t)j * w + i )];
            *getPixelPtr( m_renderImage, i, j ) = color3( c[0], c[1], c[2] );
        }
    auto stopTime = std::chrono::system_clock::now();
    std::cout << "Rendering took " << std::chrono::duration_cast<std::chrono::duration<double>>( stopTime - startTime ).count() << "s" << std::endl;
    saveImage( m_renderImage, m_imageName.data() );
}

void HIPRenderer::destroy() {
    for ( mrt_scene* s : m_dev )
        mrt_scene_destroy( s );
    m_dev.clear();
    freeImage( m_renderImage );
    m_renderImage = nullptr;
}
