// The reference's main.cpp (main.cpp:30-41: parseScene, a renderer's init / run / destroy, freeScene)
// against this repository's host library: the same calls with the mrt:: types, HIPRenderer in the
// place of CPURenderer, and what the reference hard-codes (scene 18, 1200 x 800, 32 x 32 samples,
// depth 5) as optional arguments.
//   usage: mrt_main [scene id] [width height] [spp_x spp_y] [image name] [assets directory]
#include "mrt_host.hpp"

#include <cstdio>
#include <cstdlib>
#include <exception>
#include <string>

int main( int argc, char** argv ) {
    const int scene_id   = argc > 1 ? std::atoi( argv[1] ) : 18;
    const int width      = argc > 3 ? std::atoi( argv[2] ) : 1200;
    const int height     = argc > 3 ? std::atoi( argv[3] ) : 800;
    const int spp_x      = argc > 5 ? std::atoi( argv[4] ) : 32;
    const int spp_y      = argc > 5 ? std::atoi( argv[5] ) : 32;
    std::string image    = argc > 6 ? argv[6] : "output";
    const char* assets   = argc > 7 ? argv[7] : nullptr;
    try {
        mrt::Scene* scene = mrt::parseScene( scene_id, width, height, assets );
        mrt::HIPRenderer* renderer = new mrt::HIPRenderer;
        renderer->spp_x = spp_x;
        renderer->spp_y = spp_y;
        renderer->init( image, mrt::vec2 { (float)width, (float)height }, scene );
        renderer->run();
        std::printf( "%.1f Mrays/s\n", ( renderer->stats.rays_closest + renderer->stats.rays_any ) / renderer->seconds / 1e6 );
        renderer->destroy();
        delete renderer;
        mrt::freeScene( scene );
    }
    catch ( const std::exception& e ) {
        std::fprintf( stderr, "mrt_main: %s\n", e.what() );
        return 1;
    }
    return 0;
}
