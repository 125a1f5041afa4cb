// Host-side sanitizer run: the product's host library sources compiled with
// -fsanitize=address,undefined into this one executable (raytracer_amd/build.py: build_sanitize),
// exercising scene set-up, the loaders, the kd-tree build, flatten and the C-ABI's host-side
// validation.  No GPU is touched: mrt_scene_create is expected to stop at "no device" or earlier.
// usage: host_sanitize <assets directory>
#include "host_math.hpp" // the host library's internal header: mrt_host.hpp + the PNG decoder
#include "mrt_host_c.h"

#include <cstdio>
#include <fstream>
#include <iterator>
#include <vector>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>

using namespace mrt;

#define CHECK( cond )                                                        \
    do {                                                                     \
        if ( !( cond ) ) {                                                   \
            std::fprintf( stderr, "%s:%d: %s\n", __FILE__, __LINE__, #cond ); \
            return 1;                                                        \
        }                                                                    \
    } while ( 0 )

static int mesh_scene( const std::string& assets, const char* obj, float prescale, int subdivisions, int threads, size_t expect_objects ) {
    Scene* scene = initScene();
    Material& m  = scene->CreateMaterial( MaterialModel::COOK_TORRANCE );
    size_t id    = m.m_UID;
    Transform t;
    t.scale( 0.5f, 0.5f, 0.5f );
    t.rotate( vec3( 0, 1, 0 ), 30.f );
    t.translate( vec3( 0.1f, 0.2f, 0.3f ) );
    addObjectsFromFileEx( ( assets + "/" + obj ).c_str(), scene, id, t, prescale, subdivisions );
    Transform pt;
    pt.scale( 5, 5, 1 );
    pt.rotate( vec3( -1, 0, 0 ), 90.f );
    addObject( scene, new Plane( id, pt ) );
    Transform st;
    st.scale( 0.3f, 0.3f, 0.3f );
    st.translate( vec3( 1, 1, 1 ) );
    Object* sphere = initSphere( (int)id, st );
    addObject( scene, sphere );
    addLight( scene, new ShapeLight( vec3( 0.f ), color3( 1, 1, 1 ), initSphere( (int)id, st ) ) );
    setSimpleCamera( scene, point3( 0, 1, 5 ), vec3( 0, 0, 0 ), vec3( 0, 1, 0 ), 60.f, 64.f, 48.f, 0.01f, 5.f );
    CHECK( scene->objects.size() == expect_objects + 2 );
    KdTree* tree = initKdTree( scene, threads );
    CHECK( tree->root != nullptr );
    for ( NodeLayout layout : { LAYOUT_PREORDER_PAIRS, LAYOUT_BFS_TOP } ) {
        std::unique_ptr<FlatScene> flat( flatten( scene, tree, layout ) );
        CHECK( flat->desc.n_objects == scene->objects.size() );
        CHECK( flat->desc.n_nodes > 0 && flat->desc.n_out_of_tree == 1 );
        // host-side validation runs to the end; without a GPU the call stops at the device
        mrt_scene* dev = nullptr;
        mrt_status st  = mrt_scene_create( &flat->desc, 0, &dev );
        if ( st == MRT_OK ) mrt_scene_destroy( dev );
        else CHECK( st == MRT_ERR_DEVICE );
        // malformed: a child that points backwards
        mrt_kdnode saved = flat->nodes[0];
        if ( ( saved.w1 & 3u ) != 3u ) {
            flat->nodes[0].w1 = ( saved.w1 & 3u ) | ( 0u << 2 );
            CHECK( mrt_scene_create( &flat->desc, 0, &dev ) == MRT_ERR_INVALID );
            flat->nodes[0] = saved;
        }
    }
    {
        // seeded corruptions of the tree and the per-object tables: validation reads nothing out of
        // bounds (a sanitizer report otherwise) and answers with a status every time
        std::unique_ptr<FlatScene> flat( flatten( scene, tree ) );
        uint32_t state = 2024u + (uint32_t)expect_objects;
        auto next      = [&state]() { return state = state * 1664525u + 1013904223u; };
        for ( int round = 0; round < 120; round++ ) {
            std::vector<mrt_kdnode> nodes( flat->desc.nodes, flat->desc.nodes + flat->desc.n_nodes );
            std::vector<uint32_t> refs( flat->desc.refs, flat->desc.refs + flat->desc.n_refs );
            std::vector<uint32_t> mats( flat->desc.obj_material, flat->desc.obj_material + flat->desc.n_objects );
            std::vector<uint8_t> kinds( flat->desc.obj_kind, flat->desc.obj_kind + flat->desc.n_objects );
            for ( int k = 0; k < 3; k++ ) {
                const uint32_t what = next() >> 8, at = next() >> 4, value = ( next() & 1u ) ? next() : ( 1u << ( next() >> 27 ) );
                switch ( what % 5u ) {
                case 0: nodes[at % nodes.size()].w0 ^= value; break;
                case 1: nodes[at % nodes.size()].w1 ^= value; break;
                case 2: refs[at % refs.size()] ^= value; break;
                case 3: mats[at % mats.size()] ^= value; break;
                default: kinds[at % kinds.size()] ^= (uint8_t)value; break;
                }
            }
            mrt_scene_desc d = flat->desc;
            d.nodes = nodes.data(), d.refs = refs.data(), d.obj_material = mats.data(), d.obj_kind = kinds.data();
            mrt_scene* dev = nullptr;
            mrt_status st  = mrt_scene_create( &d, 0, &dev );
            if ( st == MRT_OK ) mrt_scene_destroy( dev );
            else CHECK( st == MRT_ERR_INVALID || st == MRT_ERR_UNSUPPORTED || st == MRT_ERR_DEVICE );
        }
    }
    delete tree;
    freeScene( scene );
    return 0;
}

int main( int argc, char** argv ) {
    if ( argc < 2 ) {
        std::fprintf( stderr, "usage: %s <assets directory>\n", argv[0] );
        return 2;
    }
    const std::string assets = argv[1];
    if ( mesh_scene( assets, "cornell-box.obj", 1.f, 0, 1, 32 ) ) return 1;
    if ( mesh_scene( assets, "bunny.obj", 1.f, 0, 3, 4968 ) ) return 1;
    if ( mesh_scene( assets, "veach_mi/floor.obj", 2.f, 2, 8, 64 ) ) return 1;
    if ( mesh_scene( assets, "indoor_plant_02.obj", 4.f, 0, 4, 45840 ) ) return 1;
    {
        // loaders
        IBL ibl( assets + "/brown_photostudio_06_1k.hdr" );
        CHECK( ibl.m_width == 1024 && ibl.m_height == 512 && ibl.m_pixels.size() == 3u * 1024 * 512 );
        std::unique_ptr<texture> tex( image_texture( ( assets + "/indoor_plant_2_COL_1k.ppm" ).c_str() ) );
        CHECK( tex && tex->width == 1024 && tex->texels.size() == 3u * 1024 * 1024 );
        CHECK( image_texture( ( assets + "/cornell-box.obj" ).c_str() ) == nullptr ); // not a PPM
        bool threw = false;
        try {
            IBL bad( assets + "/cornell-box.obj" );
        }
        catch ( const std::runtime_error& ) {
            threw = true;
        }
        CHECK( threw );
        threw = false;
        try {
            Scene* s    = initScene();
            Material& m = s->CreateMaterial( MaterialModel::COOK_TORRANCE );
            Transform t;
            try {
                addObjectsFromFile( ( assets + "/does-not-exist.obj" ).c_str(), s, m, t );
            }
            catch ( ... ) {
                freeScene( s );
                throw;
            }
            freeScene( s );
        }
        catch ( const std::runtime_error& ) {
            threw = true;
        }
        CHECK( threw );
    }
    {
        // OBJ reader on files the reference's loader reads out of bounds for: faces of fewer than three
        // corners, indices outside the tables, a face before any vertex, bytes that are no text.
        // Either a scene or a std::runtime_error, never a sanitizer report.
        const char* cases[] = {
            "v 0 0 0\nv 1 0 0\nv 0 1 0\nf\nf 1\nf 1 2\nf 1 2 3\n",
            "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 4\n",
            "f 1 2 3\nv 0 0 0\n",
            "v 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nf 1//1 2//2 3//-5\n",
            "v 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nf 1/9 2/1 3/1\n",
            "v 0 0 0\nv 1 0 0\nv 0 1 0\nf -1 -2 -7\n",
            "v 1e999 -1e-999 1e+\nv . - +\nv 1 2\nf 1 2 3\nusemtl\nmtllib\ng\no\n",
            "\x01\x02\xff\xfe f 1 2 3 \x00 v v v\rf 2147483648 -2147483649 99999999999999999999\r",
        };
        const std::string path = "/tmp/mrt_host_sanitize_case.obj";
        for ( const char* text : cases ) {
            {
                std::ofstream out( path, std::ios::binary );
                out << text;
            }
            Scene* s    = initScene();
            Material& m = s->CreateMaterial( MaterialModel::COOK_TORRANCE );
            Transform t;
            try {
                addObjectsFromFile( path.c_str(), s, m, t );
            }
            catch ( const std::runtime_error& ) {
            }
            freeScene( s );
        }
        std::remove( path.c_str() );
    }
    {
        // PNG reader: the texture of scene 18, then the same bytes cut short and damaged (every
        // outcome but a clean decode or a clean refusal is a sanitizer report)
        const std::string png = assets + "/pf-s125-ake7011-a.png";
        std::unique_ptr<texture> tex( image_texture( png.c_str() ) );
        CHECK( tex && tex->width == 800 && tex->height == 533 && tex->texels.size() == 3u * 800 * 533 );
        std::ifstream f( png, std::ios::binary );
        std::vector<unsigned char> bytes( ( std::istreambuf_iterator<char>( f ) ), std::istreambuf_iterator<char>() );
        uint32_t w = 0, h = 0;
        std::vector<unsigned char> rgb;
        CHECK( decode_png_rgb8( bytes.data(), bytes.size(), w, h, rgb ) && w == 800 && h == 533 );
        for ( size_t cut : { (size_t)0, (size_t)7, (size_t)33, (size_t)57, (size_t)1000, bytes.size() / 2, bytes.size() - 13 } )
            CHECK( !decode_png_rgb8( bytes.data(), cut, w, h, rgb ) );
        uint32_t state = 12345u;
        for ( int round = 0; round < 40; round++ ) {
            std::vector<unsigned char> bad = bytes;
            for ( int k = 0; k < 8; k++ ) {
                state = state * 1664525u + 1013904223u;
                bad[41 + ( state >> 8 ) % ( bad.size() - 41 )] ^= (unsigned char)( 1u << ( state & 7u ) );
            }
            decode_png_rgb8( bad.data(), bad.size(), w, h, rgb ); // either outcome
        }
        // headers that promise far more than the file holds: refused before anything is allocated for them
        // (the IHDR checksum is not checked by the reader, so patching the fields is enough)
        for ( const uint32_t dim : { 0x7fffffffu, 0x00ffffffu, 65536u, 46341u } ) {
            std::vector<unsigned char> big = bytes;
            for ( int k = 0; k < 4; k++ )
                big[16 + k] = big[20 + k] = (unsigned char)( dim >> ( 24 - 8 * k ) ); // width, height
            big[24] = 16, big[25] = 6; // 16 bits, RGBA: 64 bits per pixel
            CHECK( !decode_png_rgb8( big.data(), big.size(), w, h, rgb ) );
        }
        {
            // a 50-byte PPM whose header claims 2^31 - 1 by 2^31 - 1 pixels, and one within the size limit
            // of a texture that still claims more than the file holds
            for ( const char* header : { "P6 2147483647 2147483647 255\n", "P6 60000 60000 255\n", "P6 65536 1 255\n" } ) {
                const std::string path = "/tmp/mrt_host_sanitize_big.ppm";
                {
                    std::ofstream o( path, std::ios::binary );
                    o << header << "0123456789abcdefghij";
                }
                CHECK( image_texture( path.c_str() ) == nullptr );
                uint32_t iw = 0, ih = 0;
                CHECK( mrth_load_image( path.c_str(), &iw, &ih, nullptr, 0 ) == -1 );
                std::remove( path.c_str() );
            }
        }
    }
    {
        // parseScene over the scene data files, and a file with an error in it
        for ( int id : { 9, 16, 17, 18, 19, 20, 21 } ) {
            Scene* scene = parseScene( id, 120, 80, assets.c_str() );
            CHECK( scene && scene->cam && !scene->objects.empty() );
            KdTree* tree = initKdTree( scene, 2 );
            std::unique_ptr<FlatScene> flat( flatten( scene, tree ) );
            CHECK( flat->desc.n_objects == scene->objects.size() );
            delete tree;
            freeScene( scene );
        }
        bool threw = false;
        try {
            parseScene( 12, 120, 80, assets.c_str() );
        }
        catch ( const std::invalid_argument& ) {
            threw = true;
        }
        CHECK( threw );
        threw = false;
        try {
            loadSceneFile( ( assets + "/cornell-box.mtl" ).c_str(), 64, 64, assets.c_str() ); // not a scene file
        }
        catch ( const std::runtime_error& ) {
            threw = true;
        }
        CHECK( threw );
    }
    {
        // Radiance HDR and PPM readers on damaged and truncated files: a picture, a null texture or an
        // exception
        struct Case {
            std::string file;
            bool hdr;
        };
        for ( const Case& c : { Case { assets + "/brown_photostudio_06_1k.hdr", true }, Case { assets + "/indoor_plant_2_COL_1k.ppm", false } } ) {
            std::ifstream f( c.file, std::ios::binary );
            std::vector<char> bytes( ( std::istreambuf_iterator<char>( f ) ), std::istreambuf_iterator<char>() );
            CHECK( bytes.size() > 1000 );
            const std::string path = c.hdr ? "/tmp/mrt_host_sanitize_case.hdr" : "/tmp/mrt_host_sanitize_case.ppm";
            uint32_t state         = 4242u;
            auto next              = [&state]() { return state = state * 1664525u + 1013904223u; };
            for ( int round = 0; round < 30; round++ ) {
                std::vector<char> bad = bytes;
                if ( round % 3 == 0 ) bad.resize( ( next() >> 4 ) % bad.size() ); // truncated
                const size_t span = round % 2 ? std::min<size_t>( bad.size(), 120 ) : bad.size(); // the header, or anywhere
                for ( int k = 0; k < 6 && span; k++ )
                    bad[( next() >> 4 ) % span] = (char)( next() >> 24 );
                {
                    std::ofstream out( path, std::ios::binary );
                    out.write( bad.data(), (std::streamsize)bad.size() );
                }
                try {
                    if ( c.hdr ) {
                        IBL ibl( path );
                        CHECK( ibl.m_pixels.size() == 3u * (size_t)ibl.m_width * (size_t)ibl.m_height );
                    }
                    else {
                        std::unique_ptr<texture> tex( image_texture( path.c_str() ) );
                        if ( tex ) CHECK( tex->texels.size() == 3u * (size_t)tex->width * (size_t)tex->height );
                    }
                }
                catch ( const std::exception& ) {
                }
            }
            std::remove( path.c_str() );
        }
    }
    {
        // scene data files with damaged text: a scene or an exception, never a sanitizer report
        for ( int id : { 9, 17, 18, 21 } ) {
            std::ifstream f( assets + "/scenes/scene" + std::to_string( id ) + ".scene", std::ios::binary );
            const std::string text( ( std::istreambuf_iterator<char>( f ) ), std::istreambuf_iterator<char>() );
            CHECK( text.size() > 100 );
            const std::string path = "/tmp/mrt_host_sanitize_case.scene";
            uint32_t state         = 99u + (uint32_t)id;
            auto next              = [&state]() { return state = state * 1664525u + 1013904223u; };
            for ( int round = 0; round < 25; round++ ) {
                std::string bad = text;
                for ( int k = 0; k < 3; k++ ) {
                    const size_t at = ( next() >> 4 ) % bad.size();
                    switch ( ( next() >> 9 ) % 5u ) {
                    case 0: bad[at] = (char)( next() >> 24 ); break;                          // any byte
                    case 1: bad.erase( at, ( next() >> 8 ) % 40u ); break;                     // a cut
                    case 2: bad.insert( at, " -1e999 nan 4294967296 " ); break;                // odd numbers
                    case 3: bad.insert( at, "\nmaterial\nmesh\nlight\ntexture 99999 x\n" ); break; // statements without arguments
                    default: bad.resize( at ); break;                                          // a truncated file
                    }
                    if ( bad.empty() ) bad = "camera";
                }
                {
                    std::ofstream out( path, std::ios::binary );
                    out << bad;
                }
                try {
                    Scene* scene = loadSceneFile( path.c_str(), 64, 48, assets.c_str() );
                    if ( scene ) freeScene( scene );
                }
                catch ( const std::exception& ) {
                }
            }
            std::remove( path.c_str() );
        }
    }
    {
        // an empty scene: no tree, nothing to flatten but the tables
        Scene* scene = initScene();
        scene->CreateMaterial( MaterialModel::COOK_TORRANCE );
        KdTree* tree = initKdTree( scene, 2 );
        std::unique_ptr<FlatScene> flat( flatten( scene, tree ) );
        CHECK( flat->desc.n_nodes == 0 && flat->desc.n_objects == 0 );
        delete tree;
        freeScene( scene );
    }
    std::printf( "host_sanitize ok\n" );
    return 0;
}
