// C entry points of the host library (libmrt_host.so) -- a thin shell over the C++ API of
// mrt_host.hpp for callers without a C++ toolchain (the Python tests and bench driver).
// Declared in include/mrt_host_c.h.
#include "host_math.hpp"
#include "mrt_host_c.h"

#include <cstring>
#include <exception>
#include <stdexcept>
#include <string>

using namespace mrt;

namespace {
thread_local std::string g_err;
template <typename F>
int guarded( F&& f ) {
    try {
        return f();
    }
    catch ( const std::exception& e ) {
        g_err = e.what();
        return -1;
    }
    catch ( ... ) {
        g_err = "unknown error";
        return -1;
    }
}
vec3 v3( const float* p ) {
    return vec3( p[0], p[1], p[2] );
}
void put3( float* d, const vec3& v ) {
    d[0] = v.x;
    d[1] = v.y;
    d[2] = v.z;
}
struct TreeDump {
    uint8_t* leaf;
    int32_t* axis;
    float* split;
    int32_t* left;
    int32_t* right;
    float* bmin;
    float* bmax;
    int32_t* ref_off;
    int32_t* ref_cnt;
    int32_t* refs;
    int64_t n_nodes = 0, n_refs = 0;
    int depth = 0;
};
int64_t walk( const KdTreeNode* node, TreeDump& c, int depth ) {
    int64_t id = c.n_nodes++;
    if ( depth > c.depth ) c.depth = depth;
    if ( c.leaf ) {
        c.leaf[id]  = node->leaf ? 1 : 0;
        c.axis[id]  = node->leaf ? -1 : node->axis;
        c.split[id] = node->leaf ? 0.f : node->split;
        put3( c.bmin + 3 * id, node->min );
        put3( c.bmax + 3 * id, node->max );
        c.ref_off[id] = (int32_t)c.n_refs;
        c.ref_cnt[id] = node->leaf ? (int32_t)node->objects.size() : 0;
        c.left[id] = c.right[id] = -1;
    }
    if ( node->leaf ) {
        if ( c.refs )
            for ( size_t i = 0; i < node->objects.size(); i++ )
                c.refs[c.n_refs + i] = node->objects[i];
        c.n_refs += node->objects.size();
        return id;
    }
    int64_t l = walk( node->left, c, depth + 1 );
    int64_t r = walk( node->right, c, depth + 1 );
    if ( c.leaf ) {
        c.left[id]  = (int32_t)l;
        c.right[id] = (int32_t)r;
    }
    return id;
}
} // namespace

extern "C" {

const char* mrth_last_error( void ) {
    return g_err.c_str();
}

void* mrth_transform_new( void ) {
    return new Transform();
}
void mrth_transform_free( void* xf ) {
    delete static_cast<Transform*>( xf );
}
void mrth_transform_scale( void* xf, float sx, float sy, float sz ) {
    static_cast<Transform*>( xf )->scale( sx, sy, sz );
}
void mrth_transform_rotate( void* xf, float ax, float ay, float az, float degrees ) {
    static_cast<Transform*>( xf )->rotate( vec3( ax, ay, az ), degrees );
}
void mrth_transform_translate( void* xf, float x, float y, float z ) {
    static_cast<Transform*>( xf )->translate( vec3( x, y, z ) );
}
void mrth_transform_get( void* xf, mrt_transform* out ) {
    auto* t = static_cast<Transform*>( xf );
    for ( int c = 0; c < 3; c++ )
        for ( int r = 0; r < 3; r++ ) {
            out->m[3 * c + r]   = t->getTransform().c[c][r];
            out->inv[3 * c + r] = t->getInvTransform().c[c][r];
        }
    put3( out->t, t->getTranslation() );
}

void* mrth_scene_new( void ) {
    return initScene();
}
void mrth_scene_free( void* scene ) {
    freeScene( static_cast<Scene*>( scene ) );
}
// parseScene( id ) / loadSceneFile( path ): null on failure (mrth_last_error says why)
void* mrth_parse_scene( int scene_id, int width, int height, const char* asset_dir ) {
    void* out = nullptr;
    guarded( [&] {
        out = parseScene( scene_id, width, height, asset_dir );
        return 0;
    } );
    return out;
}
void* mrth_load_scene_file( const char* filename, int width, int height, const char* asset_dir ) {
    void* out = nullptr;
    guarded( [&] {
        out = loadSceneFile( filename, width, height, asset_dir );
        return 0;
    } );
    return out;
}
int mrth_create_material( void* scene, int mat_type ) {
    return (int)static_cast<Scene*>( scene )->CreateMaterial( MaterialModel::COOK_TORRANCE, (MatType)mat_type ).m_UID;
}
int mrth_material_set( void* scene_, int id, const mrt_material* m ) {
    auto* scene = static_cast<Scene*>( scene_ );
    if ( id < 0 || (size_t)id >= scene->m_Materials.size() ) {
        g_err = "material id out of range";
        return -1;
    }
    Material& mat   = scene->GetMaterial( id );
    mat.m_albedo    = v3( m->albedo );
    mat.m_emission  = v3( m->emission );
    mat.m_IOR       = m->ior;
    mat.m_roughness = m->roughness;
    mat.m_metalness = m->metalness;
    mat.m_MatType   = (MatType)m->type;
    return 0;
}
int mrth_material_get( void* scene_, int id, mrt_material* m ) {
    auto* scene = static_cast<Scene*>( scene_ );
    if ( id < 0 || (size_t)id >= scene->m_Materials.size() ) {
        g_err = "material id out of range";
        return -1;
    }
    const Material& mat = scene->GetMaterial( id );
    put3( m->albedo, mat.m_albedo );
    put3( m->emission, mat.m_emission );
    m->ior       = mat.m_IOR;
    m->roughness = mat.m_roughness;
    m->metalness = mat.m_metalness;
    m->type      = (int32_t)mat.m_MatType;
    m->texture   = mat.m_texture ? 0 : -1;
    return 0;
}
int mrth_material_set_checker( void* scene_, int id, const float* even3, const float* odd3, void* tex_xf ) {
    auto* scene = static_cast<Scene*>( scene_ );
    texture* t  = checker_texture( v3( even3 ), v3( odd3 ) );
    if ( tex_xf ) t->m_transform = *static_cast<Transform*>( tex_xf );
    scene->owned_textures.push_back( t );
    scene->GetMaterial( id ).m_texture = t;
    return 0;
}
int mrth_material_set_image( void* scene_, int id, uint32_t w, uint32_t h, const float* rgb, void* tex_xf ) {
    auto* scene = static_cast<Scene*>( scene_ );
    texture* t  = new texture;
    t->kind     = MRT_TEX_IMAGE;
    t->width    = w;
    t->height   = h;
    t->texels.assign( rgb, rgb + (size_t)w * h * 3 );
    if ( tex_xf ) t->m_transform = *static_cast<Transform*>( tex_xf );
    scene->owned_textures.push_back( t );
    scene->GetMaterial( id ).m_texture = t;
    return 0;
}
int mrth_materials_share_texture( void* scene_, int from_id, int first_id, int last_id ) {
    auto* scene = static_cast<Scene*>( scene_ );
    const int n = (int)scene->m_Materials.size();
    if ( from_id < 0 || from_id >= n || first_id < 0 || last_id >= n || first_id > last_id ) {
        g_err = "material id out of range";
        return -1;
    }
    texture* t = scene->GetMaterial( from_id ).m_texture;
    for ( int id = first_id; id <= last_id; id++ )
        scene->GetMaterial( id ).m_texture = t;
    return 0;
}
// Puts scene->objects into a pseudo-random order (Fisher-Yates on a 64-bit LCG): a mesh whose file
// order carries no locality, for measurements.
void mrth_shuffle_objects( void* scene_, uint64_t seed ) {
    auto& objs     = static_cast<Scene*>( scene_ )->objects;
    uint64_t state = seed * 6364136223846793005ull + 1442695040888963407ull;
    for ( size_t i = objs.size(); i > 1; i-- ) {
        state    = state * 6364136223846793005ull + 1442695040888963407ull;
        size_t j = (size_t)( ( state >> 33 ) % i );
        std::swap( objs[i - 1], objs[j] );
    }
}
int mrth_n_objects( void* scene ) {
    return (int)static_cast<Scene*>( scene )->objects.size();
}
int mrth_n_materials( void* scene ) {
    return (int)static_cast<Scene*>( scene )->m_Materials.size();
}
int mrth_n_lights( void* scene ) {
    return (int)static_cast<Scene*>( scene )->lights.size();
}

int mrth_add_obj_file( void* scene_, const char* path, int default_mat, void* xf, float prescale, int subdivisions ) {
    return guarded( [&] {
        auto* scene   = static_cast<Scene*>( scene_ );
        size_t before = scene->objects.size();
        addObjectsFromFileEx( path, scene, (size_t)default_mat, *static_cast<Transform*>( xf ), prescale, subdivisions );
        return (int)( scene->objects.size() - before );
    } );
}
int mrth_add_triangle( void* scene_, const float* p9, const float* n9, const float* fn3, const float* uv6, void* xf, int mat ) {
    auto* scene = static_cast<Scene*>( scene_ );
    vec2 t[3]   = { vec2 { uv6[0], uv6[1] }, vec2 { uv6[2], uv6[3] }, vec2 { uv6[4], uv6[5] } };
    addObject( scene, initSmoothTriangle( v3( p9 ), v3( p9 + 3 ), v3( p9 + 6 ), v3( fn3 ), t, v3( n9 ), v3( n9 + 3 ), v3( n9 + 6 ), *static_cast<Transform*>( xf ), mat ) );
    return (int)scene->objects.size() - 1;
}
int mrth_add_sphere( void* scene_, void* xf, int mat ) {
    auto* scene = static_cast<Scene*>( scene_ );
    addObject( scene, initSphere( mat, *static_cast<Transform*>( xf ) ) );
    return (int)scene->objects.size() - 1;
}
int mrth_add_plane( void* scene_, void* xf, int mat ) {
    auto* scene = static_cast<Scene*>( scene_ );
    addObject( scene, new Plane( mat, *static_cast<Transform*>( xf ) ) );
    return (int)scene->objects.size() - 1;
}
int mrth_add_sphere_light( void* scene_, void* xf, int mat, const float* color3_ ) {
    auto* scene = static_cast<Scene*>( scene_ );
    addLight( scene, new ShapeLight( vec3( 0.f ), v3( color3_ ), initSphere( mat, *static_cast<Transform*>( xf ) ) ) );
    return (int)scene->lights.size() - 1;
}
int mrth_add_triangle_light( void* scene_, const float* p9, const float* n9, const float* fn3, void* xf, int mat, const float* color3_ ) {
    auto* scene = static_cast<Scene*>( scene_ );
    vec2 t[3]   = {};
    addLight( scene, new ShapeLight( vec3( 0.f ), v3( color3_ ), initSmoothTriangle( v3( p9 ), v3( p9 + 3 ), v3( p9 + 6 ), v3( fn3 ), t, v3( n9 ), v3( n9 + 3 ), v3( n9 + 6 ), *static_cast<Transform*>( xf ), mat ) ) );
    return (int)scene->lights.size() - 1;
}
int mrth_add_ibl_file( void* scene_, const char* path, int also_env ) {
    return guarded( [&] {
        auto* scene = static_cast<Scene*>( scene_ );
        IBL* sky    = new IBL( std::string( path ) );
        addLight( scene, sky );
        if ( also_env ) scene->envLights.push_back( sky );
        return (int)scene->lights.size() - 1;
    } );
}
int mrth_add_ibl_pixels( void* scene_, int w, int h, const float* rgb, int also_env ) {
    auto* scene = static_cast<Scene*>( scene_ );
    IBL* sky    = new IBL( w, h, std::vector<float>( rgb, rgb + (size_t)w * h * 3 ) );
    addLight( scene, sky );
    if ( also_env ) scene->envLights.push_back( sky );
    return (int)scene->lights.size() - 1;
}
void mrth_set_simple_camera( void* scene, const float* from3, const float* at3, const float* up3, float fov, float w, float h, float radius, float focus ) {
    setSimpleCamera( static_cast<Scene*>( scene ), v3( from3 ), v3( at3 ), v3( up3 ), fov, w, h, radius, focus );
}

void* mrth_build_kdtree( void* scene, int n_threads ) {
    void* out = nullptr;
    guarded( [&] {
        out = initKdTree( static_cast<Scene*>( scene ), n_threads );
        return 0;
    } );
    return out;
}
void mrth_tree_free( void* tree ) {
    delete static_cast<KdTree*>( tree );
}
void mrth_tree_counts( void* tree_, int64_t* n_nodes, int64_t* n_refs, int32_t* depth ) {
    auto* tree = static_cast<KdTree*>( tree_ );
    TreeDump c {};
    if ( tree->root ) walk( tree->root, c, 0 );
    *n_nodes = c.n_nodes;
    *n_refs  = c.n_refs;
    *depth   = c.depth;
}
void mrth_tree_dump( void* tree_,
                     uint8_t* leaf,
                     int32_t* axis,
                     float* split,
                     int32_t* left,
                     int32_t* right,
                     float* bmin,
                     float* bmax,
                     int32_t* ref_off,
                     int32_t* ref_cnt,
                     int32_t* refs ) {
    auto* tree = static_cast<KdTree*>( tree_ );
    TreeDump c { leaf, axis, split, left, right, bmin, bmax, ref_off, ref_cnt, refs };
    if ( tree->root ) walk( tree->root, c, 0 );
}

void* mrth_flatten( void* scene, void* tree, int layout ) {
    void* out = nullptr;
    guarded( [&] {
        out = flatten( static_cast<Scene*>( scene ), static_cast<KdTree*>( tree ), (NodeLayout)layout );
        return 0;
    } );
    return out;
}
const mrt_scene_desc* mrth_flat_desc( void* flat ) {
    return &static_cast<FlatScene*>( flat )->desc;
}
uint32_t mrth_flat_depth( void* flat ) {
    return static_cast<FlatScene*>( flat )->tree_depth;
}
void mrth_flat_free( void* flat ) {
    delete static_cast<FlatScene*>( flat );
}

// bool intersectKdTree(Scene*, KdTree*, Ray*, Intersection*) for one ray.  out4 = prim, t, u, v.
// Returns 1 on a hit, 0 on a miss, -1 on an error.
int mrth_intersect_kdtree( void* scene, void* tree, const float* o3, const float* d3, float tmin, float tmax, int shadow, float* out4 ) {
    int hit = 0;
    int rc  = guarded( [&] {
        Ray ray( point3( o3[0], o3[1], o3[2] ), vec3( d3[0], d3[1], d3[2] ), tmin, tmax );
        ray.shadow = shadow != 0;
        Intersection is;
        hit     = intersectKdTree( static_cast<Scene*>( scene ), static_cast<KdTree*>( tree ), &ray, &is ) ? 1 : 0;
        out4[0] = hit ? (float)is.prim : -1.f;
        out4[1] = ray.tmax;
        out4[2] = hit ? is.u : 0.f;
        out4[3] = hit ? is.v : 0.f;
        return 0;
    } );
    return rc < 0 ? -1 : hit;
}

// HIPRenderer::init + run + destroy; rgb_sum receives the linear radiance sums.
int mrth_save_png( const char* filename, const float* rgb, uint32_t w, uint32_t h ) {
    return savePNG( filename, rgb, w, h ) ? 0 : -1;
}

// image_texture( filename ): the texels (byte / 255.f, [h][w][3]) into rgb when it holds
// capacity_floats >= w * h * 3; w, h are always written.  -1 when the file cannot be decoded.
int mrth_load_image( const char* filename, uint32_t* w, uint32_t* h, float* rgb, uint64_t capacity_floats ) {
    return guarded( [&] {
        std::unique_ptr<texture> t( image_texture( filename ) );
        if ( !t ) {
            g_err = std::string( "cannot decode " ) + filename;
            return -1;
        }
        *w = t->width, *h = t->height;
        if ( rgb && capacity_floats >= t->texels.size() ) std::memcpy( rgb, t->texels.data(), t->texels.size() * sizeof( float ) );
        return 0;
    } );
}
int mrth_material_set_image_file( void* scene_, int id, const char* filename, void* tex_xf ) {
    auto* scene = static_cast<Scene*>( scene_ );
    if ( id < 0 || (size_t)id >= scene->m_Materials.size() ) {
        g_err = "material id out of range";
        return -1;
    }
    return guarded( [&] {
        std::unique_ptr<texture> t( image_texture( filename ) );
        if ( !t ) {
            g_err = std::string( "cannot decode " ) + filename;
            return -1;
        }
        if ( tex_xf ) t->m_transform = *static_cast<Transform*>( tex_xf );
        scene->owned_textures.push_back( t.get() );
        scene->GetMaterial( id ).m_texture = t.release();
        return 0;
    } );
}

int mrth_hip_render( void* scene, uint32_t w, uint32_t h, int spp_x, int spp_y, int max_depth, uint32_t seed, int device, float* rgb_sum, mrt_render_stats* stats, double* seconds ) {
    return mrth_hip_render_devices( scene, w, h, spp_x, spp_y, max_depth, seed, &device, 1, rgb_sum, stats, seconds );
}

int mrth_hip_render_devices( void* scene, uint32_t w, uint32_t h, int spp_x, int spp_y, int max_depth, uint32_t seed, const int* devices, int n_devices, float* rgb_sum, mrt_render_stats* stats, double* seconds ) {
    return guarded( [&] {
        if ( !devices || n_devices < 1 ) throw std::invalid_argument( "no device" );
        HIPRenderer r;
        r.spp_x    = spp_x;
        r.spp_y    = spp_y;
        r.maxDepth = max_depth;
        r.seed     = seed;
        r.device   = devices[0];
        r.devices.assign( devices, devices + n_devices );
        std::string name;
        r.init( name, vec2 { (float)w, (float)h }, static_cast<Scene*>( scene ) );
        r.run();
        std::memcpy( rgb_sum, r.linearSum().data(), r.linearSum().size() * sizeof( float ) );
        if ( stats ) *stats = r.stats;
        if ( seconds ) *seconds = r.seconds;
        r.destroy();
        return 0;
    } );
}

// The same frame rendered progressively, samples_per_step sample indices per HIPRenderer::Render
// call.  With checkpoint_path, the renderer is checkpointed after `checkpoint_after` steps,
// destroyed, and a new one resumes from the file.  Returns the number of Render calls made.
int mrth_hip_render_progressive( void* scene, uint32_t w, uint32_t h, int spp_x, int spp_y, int max_depth, uint32_t seed, int device, uint32_t samples_per_step,
                                 int checkpoint_after, const char* checkpoint_path, float* rgb_sum, mrt_render_stats* stats ) {
    return guarded( [&] {
        auto make = [&]( HIPRenderer& r ) {
            r.spp_x    = spp_x;
            r.spp_y    = spp_y;
            r.maxDepth = max_depth;
            r.seed     = seed;
            r.device   = device;
            std::string name;
            r.init( name, vec2 { (float)w, (float)h }, static_cast<Scene*>( scene ) );
        };
        std::unique_ptr<HIPRenderer> r( new HIPRenderer );
        make( *r );
        int steps = 0;
        mrt_render_stats total {};
        bool more = true;
        while ( more ) {
            more = r->Render( samples_per_step );
            steps++;
            if ( more && checkpoint_path && steps == checkpoint_after ) {
                if ( !r->saveCheckpoint( checkpoint_path ) ) throw std::runtime_error( "cannot write the checkpoint" );
                total = r->stats;
                r->destroy();
                r.reset( new HIPRenderer );
                make( *r );
                if ( !r->loadCheckpoint( checkpoint_path ) ) throw std::runtime_error( "cannot resume from the checkpoint" );
            }
        }
        std::memcpy( rgb_sum, r->linearSum().data(), r->linearSum().size() * sizeof( float ) );
        if ( stats ) {
            *stats = r->stats;
            stats->paths += total.paths, stats->rays_closest += total.rays_closest, stats->rays_any += total.rays_any, stats->rays_mis += total.rays_mis, stats->passes += total.passes;
        }
        r->destroy();
        return steps;
    } );
}

} // extern "C"
