/*
 * mrt_host_c.h -- C view of the host library (libmrt_host.so).
 *
 * The host library is the C++ API of mrt_host.hpp (the reference's Scene / Transform / Material /
 * initKdTree / ImageRenderer surface, see the citations there).  These functions expose the same
 * operations to callers that cannot include C++ headers; each one names the C++ call it forwards
 * to.  Handles are opaque pointers owned by the caller and released with the matching *_free.
 * Functions returning int give -1 on failure and set mrth_last_error().
 */
#ifndef MRT_HOST_C_H
#define MRT_HOST_C_H

#include "mrt.h"

#ifdef __cplusplus
extern "C" {
#endif

const char* mrth_last_error( void );

/* Transform (defines.h:67-148) */
void* mrth_transform_new( void );
void mrth_transform_free( void* xf );
void mrth_transform_scale( void* xf, float sx, float sy, float sz );
void mrth_transform_rotate( void* xf, float ax, float ay, float az, float degrees );
void mrth_transform_translate( void* xf, float x, float y, float z );
void mrth_transform_get( void* xf, mrt_transform* out );

/* initScene / freeScene / Scene::CreateMaterial (scene.h:33-37, scene.cpp:130-139) */
void* mrth_scene_new( void );
void mrth_scene_free( void* scene );
/* parseScene(sceneId) (example_scenes.cpp:1594-1647) over the scene data files of asset_dir/scenes/
 * (null asset_dir: $MRT_ASSETS, else "../assets"); loadSceneFile for one file.  Null on failure. */
void* mrth_parse_scene( int scene_id, int width, int height, const char* asset_dir );
void* mrth_load_scene_file( const char* filename, int width, int height, const char* asset_dir );
int mrth_create_material( void* scene, int mat_type );
int mrth_material_set( void* scene, int id, const mrt_material* m ); /* texture field ignored */
int mrth_material_get( void* scene, int id, mrt_material* m );
int mrth_material_set_checker( void* scene, int id, const float* even3, const float* odd3, void* tex_xf );
int mrth_material_set_image( void* scene, int id, uint32_t w, uint32_t h, const float* rgb, void* tex_xf );
/* materials first_id..last_id use the texture object of material from_id (one copy of the texels) */
int mrth_materials_share_texture( void* scene, int from_id, int first_id, int last_id );
/* scene->objects into a pseudo-random order (measurements: a mesh without locality in its file order) */
void mrth_shuffle_objects( void* scene, uint64_t seed );
int mrth_n_objects( void* scene );
int mrth_n_materials( void* scene );
int mrth_n_lights( void* scene );

/* addObjectsFromFile (example_scenes.cpp:22-153) with optional vertex pre-scale and 1->4 subdivision */
int mrth_add_obj_file( void* scene, const char* path, int default_mat, void* xf, float prescale, int subdivisions );
/* addObject(scene, initSmoothTriangle / initSphere / new Plane) (scene.cpp:33-110) */
int mrth_add_triangle( void* scene, const float* p9, const float* n9, const float* fn3, const float* uv6, void* xf, int mat );
int mrth_add_sphere( void* scene, void* xf, int mat );
int mrth_add_plane( void* scene, void* xf, int mat );
/* addLight(scene, new ShapeLight(..., initSphere / initSmoothTriangle)) (example_scenes.cpp:1501-1503, 119-130) */
int mrth_add_sphere_light( void* scene, void* xf, int mat, const float* color3 );
int mrth_add_triangle_light( void* scene, const float* p9, const float* n9, const float* fn3, void* xf, int mat, const float* color3 );
/* new IBL(file); addLight; envLights.push_back (example_scenes.cpp:1384-1392) */
int mrth_add_ibl_file( void* scene, const char* path, int also_env );
int mrth_add_ibl_pixels( void* scene, int w, int h, const float* rgb, int also_env );
/* setSimpleCamera (scene.cpp:151-161) */
void mrth_set_simple_camera( void* scene, const float* from3, const float* at3, const float* up3, float fov, float w, float h, float radius, float focus );

/* initKdTree (kdtree.h:35); n_threads 0 = all */
void* mrth_build_kdtree( void* scene, int n_threads );
void mrth_tree_free( void* tree );
void mrth_tree_counts( void* tree, int64_t* n_nodes, int64_t* n_refs, int32_t* depth );
/* pre-order dump of the pointer tree, one entry per node */
void mrth_tree_dump( void* tree,
                     uint8_t* leaf,
                     int32_t* axis,
                     float* split,
                     int32_t* left,
                     int32_t* right,
                     float* bmin,
                     float* bmax,
                     int32_t* ref_off,
                     int32_t* ref_cnt,
                     int32_t* refs );

/* flatten(scene, tree, layout): layout 0 = pre-order pairs, 1 = breadth-first top + pre-order */
void* mrth_flatten( void* scene, void* tree, int layout );
const mrt_scene_desc* mrth_flat_desc( void* flat );
uint32_t mrth_flat_depth( void* flat );
void mrth_flat_free( void* flat );

/* HIPRenderer: init, run, destroy (mrt_host.hpp) */
int mrth_intersect_kdtree( void* scene, void* tree, const float* o3, const float* d3, float tmin, float tmax, int shadow, float* out4 );
/* HIPRenderer::Render in steps of samples_per_step sample indices; with checkpoint_path the renderer
 * is checkpointed after checkpoint_after steps, destroyed and resumed from the file by a new one.
 * Returns the number of steps, -1 on failure. */
int mrth_hip_render_progressive( void* scene, uint32_t w, uint32_t h, int spp_x, int spp_y, int max_depth, uint32_t seed, int device, uint32_t samples_per_step,
                                 int checkpoint_after, const char* checkpoint_path, float* rgb_sum, mrt_render_stats* stats );
/* savePNG: linear or gamma-encoded floats [h][w][3] -> 8-bit RGB PNG as saveImage writes it (image.cpp:127-146). */
int mrth_save_png( const char* filename, const float* rgb, uint32_t w, uint32_t h );
/* image_texture(filename) (textures.hpp:50-62): a PNG (any colour type, bit depth, interlacing) or a
 * binary PPM; texels = byte / 255.f as loadPng makes them (image.cpp:30-34).  mrth_load_image copies
 * them to rgb when capacity_floats >= w * h * 3 (w, h are always written). */
int mrth_load_image( const char* filename, uint32_t* w, uint32_t* h, float* rgb, uint64_t capacity_floats );
int mrth_material_set_image_file( void* scene, int id, const char* filename, void* tex_xf );
int mrth_hip_render( void* scene, uint32_t w, uint32_t h, int spp_x, int spp_y, int max_depth, uint32_t seed, int device, float* rgb_sum, mrt_render_stats* stats, double* seconds );
/* The same with HIPRenderer::devices: the samples are divided among the listed devices (a device
 * may be listed more than once: one handle, with its own workspace, per entry). */
int mrth_hip_render_devices( void* scene, uint32_t w, uint32_t h, int spp_x, int spp_y, int max_depth, uint32_t seed, const int* devices, int n_devices, float* rgb_sum, mrt_render_stats* stats, double* seconds );

#ifdef __cplusplus
}
#endif
#endif
