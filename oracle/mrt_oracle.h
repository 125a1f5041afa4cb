/*
 * TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement ("oracle") of the reference hot path: SAH kd-tree build, kd-tree traversal,
 * ray/primitive intersection, sampling, Cook-Torrance BSDF, lights, MIS and the path-tracing
 * loop.  Plain scalar C++ that follows the reference statement by statement; every function
 * cites the reference file:line it restates (paths relative to the reference's src/).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library,
 * and only as the checker.  The product (libmrt_hip.so, libmrt_host.so) never links, loads or
 * calls anything in this directory.
 *
 * Pinning: in the build container the restatement is checked against oracle/_ref/libmrt_ref.so
 * (the reference's own kdtree/shape/sampling/bsdf/camera translation units) and the outputs are
 * committed as tests/golden/ fixtures, together with src/expected.h's known-answer tables.
 * Material::eval/sample, the Light classes and Pathtracer::trace_ray live in translation units
 * that cannot be built here (they include <stb_image.h>, which the image lacks); their
 * restatement below is pinned only through the functions they call: for those three the
 * status is "parity unpinned".
 *
 * The scene description is the product's own flat format (include/mrt.h), so the oracle and the
 * HIP library are driven with byte-identical inputs.
 */
#ifndef MRT_ORACLE_H
#define MRT_ORACLE_H

#include "../include/mrt.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- Transform (defines.h:67-148).  ops: rows of 5 floats {kind, a, b, c, d};
 * kind 0 = scale(a,b,c), 1 = rotate(axis=(a,b,c), degrees=d), 2 = translate(a,b,c). */
void orc_transform( int n_ops, const float* ops, mrt_transform* out );

/* ---- initSphere (scene.cpp:33-44): centre and radius from a transform. */
void orc_sphere_from_transform( const mrt_transform* xf, float* center_radius4 );

/* ---- initKdTree (kdtree.cpp:54-464) on the object arrays of `desc` (tree fields ignored). */
typedef struct orc_tree orc_tree;
orc_tree* orc_build( const mrt_scene_desc* desc );
void orc_tree_free( orc_tree* );
void orc_tree_counts( const orc_tree*, int64_t* n_nodes, int64_t* n_refs, int32_t* depth );
/* Pre-order dump, one entry per node (same layout as the reference-side dump). */
void orc_tree_flatten( const orc_tree*,
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
/* The same tree in the ABI node format (children adjacent, pre-order pairs). */
void orc_tree_to_abi( const orc_tree*,
                      mrt_kdnode* nodes,
                      uint32_t* refs,
                      float* root_min3,
                      float* root_max3,
                      uint32_t* out_of_tree,
                      uint32_t* n_out_of_tree );

/* ---- intersectKdTree (kdtree.cpp:555-613) over the flattened tree in `desc`.
 * shadow != 0 -> Ray::shadow.  Outputs may be NULL.  stats: summed over the batch. */
void orc_intersect( const mrt_scene_desc* desc,
                    int64_t n,
                    const float* o3,
                    const float* d3,
                    const float* tmin,
                    const float* tmax,
                    int shadow,
                    int32_t* prim,
                    float* t,
                    float* u,
                    float* v,
                    float* pos3,
                    float* nrm3,
                    uint8_t* outside,
                    mrt_trace_stats* stats,
                    int n_threads );

/* Object::intersect of one object, no tree (triangle.cpp:4-84, sphere.cpp:52-180, plane.cpp:6-68). */
void orc_object_intersect( const mrt_scene_desc* desc,
                           int obj,
                           int64_t n,
                           const float* o3,
                           const float* d3,
                           const float* tmin,
                           const float* tmax,
                           int shadow,
                           uint8_t* hit,
                           float* t,
                           float* u,
                           float* v,
                           float* pos3,
                           float* nrm3,
                           uint8_t* outside );

/* intersectAabb (kdtree.cpp:162-180). */
void orc_intersect_aabb( int64_t n,
                         const float* o3,
                         const float* d3,
                         const float* tmin,
                         const float* tmax,
                         const float* bmin3,
                         const float* bmax3,
                         uint8_t* hit,
                         float* otmin,
                         float* otmax );

/* ---- shape sampling: Object::sample / Area / pdf of a light's shape
 * (triangle.cpp:86-104, sphere.cpp:182-236, Object.cpp:13-22). */
void orc_light_shape_sample( const mrt_scene_desc* desc,
                             int light,
                             int64_t n,
                             const float* ipos3,
                             const float* inrm3,
                             const uint8_t* ioutside,
                             const float* u2,
                             float* opos3,
                             float* onrm3,
                             float* opdf );
float orc_light_shape_area( const mrt_scene_desc* desc, int light );
void orc_light_shape_pdf( const mrt_scene_desc* desc,
                          int light,
                          int64_t n,
                          const float* ipos3,
                          const float* inrm3,
                          const float* wi3,
                          float* opdf );

/* ---- bsdf.cpp / Material.h inline helpers */
void orc_beckmann( int64_t n, const float* ndoth, const float* alpha, float* out );
void orc_fresnel( int64_t n, const float* ldoth, const float* ext, const float* in, float* out );
void orc_smith( int64_t n,
                const float* ldoth,
                const float* ldotn,
                const float* vdoth,
                const float* vdotn,
                const float* alpha,
                float* out );
void orc_bsdf( int64_t n, const float* in12, float* out3 );
void orc_schlick_f( int64_t n,
                    const float* base3,
                    const float* metal,
                    const float* ldoth,
                    float* f0_3,
                    float* diff3,
                    float* F3 );
/* Material::eval (Material.cpp:983-1013) for a material given inline. */
void orc_material_eval( const mrt_material* mat,
                        int64_t n,
                        const float* raydir3,
                        const float* nrm3,
                        const float* wi3,
                        float* bsdf3,
                        float* pdf );

/* ---- sampling/*.cpp, rng.h */
void orc_rng_u32( uint64_t seq, int64_t n, uint32_t* out );
void orc_rng_float( uint64_t seq, int64_t n, float* out );
void orc_rng_bounded( uint64_t seq, int64_t n, const uint32_t* b, uint32_t* out );
void orc_stratified_stream( int nx, int ny, int ndims, int seed, int npixels, int nget, float* out );
void orc_uniform_sample_triangle( int64_t n, const float* u2, float* out2 );
void orc_concentric_disk( int64_t n, const float* u2, float* out2 );
void orc_power_heuristic( int64_t n, const float* f, const float* g, float* out );
void orc_dist1d_discrete( const float* func, int n, int64_t m, const float* u, int32_t* off, float* pdf );
void orc_dist2d_pdf( const float* data, int nu, int nv, int64_t m, const float* p2, float* pdf );

/* ---- SimpleCamera (Camera.cpp:5-67) */
void orc_camera( const float* from3,
                 const float* at3,
                 const float* up3,
                 float fov,
                 float w,
                 float h,
                 float radius,
                 float focus,
                 mrt_camera* out );
void orc_camera_rays( const mrt_camera* cam, int64_t n, const float* st2, const float* lens2, float* o3, float* d3 );

/* ---- the counter-based sample streams shared with the HIP kernels (see orc_sampler.h) */
void orc_sample_streams( uint32_t seed,
                         uint32_t pixel,
                         uint32_t spp_x,
                         uint32_t spp_y,
                         uint32_t sample,
                         float* cam5,
                         float* light1d_8,
                         float* light2d_16 );

/* ---- CPURenderer::run per-sample body + Pathtracer::trace_ray (CPURenderer.cpp:46-62,
 * integrator.cpp:69-211).  Adds linear radiance sums into rgb_sum[h][w][3].
 * sampler_mode 0 = counter-based streams (what the HIP kernels use),
 *              1 = the reference's sequential StratifiedSampler, one clone for the whole image.
 * per_sample (optional, [rows][w][samples][3]) receives each sample's radiance. */
void orc_render( const mrt_scene_desc* desc,
                 const mrt_render_params* params,
                 int sampler_mode,
                 float* rgb_sum,
                 float* per_sample,
                 mrt_render_stats* stats,
                 mrt_trace_stats* tstats3, /* traversal counters by ray class: continuation, shadow, MIS */
                 uint64_t* rays3,          /* rays traced per class */
                 int n_threads );

#ifdef __cplusplus
}
#endif
#endif
