/*
 * mrt.h -- C-ABI of the MI355X (gfx950) path-tracing backend.
 *
 * The library (libmrt_hip.so) replaces ONE path of the Mafo369/Raytracer CPU renderer: what a
 * sample does after camera-ray generation -- kd-tree traversal, ray/primitive intersection and
 * the Cook-Torrance / light-sampling / MIS shading loop.  The reference reaches that path through
 * per-ray C++ calls; a GPU needs batches, so every entry point below takes a whole batch and the
 * comment on it names the reference interface it stands in for (paths relative to the
 * reference's src/ directory).
 *
 * Conventions
 *   - plain C types only; every function returns mrt_status (0 = ok) and never throws;
 *     mrt_last_error() returns a per-thread message for the last failing call;
 *   - all arrays in mrt_scene_desc are HOST memory, read during mrt_scene_create and not
 *     retained (the library keeps its own device copy in its own layout);
 *   - ray / hit / image buffers may be host or device memory, selected by MRT_BUF_DEVICE;
 *   - one mrt_scene belongs to one HIP device; calls on one handle must be made one after another
 *     (not from two threads at once); the device work of successive calls on one handle is
 *     executed in call order even when the calls name different streams (the handle's scratch is
 *     shared, so the library orders them with an event); calls on different handles may run
 *     concurrently;
 *   - all arithmetic is IEEE fp32, unfused, in the reference's evaluation order, so that hit
 *     primitive ids are bit-identical to the reference's intersectKdTree.
 */
#ifndef MRT_H
#define MRT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t mrt_status;
enum {
    MRT_OK              = 0,
    MRT_ERR_INVALID     = 1, /* bad argument / inconsistent scene description */
    MRT_ERR_DEVICE      = 2, /* HIP runtime error (no device, launch failure ...) */
    MRT_ERR_NOMEM       = 3, /* device or host allocation failed */
    MRT_ERR_UNSUPPORTED = 4
};

/* Object kinds: the values of the reference's Etype (geometry.h:5). */
enum { MRT_SPHERE = 1, MRT_PLANE = 2, MRT_TRIANGLE = 3 };

/* MatType (Material.h:14). */
enum { MRT_MAT_DIFFUSE = 0, MRT_MAT_SPECULAR = 1, MRT_MAT_TRANSPARENT = 2 };

/* Transform (defines.h:67-148): m_transform, m_invTransform as glm column-major mat3
 * (element [3*col+row]) and m_translation. */
typedef struct mrt_transform {
    float m[9];
    float inv[9];
    float t[3];
} mrt_transform;

/* The Cook-Torrance fields of Material (Material.h:102-140) read by eval()/sample(). */
typedef struct mrt_material {
    float albedo[3];
    float emission[3];
    float ior;
    float roughness;
    float metalness;
    int32_t type;    /* MRT_MAT_* */
    int32_t texture; /* index into textures, -1 = none (m_texture == nullptr) */
} mrt_material;

/* texture (textures.hpp:13-125): image_texture or checker_texture of two solid colours. */
enum { MRT_TEX_IMAGE = 0, MRT_TEX_CHECKER = 1 };
typedef struct mrt_texture {
    int32_t kind;
    uint32_t width, height; /* image: texel grid */
    uint64_t texel_offset;  /* image: first float of this image in mrt_scene_desc.texels (RGB) */
    float even[3], odd[3];  /* checker colours */
    mrt_transform xf;       /* texture::m_transform */
} mrt_texture;

/* One flattened kd-tree node, 8 bytes (replaces KdTreeNode, kdtree.h:11-21).
 *   inner: w0 = bit pattern of the float split position
 *          w1 = axis (bits 0-1, 0..2) | index of the LEFT child << 2 ; right child = left + 1
 *   leaf : w0 = offset of the first reference in mrt_scene_desc.refs
 *          w1 = 3 | number of references << 2
 * Node 0 is the root.  Leaf references keep the order of KdTreeNode::objects. */
typedef struct mrt_kdnode {
    uint32_t w0;
    uint32_t w1;
} mrt_kdnode;

/* Light (Light.h:167-254).  A ShapeLight owns a private copy of its emitter shape
 * (example_scenes.cpp:119-129, 1501-1502), described inline here. */
enum { MRT_LIGHT_SHAPE = 0, MRT_LIGHT_IBL = 1 };
typedef struct mrt_light {
    int32_t kind;       /* MRT_LIGHT_* */
    float color[3];     /* Light::m_color */
    int32_t shape_kind; /* MRT_TRIANGLE or MRT_SPHERE (shape lights) */
    uint32_t transform; /* index into transforms */
    float tri_p[9];     /* geom.triangle.p1..p3 */
    float tri_n[9];     /* geom.triangle.n1..n3 */
    float sphere[4];    /* geom.sphere.center, geom.sphere.radius */
} mrt_light;

/* SimpleCamera (Camera.h:24-49) after its constructor ran. */
typedef struct mrt_camera {
    float pos[3];
    float near_top_left[3];
    float dx_pixel[3];
    float dy_pixel[3];
    float x_dir[3];
    float up[3];
    float lens_radius;
} mrt_camera;

/* Flattened Scene + KdTree (scene.h:19-43, kdtree.h:23-32).  Object i of the reference's
 * Scene::objects is entry i of every obj_* array; primitive ids reported by the library are
 * these indices. */
typedef struct mrt_scene_desc {
    uint32_t struct_size; /* sizeof(mrt_scene_desc), for versioning */

    uint32_t n_objects;
    const uint8_t* obj_kind;       /* [n] MRT_SPHERE / MRT_PLANE / MRT_TRIANGLE */
    const uint32_t* obj_material;  /* [n] Object::m_MaterialIndex */
    const uint32_t* obj_transform; /* [n] index into transforms */
    const float* tri_p;            /* [n][9] p1 p2 p3 (object space); unused rows ignored */
    const float* tri_n;            /* [n][9] n1 n2 n3 */
    const float* tri_uv;           /* [n][6] tex[0] tex[1] tex[2] */
    const float* sphere;           /* [n][4] geom.sphere.center, radius */

    uint32_t n_transforms;
    const mrt_transform* transforms;

    uint32_t n_materials;
    const mrt_material* materials;

    uint32_t n_textures;
    const mrt_texture* textures;
    uint64_t n_texels; /* floats */
    const float* texels;

    /* KdTree */
    uint32_t n_nodes; /* 0 when the scene has no in-tree object */
    const mrt_kdnode* nodes;
    uint32_t n_refs;
    const uint32_t* refs;
    float root_min[3], root_max[3]; /* root->min / root->max (already padded by 1) */
    uint32_t n_out_of_tree;
    const uint32_t* out_of_tree; /* KdTree::outOfTree (planes) */

    /* Scene::lights, Scene::envLights */
    uint32_t n_lights;
    const mrt_light* lights;
    int32_t env_light; /* index in lights of the IBL that is also in envLights, -1 = none */
    uint32_t env_width, env_height;
    const float* env_rgb; /* [h][w][3] IBL::m_pixels */
    float env_inv[9];     /* IBL::m_transform.getInvTransform() */

    mrt_camera camera;
    uint32_t has_camera;
} mrt_scene_desc;

typedef struct mrt_scene mrt_scene; /* opaque, one per device */

/* Buffer placement flag for rays / hits / images. */
enum { MRT_BUF_HOST = 0, MRT_BUF_DEVICE = 1 };

/* Ray batch, struct of arrays.  tmin/tmax as in Ray (ray.h:5-30). */
typedef struct mrt_rays {
    const float* ox;
    const float* oy;
    const float* oz;
    const float* dx;
    const float* dy;
    const float* dz;
    const float* tmin;
    const float* tmax;
} mrt_rays;

/* Hit batch.  prim = object index or -1; t = hit distance (Ray::tmax after the call), or the
 * tmax the ray came with when nothing was hit; u,v = barycentric weights of p2 and p3 for
 * triangles (triangle.cpp:24-31), the unit-square coordinates of a plane hit (plane.cpp:25-26),
 * 0 for spheres and misses. */
typedef struct mrt_hits {
    int32_t* prim;
    float* t;
    float* u;
    float* v;
} mrt_hits;

enum {
    MRT_CLOSEST_HIT = 0, /* Ray::shadow == false */
    MRT_ANY_HIT     = 1  /* Ray::shadow == true: first non-emissive occluder (kdtree.cpp:532-538) */
};

/* Per-batch traversal statistics (filled when non-NULL): totals over the batch of inner nodes
 * visited, leaves visited, leaf references read and primitive tests executed. */
typedef struct mrt_trace_stats {
    uint64_t inner, leaves, refs, prim_tests;
} mrt_trace_stats;

typedef struct mrt_render_params {
    uint32_t width, height;
    uint32_t spp_x, spp_y;                 /* StratifiedSampler(spp_x, spp_y, true, 8) */
    uint32_t sample_begin, sample_end;     /* sample indices rendered by this call */
    uint32_t max_depth;                    /* Pathtracer(maxDepth, ...) */
    uint32_t seed;                         /* replaces time(NULL) of CPURenderer.cpp:42 */
    uint32_t row_begin, row_end;           /* scanlines rendered by this call (tile split) */
    uint32_t paths_per_pass;               /* 0 = library default; at most 2^30 (MRT_ERR_UNSUPPORTED beyond) */
    uint32_t flags;                        /* MRT_RENDER_* */
    /* tile_step > 1: only the 8x8-pixel tiles (tx, ty) with (tx + ty) % tile_step == tile_phase are
     * rendered (tiles counted from the image's top left corner) -- an interleaved share of the image for
     * one of tile_step GPUs, all samples of its pixels; the shares of the tile_step phases tile the frame
     * and every pixel's sum is the one a whole-frame call makes.  Needs width % (8 * tile_step) == 0 and
     * row_begin, row_end (or height) multiples of 8 (MRT_ERR_UNSUPPORTED otherwise).  0 or 1: all tiles. */
    uint32_t tile_step, tile_phase;
} mrt_render_params;

enum {
    MRT_RENDER_TRACE_STATS = 1, /* count the traversal work of the call into mrt_render_stats.trace (slower kernels) */
    MRT_RENDER_SERIAL      = 2, /* one kernel at a time on the caller's stream.  By default the three ray lists a
                                   bounce leaves (shadow, MIS, continuation) are traced side by side on the handle's
                                   internal streams, forked from and joined to the caller's stream with events; the
                                   per-class device times of mrt_render_stats then overlap.  Serial runs give the
                                   time of each kernel alone. */
    MRT_RENDER_NO_KIND_BUCKETS = 4 /* in a scene whose materials are of more than one kind (plain, textured,
                                   transparent) the shade kernel regroups the paths of a block by kind from the
                                   second bounce on; this flag turns that off (measurements) */
};

typedef struct mrt_render_stats {
    uint64_t rays_closest; /* intersectKdTree-equivalent closest-hit queries */
    uint64_t rays_any;     /* shadow queries */
    uint64_t paths;        /* camera samples */
    double extend_ms, shade_ms, connect_ms, generate_ms; /* device time per kernel class (generate_ms is 0: the first extend makes the camera rays) */
    /* with MRT_RENDER_TRACE_STATS: nodes, leaves, references and primitive tests the traversal
     * kernels actually visited for [0] continuation rays, [1] shadow rays, [2] MIS rays */
    mrt_trace_stats trace[3];
    uint64_t rays_mis; /* the part of rays_closest that were MIS rays (integrator.cpp:104-109) */
    double connect_mis_ms; /* the part of connect_ms spent on the MIS rays' kernel */
    uint64_t passes;       /* passes the call was cut into (every kernel class is launched max_depth + 1 times per pass) */
} mrt_render_stats;

/* initKdTree + the Scene the reference hands to its integrator (kdtree.h:35, scene.h:19-43),
 * uploaded to `device` in the library's own layout.  That layout keeps the objects in the order in
 * which the tree's leaves first reference them (so a mesh stored in any order traverses alike); the
 * order inside a leaf's list, which decides ties, is the caller's, and every object index that
 * leaves the library (mrt_hits.prim) is the caller's.  MRT_OBJECT_ORDER=input in the environment
 * keeps the caller's order (measurements only).
 * Size limits (MRT_ERR_UNSUPPORTED beyond): fewer than 2^26 objects and transforms, 2^28 tree nodes,
 * 2^30 leaf references, tree depth at most 126, and tree records plus leaf lists below 4 GB (about
 * 60 M nodes with 150 M references).
 * Every index the description carries is checked on the host before the device is touched (tree
 * children and leaf ranges, one parent per node, depth; leaf references; out-of-tree planes; object
 * material / transform / kind; material type and texture; texture kind and texel window; light
 * kind / shape / transform; env_light naming an IBL light): MRT_ERR_INVALID or MRT_ERR_UNSUPPORTED
 * with a message, MRT_ERR_NOMEM when the host runs out of memory for the tables.  A description that
 * is accepted cannot make a kernel read outside its tables, whatever its floats hold. */
mrt_status mrt_scene_create( const mrt_scene_desc* desc, int device, mrt_scene** out );
void mrt_scene_destroy( mrt_scene* scene );

/* Batched bool intersectKdTree(Scene*, KdTree*, Ray*, Intersection*) (kdtree.h:34,
 * kdtree.cpp:555-613).  flags: MRT_CLOSEST_HIT or MRT_ANY_HIT; placement: MRT_BUF_*.
 * stream: a hipStream_t (NULL = the default stream); the call returns after the work has
 * completed when the buffers are host memory, and is asynchronous on `stream` when they are
 * device memory. */
mrt_status mrt_intersect( mrt_scene* scene,
                          uint64_t n,
                          const mrt_rays* rays,
                          const mrt_hits* hits,
                          uint32_t flags,
                          uint32_t placement,
                          void* stream,
                          mrt_trace_stats* stats );

/* mrt_intersect for callers that cannot allocate device memory (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.
 * Same results as mrt_intersect with MRT_BUF_DEVICE and stats == NULL: all eight ray arrays and
 * four hit arrays are device memory, the call is asynchronous on `stream`.  It calls no hipMalloc /
 * hipFree: the traversal's chunk-deal counters live in a static device array of
 * MRT_NOALLOC_SCRATCH_BYTES that every GPU the library runs on holds until the process ends
 * (its address on a device is looked up when the first mrt_scene is created there).  Every call on
 * one GPU, whatever the scene handle or stream, shares that GPU's array: calls that overlap in
 * time on one GPU race (rays are skipped or traced twice) -- order them on one stream or with
 * events.  Unlike the other calls it neither waits on nor records the handle's ordering event (it
 * shares no per-handle scratch): it is ordered by `stream` alone, also against other calls on the
 * same handle.  n > MRT_NOALLOC_MAX_RAYS is rejected with MRT_ERR_UNSUPPORTED. */
#define MRT_NOALLOC_MAX_RAYS 0x7fffffffull
#define MRT_NOALLOC_SCRATCH_BYTES 1088u
mrt_status mrt_intersect_noalloc( mrt_scene* scene,
                                  uint64_t n,
                                  const mrt_rays* rays,
                                  const mrt_hits* hits,
                                  uint32_t flags,
                                  void* stream );

/* The per-sample body of CPURenderer::run (rendering/CPURenderer.cpp:46-62): camera sample,
 * SimpleCamera::get_ray, Pathtracer::trace_ray (integrator.cpp:128-211), summed per pixel.
 * rgb_sum is [height][width][3] floats of LINEAR radiance SUMS (not averaged, not gamma
 * encoded); the call ADDS to it, so sample ranges and passes accumulate. */
mrt_status mrt_render( mrt_scene* scene,
                       const mrt_render_params* params,
                       float* rgb_sum,
                       uint32_t placement,
                       void* stream,
                       mrt_render_stats* stats );

/* Average + gamma of CPURenderer.cpp:64-69 on a sum buffer: out = powf(sum / spp, 1.0f / 2.2). */
mrt_status mrt_resolve( mrt_scene* scene,
                        uint64_t n_pixels,
                        const float* rgb_sum,
                        uint32_t spp,
                        float* rgb_out,
                        uint32_t placement,
                        void* stream );

/* Several GPUs of one node driven from ONE process (the reference renders in one process; nothing
 * in it corresponds to these two calls -- they are what its CPURenderer::run would call in place of
 * its OpenMP loop over scanlines, CPURenderer.cpp:40-44, when the node has more than one GPU).
 *
 * mrt_reduce: images[i] is a device-resident [n_floats] radiance-sum image on the device of
 * scenes[i]; all of them are added, in handle order, onto images[root] (peer copies over xGMI +
 * one add per source).  Waits for all work on the devices involved before and after.
 *
 * mrt_render_multi: mrt_render of one frame with the sample range [sample_begin, sample_end) of
 * `params` divided among the n handles (one per device, each created from the same description),
 * rendered side by side, reduced onto handle 0 and ADDED to the host image rgb_sum_host.  Sampling
 * is keyed by (seed, pixel, sample index), so the frame equals the single-handle frame up to the
 * order of the fp32 additions of the per-handle sums. */
mrt_status mrt_reduce( mrt_scene* const* scenes, float* const* images, int32_t n, uint64_t n_floats, int32_t root );
/* mrt_reduce without waiting on the host, for a loop that reduces every few samples (progressive
 * display): ordered after the calls made so far through the handles (mrt_render into images[i] on any
 * stream) by their per-handle events, executed on `stream` (a hipStream_t of the root's device, NULL =
 * its default stream), and later calls through any of the handles are ordered after it.  Images the
 * caller wrote by other means must be complete on its own account.  The root handle keeps a staging
 * image of n_floats until mrt_scene_trim or mrt_scene_destroy. */
mrt_status mrt_reduce_async( mrt_scene* const* scenes, float* const* images, int32_t n, uint64_t n_floats, int32_t root, void* stream );

/* Gives back the device memory a handle keeps between calls: the path-state workspace of mrt_render
 * (by default a pass takes what the frame needs, up to 2^30 paths and up to 82 % of the memory that is
 * free at the time -- and the workspace stays allocated for the next frame) and the staging image of
 * mrt_reduce.  Waits for the handle's work; the next call allocates again.  No counterpart in the
 * reference (its per-sample state lives on the C++ stack). */
mrt_status mrt_scene_trim( mrt_scene* scene );
mrt_status mrt_render_multi( mrt_scene* const* scenes, int32_t n, const mrt_render_params* params, float* rgb_sum_host, mrt_render_stats* stats );

/* Diagnostics: evaluates ONE device function of the shading code on `n` rows of host floats
 * (row strides in floats), so that each function can be compared with a CPU implementation in
 * isolation.  The reference functions each id corresponds to:
 *   MATERIAL_EVAL    Material::eval (Material.cpp:983-1013)     in: material, raydir3, normal3, wi3           out: bsdf3, pdf
 *   MATERIAL_SAMPLE  Material::sample (Material.cpp:1015-1100)  in: material, raydir3, normal3, outside, seq2  out: bsdf3, pdf, wi3
 *   LIGHT_SAMPLE     Object::sample(inter,u,&pdf) of a ShapeLight's shape (triangle.cpp:86-97, sphere.cpp:191-223)
 *                                                               in: light, pos3, normal3, outside, u2          out: pos3, normal3, pdf
 *   LIGHT_PDF        Object::pdf / Sphere::pdf (Object.cpp:13-22, sphere.cpp:225-236)
 *                                                               in: light, pos3, normal3, wi3                  out: pdf
 *   IBL              IBL::Le, the texel IBL::sample_Li returns, IBL::pdf_Li (Light.cpp:277-342)
 *                                                               in: dir3                                       out: Le3, texel3, pdf
 *   RANDOM_DIR       random_dir (sampling.h:199-211)            in: normal3, seq2                              out: dir3
 *   SAMPLER          the counter-based sample streams           in: seed, pixel, spp_x, spp_y, sample (uint bit patterns)
 *                                                               out: camera5, 1D dims 8, 2D dims 16, first 4 engine draws
 *   CAMERA           SimpleCamera::get_ray (Camera.cpp:37-67)   in: s, t, ulens, vlens                         out: origin3, dir3
 *   BSDF_TERMS       RDM_Beckmann, RDM_Fresnel, RDM_G1 (bsdf.cpp:43-58,100-118,141-154)
 *                                                               in: ndoth, alpha, ldoth, ext, int, ddotn       out: D, F, G1
 *   TEXTURE          texture::value(u, v, duv) (textures.hpp:21-43, 64-125) in: texture, u, v                       out: rgb3
 *   LIGHT_PICK       UniformLightDistribution::lookup()->SampleDiscrete (LightDistribution.cpp:5-12, sampling.h:86-119)
 *                                                               in: u                                          out: light index, pdf
 *   RANDOM_DIR_DRAWS random_dir on five given engine draws      in: normal3, draws5 (drawing order)            out: dir3
 * seq2 = the two 32-bit halves of a PCG32 sequence index, passed as float bit patterns. */
enum {
    MRT_DBG_MATERIAL_EVAL   = 0,
    MRT_DBG_MATERIAL_SAMPLE = 1,
    MRT_DBG_LIGHT_SAMPLE    = 2,
    MRT_DBG_LIGHT_PDF       = 3,
    MRT_DBG_IBL             = 4,
    MRT_DBG_RANDOM_DIR      = 5,
    MRT_DBG_SAMPLER         = 6,
    MRT_DBG_CAMERA          = 7,
    MRT_DBG_BSDF_TERMS      = 8,
    MRT_DBG_TEXTURE         = 9,
    MRT_DBG_LIGHT_PICK      = 10,
    MRT_DBG_RANDOM_DIR_DRAWS = 11
};
mrt_status mrt_debug_eval( mrt_scene* scene, uint32_t fn, uint32_t n, const float* in, uint32_t in_stride, float* out, uint32_t out_stride );

/* What the library found out about the device of `scene` (diagnostics): xcd_mask = the value for
 * which blockIdx.x & xcd_mask names the XCD a workgroup runs on (7 on an unpartitioned MI355X),
 * probed from the XCC_ID hardware register at mrt_scene_create. */
typedef struct mrt_device_info {
    int32_t device;
    int32_t compute_units;
    uint32_t xcd_mask;
    uint32_t kd_tree_depth;
} mrt_device_info;
mrt_status mrt_scene_device_info( const mrt_scene* scene, mrt_device_info* out );

/* Diagnostics: the number of rays this handle's traversal launches have handed over to follow-up
 * launches so far (waits for the device).  A traversal launch lasts as long as its longest ray, so a
 * ray still in flight some hundred trips after the launch's list ran out is handed over at the next
 * leaf boundary and the rest of its interval is traced as several shorter rays side by side; the
 * result is the same hit.  MRT_RELAY=0 in the environment when the scene is created switches this
 * off, MRT_RELAY_AFTER=<trips> sets how long a wave goes on first (default 256; 0 in the tests). */
mrt_status mrt_debug_relayed( mrt_scene* scene, uint64_t* out );

/* Diagnostics: how mrt_render traces the camera rays on this handle -- *packets = 1 when the 64 samples a
 * wave holds walk the tree together (wave-uniform nodes and primitives through scalar loads; faster
 * where a triangle covers many pixels), 0 when they go through the streaming traversal like every
 * other ray (faster where triangles are about a pixel wide).  Same hits either way, bit for bit.  Unless
 * MRT_CAMERA_PACKETS=0 / 1 in the environment fixed it when the scene was created, the first mrt_render
 * of an image size times both on three bands of scanlines (ms_stream, ms_packets: what it measured,
 * 0 before) and keeps the faster. */
mrt_status mrt_debug_camera_choice( const mrt_scene* scene, int32_t* packets, float* ms_stream, float* ms_packets );

const char* mrt_last_error( void );
/* The version of this header the library was built from (MRT_VERSION).  2: mrt_render_stats ends
 * with `passes` and mrt_render_params with tile_step, tile_phase (a caller built against version 1 must not
 * pass its shorter structs), mrt_scene_trim,
 * mrt_reduce_async, mrt_debug_relayed, mrt_debug_camera_choice. */
#define MRT_VERSION 2
int32_t mrt_version( void );

#ifdef __cplusplus
}
#endif
#endif /* MRT_H */
