// mrt_host.hpp -- C++ host side of the MI355X path-tracing backend.
//
// It mirrors the part of the Mafo369/Raytracer C++ API that sits directly above the hot path
// (same names, argument meaning and ownership; the citations are paths relative to the reference's
// src/), so that code written against the reference -- scene set-up, initKdTree, an ImageRenderer --
// can switch to the GPU backend:
//
//   Scene* scene = initScene();  ... CreateMaterial / addObjectsFromFile / setSimpleCamera ...
//   HIPRenderer r;  r.init(name, {w, h}, scene);  r.run();  r.destroy();
//
// The SAH kd-tree is still built on the CPU (initKdTree); flatten() turns Scene + KdTree into the
// mrt_scene_desc that the C-ABI (mrt.h) uploads.  Nothing here touches the GPU except HIPRenderer
// and intersectKdTree, which call the C-ABI.
#pragma once

#include "mrt.h"

#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace mrt {

struct vec2 {
    float x = 0, y = 0;
};
struct vec3 {
    float x = 0, y = 0, z = 0;
    vec3() {}
    vec3( float a, float b, float c ) : x( a ), y( b ), z( c ) {}
    explicit vec3( float a ) : x( a ), y( a ), z( a ) {}
    float operator[]( int i ) const { return i == 0 ? x : ( i == 1 ? y : z ); }
    float& operator[]( int i ) { return i == 0 ? x : ( i == 1 ? y : z ); }
};
typedef vec3 point3;
typedef vec2 point2;
typedef vec3 color3;
struct mat3 { // column-major, c[col][row]
    float c[3][3] = { { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 } };
};

// Transform (defines.h:67-148): same operations, same arithmetic.
class Transform
{
  public:
    mat3 const& getTransform() const { return m_transform; }
    vec3 const& getTranslation() const { return m_translation; }
    mat3 const& getInvTransform() const { return m_invTransform; }
    vec3 transformTo( vec3 const& p ) const;
    vec3 transformFrom( vec3 const& p ) const;
    vec3 vectorTransformFrom( vec3 const& dir ) const;
    void translate( vec3 const& p );
    void rotate( vec3 const& axis, float degrees );
    void scale( float sx, float sy, float sz );
    void transform( mat3 const& m );
    bool isIdentity() const;
    bool operator==( Transform const& o ) const;

  private:
    mat3 m_transform, m_invTransform;
    vec3 m_translation;
};

enum MatType { DIFFUSE = 0, SPECULAR = 1, TRANSPARENT = 2 }; // Material.h:14
enum class MaterialModel { COOK_TORRANCE = 0, BLINN = 1 };   // Material.h:100

// texture hierarchy of textures.hpp reduced to the two kinds the path tracer can reach.
struct texture {
    int kind = MRT_TEX_IMAGE;
    uint32_t width = 0, height = 0;
    std::vector<float> texels; // RGB, row-major, 8-bit / 255 (image.cpp:13-40)
    color3 even, odd;
    Transform m_transform;
};
texture* image_texture( const char* png_or_ppm_filename );
texture* checker_texture( color3 c1, color3 c2 );

// Material (Material.h:102-165), Cook-Torrance members.
class Material
{
  public:
    Material( size_t UID, MaterialModel matModel, MatType matType );
    MaterialModel m_MatModel;
    texture* m_texture = nullptr;
    color3 m_emission  = color3( 0, 0, 0 );
    color3 m_albedo    = color3( 0, 0, 0 );
    float m_IOR        = 1.f;
    float m_roughness  = 0.1f;
    float m_metalness  = 0.001f;
    MatType m_MatType  = DIFFUSE;
    size_t m_UID;
};

enum Etype { SPHERE = 1, PLANE, TRIANGLE, CUBE }; // geometry.h:5

struct Geometry { // geometry.h:7-42
    Etype type = TRIANGLE;
    struct {
        vec3 center;
        float radius = 0;
    } sphere;
    struct {
        vec3 p1, p2, p3, n1, n2, n3, normal;
        vec2 tex[3];
    } triangle;
};

// Object (Object.h:10-26).  Intersection and sampling run on the GPU; the host object only
// carries the data.
class Object
{
  public:
    Object( size_t materialIndex, Transform transform ) : transform( transform ), m_MaterialIndex( materialIndex ) {}
    Transform transform;
    Geometry geom;
    size_t m_MaterialIndex;
};
class Plane : public Object // shapes/plane.h
{
  public:
    Plane( size_t materialIndex, Transform transform ) : Object( materialIndex, transform ) { geom.type = PLANE; }
};

// Light (Light.h:9-44) -> the two kinds a Pathtracer scene uses.
class Light
{
  public:
    virtual ~Light() = default;
    color3 m_color;
};
class ShapeLight : public Light // Light.h:167-198; owns a private copy of its shape
{
  public:
    ShapeLight( vec3 position, color3 color, Object* obj ) : m_shape( obj ) { m_color = color; }
    ~ShapeLight() { delete m_shape; }
    Object* m_shape;
};
class IBL : public Light // Light.h:200-254; loads a Radiance .hdr file
{
  public:
    explicit IBL( const std::string& filename );
    IBL( int width, int height, std::vector<float> rgb ); // pixels supplied by the caller
    Transform m_transform;
    int m_width = 0, m_height = 0;
    std::vector<float> m_pixels;
};

class Camera // Camera.h:12-22
{
  public:
    virtual ~Camera() = default;
    float imgWidth = 0, imgHeight = 0;
};
class SimpleCamera : public Camera // Camera.h:24-49, Camera.cpp:5-35
{
  public:
    SimpleCamera( point3 lookfrom, point3 lookat, vec3 vup, float vfov, float aspect_ratio, int width, int height, float lensRadius, float focalDistance );
    vec3 pos, dir, up, left, nearPlaneTopLeft, dXPixel, dYPixel, xDir;
    float fov, aspect, l, h, w, m_lensRadius, m_focalDistance;
};

typedef std::vector<Object*> Objects;
typedef std::vector<Light*> Lights;

class Scene // scene.h:19-43
{
  public:
    ~Scene();
    Lights lights;
    Lights envLights;
    Objects objects;
    std::vector<Material> m_Materials;
    Camera* cam = nullptr;
    std::vector<texture*> owned_textures;
    Material& CreateMaterial( MaterialModel materialModel, MatType materialType = DIFFUSE ) {
        size_t id = m_Materials.size();
        return m_Materials.emplace_back( id, materialModel, materialType );
    }
    Material& GetMaterial( int index ) { return m_Materials[index]; }
};

// scene.h:45-118
Scene* initScene();
void freeScene( Scene* scene );
Object* initSphere( int materialIndex, Transform transform );
Object* initTriangle( vec3 p1, vec3 p2, vec3 p3, vec3 n, vec2 t[3], Transform transform, int materialIndex );
Object* initSmoothTriangle( vec3 p1, vec3 p2, vec3 p3, vec3 n, vec2 t[3], vec3 n1, vec3 n2, vec3 n3, Transform transform, int materialIndex );
void addObject( Scene* scene, Object* obj );
void addLight( Scene* scene, Light* light );
void setSimpleCamera( Scene* scene, point3 position, vec3 at, vec3 up, float fov, float w, float h, float radius, float distFocus );
// example_scenes.cpp:22-153.  default_mat is taken BY INDEX semantics: the reference keeps a
// reference into Scene::m_Materials across CreateMaterial calls; here the id is read up front.
void addObjectsFromFile( const char* filename, Scene* scene, Material& default_mat, Transform& transform );
// Same, with every vertex multiplied by `prescale` first and `subdivisions` rounds of 1 -> 4
// midpoint subdivision (for meshes whose triangles would fail the |det| < 1e-4 test, and for
// building large stress meshes from a small file).
void addObjectsFromFileEx( const char* filename, Scene* scene, size_t default_mat, Transform& transform, float prescale, int subdivisions );

// parseScene (example_scenes.cpp:1594-1647) over scene data files: the Cook-Torrance scenes 9, 16,
// 17, 18, 19, 20 and 21 live in <asset_dir>/scenes/scene<N>.scene (format: csrc/host/example_scenes.cpp)
// and list the reference's calls in the reference's order, so every index comes out the same; an id
// that names no scene gives scene 19 as in the reference.  Scenes 10-15 (Blinn materials, which the
// Pathtracer evaluates to black) throw std::invalid_argument.  width, height: the reference's WIDTH,
// HEIGHT macros (example_scenes.h:4-5).  asset_dir: null = $MRT_ASSETS, else "../assets".
Scene* parseScene( int sceneId, int width = 1200, int height = 800, const char* asset_dir = nullptr );
// One scene file; asset_dir null = the parent of the file's directory.
Scene* loadSceneFile( const char* filename, int width, int height, const char* asset_dir = nullptr );

// kdtree.h:11-35
struct KdTreeNode {
    bool leaf = false;
    int axis  = 0;
    float split = 0;
    int depth = 0;
    std::vector<int> objects;
    KdTreeNode* left  = nullptr;
    KdTreeNode* right = nullptr;
    vec3 min, max;
    ~KdTreeNode();
};
class KdTree
{
  public:
    ~KdTree();
    int depthLimit  = 80;
    size_t objLimit = 0;
    KdTreeNode* root = nullptr;
    std::vector<int> outOfTree;
    std::vector<int> inTree;
    mrt_scene* device_scene = nullptr; // created on first intersectKdTree / by Pathtracer::preprocess; freed with the tree
    int device_of_scene     = -1;      // the device device_scene lives on
    struct FlatScene* flat  = nullptr; // likewise
};
KdTree* initKdTree( Scene* scene );                 // kdtree.cpp:54-158, multi-threaded, same tree
KdTree* initKdTree( Scene* scene, int n_threads );

// Ray / Intersection as far as intersectKdTree's callers use them (ray.h:5-30, intersection.h:6-43).
class Ray
{
  public:
    Ray() {}
    Ray( const point3& o, const vec3& d, float tmin_ = 0, float tmax_ = 100000, int depth_ = 0 ) : orig( o ), dir( d ), tmax( tmax_ ), tmin( tmin_ ), depth( depth_ ) {}
    point3 orig;
    vec3 dir;
    float tmax = 100000, tmin = 0;
    int depth  = 0;
    bool shadow = false;
};
class Intersection
{
  public:
    point3 position;
    size_t materialIndex = 0;
    int prim             = -1;
    float u = 0, v = 0; // barycentric weights of p2, p3
    bool hit = false;
};
// kdtree.h:34.  One ray through the batched GPU entry point -- for callers that still trace ray
// by ray; anything performance-sensitive should batch through mrt_intersect / mrt_render.
bool intersectKdTree( Scene* scene, KdTree* tree, Ray* ray, Intersection* intersection );

// Scene + KdTree -> the flat description of mrt.h, with its storage.
struct FlatScene {
    mrt_scene_desc desc {};
    std::vector<uint8_t> obj_kind;
    std::vector<uint32_t> obj_material, obj_transform, refs, out_of_tree;
    std::vector<float> tri_p, tri_n, tri_uv, sphere, texels, env_rgb;
    std::vector<mrt_transform> transforms;
    std::vector<mrt_material> materials;
    std::vector<mrt_texture> textures;
    std::vector<mrt_kdnode> nodes;
    std::vector<mrt_light> lights;
    uint32_t tree_depth = 0;
};
enum NodeLayout { LAYOUT_PREORDER_PAIRS = 0, LAYOUT_BFS_TOP = 1 };
FlatScene* flatten( Scene* scene, KdTree* tree, NodeLayout layout = LAYOUT_BFS_TOP );

// Sampler configuration (sampling/sampler.h:30-75, sampling/stratified.h).  The sample values are
// generated on the GPU from (seed, pixel, sample index, dimension); the host object carries what
// CPURenderer::init puts into its sampler: the stratification grid and the number of
// pre-stratified dimensions.
class Sampler
{
  public:
    explicit Sampler( int64_t samplesPerPixel ) : samplesPerPixel( samplesPerPixel ) {}
    virtual ~Sampler() = default;
    const int64_t samplesPerPixel;
};
class StratifiedSampler : public Sampler
{
  public:
    StratifiedSampler( int xPixelSamples, int yPixelSamples, bool jitterSamples, int nSampledDimensions ) :
        Sampler( (int64_t)xPixelSamples * yPixelSamples ), xPixelSamples( xPixelSamples ), yPixelSamples( yPixelSamples ),
        jitterSamples( jitterSamples ), nSampledDimensions( nSampledDimensions ) {}
    const int xPixelSamples, yPixelSamples;
    const bool jitterSamples;
    const int nSampledDimensions;
};

// Camera samples an integrator is asked to trace: sample indices [sample_begin, sample_end) of every
// pixel of scanlines [row_begin, row_end) of a width x height image.
struct SampleRange {
    uint32_t width = 0, height = 0, sample_begin = 0, sample_end = 0, row_begin = 0, row_end = 0, seed = 1;
};

// Integrator / Pathtracer (integrator.h:25-43).  preprocess() is where the reference builds its
// UniformLightDistribution (integrator.cpp:124-126); here it also uploads scene and tree to the
// devices.  The per-ray trace_ray(Scene*, Ray&, KdTree*, Intersection*, Sampler*) cannot cross to
// a GPU one call at a time: trace_samples is its batched form -- the camera samples of `range`,
// traced to completion, their radiance ADDED per pixel to rgb_sum ([height][width][3], host).
class Integrator
{
  public:
    virtual ~Integrator() = default;
    virtual void preprocess( Scene* scene, KdTree* tree, Sampler* sampler, const std::vector<int>& devices ) = 0;
    virtual void trace_samples( const SampleRange& range, float* rgb_sum, mrt_render_stats* stats ) = 0;
};
class Pathtracer : public Integrator
{
  public:
    Pathtracer( int maxDepth, Sampler* sampler ) : m_maxDepth( maxDepth ), m_sampler( sampler ) {}
    ~Pathtracer() override;
    void preprocess( Scene* scene, KdTree* tree, Sampler* sampler, const std::vector<int>& devices ) override;
    void trace_samples( const SampleRange& range, float* rgb_sum, mrt_render_stats* stats ) override;
    int m_maxDepth;

  private:
    Sampler* m_sampler;
    KdTree* m_tree = nullptr;
    std::vector<mrt_scene*> m_devices; // one handle per device; [0] is the tree's own
    std::vector<mrt_scene*> m_owned;   // the handles this integrator created itself
};

// rendering/ImageRenderer.h:5-17
class ImageRenderer
{
  public:
    virtual ~ImageRenderer() = default;
    virtual void init( std::string& imageName, vec2 resolution, Scene* scene ) = 0;
    virtual void destroy()                                                      = 0;
    virtual void run()                                                          = 0;

  protected:
    Scene* m_scene          = nullptr;
    std::string m_imageName = "";
};

// Sibling of the reference's CPURenderer (rendering/CPURenderer.h): same init/run/destroy
// protocol; spp, depth, seed and device are parameters instead of literals
// (CPURenderer.cpp:15,18,42).
class HIPRenderer : public ImageRenderer
{
  public:
    int spp_x = 32, spp_y = 32, maxDepth = 5, device = 0;
    // more than one entry: the samples of every Render call are divided among these devices
    // (mrt_render_multi); empty = { device }
    std::vector<int> devices;
    uint32_t seed = 1;
    ~HIPRenderer() override { destroy(); }
    void init( std::string& imageName, vec2 resolution, Scene* scene ) override;
    void destroy() override;
    void run() override; // the whole spp_x * spp_y grid in one go
    const std::vector<float>& linearSum() const { return m_sum; } // [h][w][3] radiance sums
    const std::vector<float>& image() const { return m_image; }   // averaged + gamma 2.2
    mrt_render_stats stats {};
    double seconds = 0;

    // Progressive accumulation, the accumulate-and-display loop of the reference's viewer
    // (realtime/Renderer.cpp:57-110: m_accumulationData += frame; shown = accumulated /
    // m_frameIndex) over sample ranges of the one stratified grid: Render(n) adds the next n
    // sample indices to the sums and refreshes image() with the average of the samples so far.
    // Returns false once the grid is exhausted.  A checkpoint holds the sums and the position, so
    // a frame can be resumed by another renderer (same resolution, grid and seed) or process.
    bool Render( uint32_t n_samples );
    void resetAccumulation();
    uint32_t samplesDone() const { return m_next; }
    bool saveCheckpoint( const char* filename ) const;
    bool loadCheckpoint( const char* filename );

  private:
    KdTree* m_kdTree         = nullptr;
    Sampler* m_sampler       = nullptr; // CPURenderer.cpp:15
    Integrator* m_integrator = nullptr; // CPURenderer.cpp:18
    uint32_t m_w = 0, m_h = 0;
    uint32_t m_next = 0; // first sample index not rendered yet
    std::vector<float> m_sum, m_image;
};

bool savePPM( const char* filename, const float* rgb, uint32_t w, uint32_t h ); // 8-bit, clamp like image.cpp:137
// saveImage (image.cpp:60-146): the same 8-bit conversion, written as a PNG (8-bit RGB, stored
// deflate blocks; any PNG reader, lodepng included, decodes it to the same bytes)
bool savePNG( const char* filename, const float* rgb, uint32_t w, uint32_t h );

} // namespace mrt
