// Scene + KdTree -> the flat mrt_scene_desc handed to the C-ABI (see mrt.h, mrt_host.hpp).
#include "host_math.hpp"

#include <algorithm>
#include <cstring>
#include <deque>
#include <map>
#include <stdexcept>
#include <unordered_map>

namespace mrt {

namespace {

void put3( float* d, const vec3& v ) {
    d[0] = v.x;
    d[1] = v.y;
    d[2] = v.z;
}
void put_m3( float* d, const mat3& m ) {
    for ( int c = 0; c < 3; c++ )
        for ( int r = 0; r < 3; r++ )
            d[3 * c + r] = m.c[c][r];
}
mrt_transform to_abi( const Transform& t ) {
    mrt_transform o;
    put_m3( o.m, t.getTransform() );
    put_m3( o.inv, t.getInvTransform() );
    put3( o.t, t.getTranslation() );
    return o;
}

struct TransformTable {
    std::vector<Transform> uniq;
    std::vector<mrt_transform>* out;
    uint32_t intern( const Transform& t ) {
        // meshes share one transform per addObjectsFromFile call: check the most recent first
        for ( size_t i = uniq.size(); i-- > 0; )
            if ( uniq[i] == t ) return (uint32_t)i;
        uniq.push_back( t );
        out->push_back( to_abi( t ) );
        return (uint32_t)uniq.size() - 1;
    }
};

// Writes the tree into the node and reference arrays.  A subtree occupies one contiguous range of
// each (nodes in pre-order, children in adjacent slots; leaf lists in visiting order), so once the
// sizes of the subtrees are known, disjoint subtrees are written by different threads into ranges
// fixed beforehand: the arrays come out the same whatever the number of threads.
struct Emitter {
    std::vector<mrt_kdnode>& nodes;
    std::vector<uint32_t>& refs;
    uint32_t depth = 0;

    struct Size {
        uint64_t nodes = 0, refs = 0; // nodes excluding the subtree's root (whose slot the parent reserves)
    };
    // sizes of the subtrees of at least `MEMO` nodes, filled by measure_all
    static constexpr uint64_t MEMO = 512;
    std::unordered_map<const KdTreeNode*, Size> memo;
    Size measure_rec( const KdTreeNode* n, int spawn_levels, std::vector<std::pair<const KdTreeNode*, Size>>& found ) {
        Size s;
        if ( n->leaf ) {
            s.refs = n->objects.size();
            return s;
        }
        Size a, b;
        if ( spawn_levels > 0 ) {
            std::vector<std::pair<const KdTreeNode*, Size>> fa, fb;
#pragma omp task shared( a, fa ) firstprivate( n, spawn_levels )
            a = measure_rec( n->left, spawn_levels - 1, fa );
            b = measure_rec( n->right, spawn_levels - 1, fb );
#pragma omp taskwait
            found.insert( found.end(), fa.begin(), fa.end() );
            found.insert( found.end(), fb.begin(), fb.end() );
        }
        else {
            a = measure_rec( n->left, 0, found );
            b = measure_rec( n->right, 0, found );
        }
        s.nodes = 2 + a.nodes + b.nodes;
        s.refs  = a.refs + b.refs;
        if ( s.nodes >= MEMO ) found.emplace_back( n, s );
        return s;
    }
    Size measure_all( const KdTreeNode* root ) {
        std::vector<std::pair<const KdTreeNode*, Size>> found;
        Size s;
#pragma omp parallel
#pragma omp single
        s = measure_rec( root, 10, found );
        memo.reserve( found.size() * 2 );
        for ( const auto& f : found )
            memo.emplace( f.first, f.second );
        return s;
    }
    Size measure( const KdTreeNode* n ) const {
        auto it = memo.find( n );
        if ( it != memo.end() ) return it->second;
        Size s; // fewer than MEMO nodes: counted on the spot
        if ( n->leaf ) {
            s.refs = n->objects.size();
            return s;
        }
        Size a = measure( n->left ), b = measure( n->right );
        s.nodes = 2 + a.nodes + b.nodes;
        s.refs  = a.refs + b.refs;
        return s;
    }
    // the subtree of n: its root goes to slot `at`, its other nodes from `node_at`, its references from `ref_at`
    uint32_t write( const KdTreeNode* n, uint32_t at, uint64_t& node_at, uint64_t& ref_at, uint32_t d ) {
        uint32_t deepest = d;
        if ( n->leaf ) {
            nodes[at].w0 = (uint32_t)ref_at;
            nodes[at].w1 = 3u | ( (uint32_t)n->objects.size() << 2 );
            for ( int o : n->objects )
                refs[ref_at++] = (uint32_t)o;
            return deepest;
        }
        const uint32_t l = (uint32_t)node_at;
        node_at += 2;
        std::memcpy( &nodes[at].w0, &n->split, 4 );
        nodes[at].w1 = (uint32_t)n->axis | ( l << 2 );
        deepest      = std::max( deepest, write( n->left, l, node_at, ref_at, d + 1 ) );
        deepest      = std::max( deepest, write( n->right, l + 1, node_at, ref_at, d + 1 ) );
        return deepest;
    }
    struct Job {
        const KdTreeNode* n;
        uint32_t at, d;
        uint64_t node_at, ref_at;
    };
    // Splits the subtree of j into jobs of at most `grain` nodes (written later, in parallel); the
    // inner nodes above them are written here.
    void plan( const Job& j, uint64_t grain, std::vector<Job>& jobs ) {
        if ( j.n->leaf ) {
            jobs.push_back( j );
            return;
        }
        Size a = measure( j.n->left ), b = measure( j.n->right );
        if ( 2 + a.nodes + b.nodes <= grain ) {
            jobs.push_back( j );
            return;
        }
        if ( j.d > depth ) depth = j.d;
        const uint32_t l = (uint32_t)j.node_at;
        std::memcpy( &nodes[j.at].w0, &j.n->split, 4 );
        nodes[j.at].w1 = (uint32_t)j.n->axis | ( l << 2 );
        plan( Job { j.n->left, l, j.d + 1, j.node_at + 2, j.ref_at }, grain, jobs );
        plan( Job { j.n->right, l + 1, j.d + 1, j.node_at + 2 + a.nodes, j.ref_at + a.refs }, grain, jobs );
    }
    void run( std::vector<Job>& jobs ) {
        uint32_t deepest = depth;
#pragma omp parallel for schedule( dynamic, 1 ) reduction( max : deepest )
        for ( size_t i = 0; i < jobs.size(); i++ ) {
            Job j      = jobs[i];
            uint32_t v = write( j.n, j.at, j.node_at, j.ref_at, j.d );
            if ( v > deepest ) deepest = v;
        }
        depth = deepest;
    }
    void preorder( const KdTreeNode* root ) {
        Size s = measure_all( root );
        nodes.resize( 1 + s.nodes );
        refs.resize( s.refs );
        std::vector<Job> jobs;
        plan( Job { root, 0, 0, 1, 0 }, std::max<uint64_t>( 4096, s.nodes / 256 ), jobs );
        run( jobs );
    }
    // Breadth-first for the first `top` nodes, so the levels every ray walks through are one
    // contiguous prefix of the array; everything below in pre-order, so a subtree is contiguous.
    void bfs_top( const KdTreeNode* root, uint32_t top ) {
        struct Item {
            const KdTreeNode* n;
            uint32_t at, d;
        };
        Size total = measure_all( root );
        nodes.resize( 1 + total.nodes );
        refs.resize( total.refs );
        uint64_t node_at = 1, ref_at = 0;
        std::deque<Item> q;
        q.push_back( Item { root, 0, 0 } );
        while ( !q.empty() && node_at < top ) {
            Item it = q.front();
            q.pop_front();
            if ( it.d > depth ) depth = it.d;
            if ( it.n->leaf ) {
                nodes[it.at].w0 = (uint32_t)ref_at;
                nodes[it.at].w1 = 3u | ( (uint32_t)it.n->objects.size() << 2 );
                for ( int o : it.n->objects )
                    refs[ref_at++] = (uint32_t)o;
                continue;
            }
            const uint32_t l = (uint32_t)node_at;
            node_at += 2;
            std::memcpy( &nodes[it.at].w0, &it.n->split, 4 );
            nodes[it.at].w1 = (uint32_t)it.n->axis | ( l << 2 );
            q.push_back( Item { it.n->left, l, it.d + 1 } );
            q.push_back( Item { it.n->right, l + 1, it.d + 1 } );
        }
        std::vector<Job> jobs;
        const uint64_t grain = std::max<uint64_t>( 4096, total.nodes / 256 );
        for ( const Item& it : q ) {
            Size s = measure( it.n );
            plan( Job { it.n, it.at, it.d, node_at, ref_at }, grain, jobs );
            node_at += s.nodes;
            ref_at += s.refs;
        }
        run( jobs );
    }
};

} // namespace

FlatScene* flatten( Scene* scene, KdTree* tree, NodeLayout layout ) {
    std::unique_ptr<FlatScene> fs( new FlatScene );
    const size_t n = scene->objects.size();
    fs->obj_kind.resize( n );
    fs->obj_material.resize( n );
    fs->obj_transform.resize( n );
    fs->tri_p.assign( n * 9, 0.f );
    fs->tri_n.assign( n * 9, 0.f );
    fs->tri_uv.assign( n * 6, 0.f );
    fs->sphere.assign( n * 4, 0.f );

    TransformTable tt;
    tt.out = &fs->transforms;
    tt.intern( Transform() ); // index 0 is always the identity

    for ( size_t i = 0; i < n; i++ ) {
        const Object* o      = scene->objects[i];
        fs->obj_kind[i]      = (uint8_t)o->geom.type;
        fs->obj_material[i]  = (uint32_t)o->m_MaterialIndex;
        fs->obj_transform[i] = tt.intern( o->transform );
        if ( o->m_MaterialIndex >= scene->m_Materials.size() ) throw std::runtime_error( "flatten: material index out of range" );
        // the reference's path tracer evaluates a Blinn material to black (Material.cpp:1011, 1098) and leaves
        // the sampled direction unset; the kernels implement the Cook-Torrance model only
        if ( scene->m_Materials[o->m_MaterialIndex].m_MatModel != MaterialModel::COOK_TORRANCE )
            throw std::runtime_error( "flatten: object " + std::to_string( i ) + " uses a Blinn material, which the GPU backend does not shade" );
        if ( o->geom.type == TRIANGLE ) {
            const auto& t = o->geom.triangle;
            put3( &fs->tri_p[9 * i], t.p1 );
            put3( &fs->tri_p[9 * i + 3], t.p2 );
            put3( &fs->tri_p[9 * i + 6], t.p3 );
            put3( &fs->tri_n[9 * i], t.n1 );
            put3( &fs->tri_n[9 * i + 3], t.n2 );
            put3( &fs->tri_n[9 * i + 6], t.n3 );
            for ( int k = 0; k < 3; k++ ) {
                fs->tri_uv[6 * i + 2 * k]     = t.tex[k].x;
                fs->tri_uv[6 * i + 2 * k + 1] = t.tex[k].y;
            }
        }
        else if ( o->geom.type == SPHERE ) {
            put3( &fs->sphere[4 * i], o->geom.sphere.center );
            fs->sphere[4 * i + 3] = o->geom.sphere.radius;
        }
        else if ( o->geom.type != PLANE ) { throw std::runtime_error( "flatten: unsupported object kind" ); }
    }

    std::map<const texture*, int32_t> tex_index;
    for ( const Material& m : scene->m_Materials ) {
        mrt_material o {};
        put3( o.albedo, m.m_albedo );
        put3( o.emission, m.m_emission );
        o.ior       = m.m_IOR;
        o.roughness = m.m_roughness;
        o.metalness = m.m_metalness;
        o.type      = (int32_t)m.m_MatType;
        o.texture   = -1;
        if ( m.m_texture && m.m_MatModel == MaterialModel::COOK_TORRANCE ) {
            auto it = tex_index.find( m.m_texture );
            if ( it == tex_index.end() ) {
                mrt_texture t {};
                t.kind         = m.m_texture->kind;
                t.width        = m.m_texture->width;
                t.height       = m.m_texture->height;
                t.texel_offset = fs->texels.size();
                put3( t.even, m.m_texture->even );
                put3( t.odd, m.m_texture->odd );
                t.xf = to_abi( m.m_texture->m_transform );
                fs->texels.insert( fs->texels.end(), m.m_texture->texels.begin(), m.m_texture->texels.end() );
                it = tex_index.emplace( m.m_texture, (int32_t)fs->textures.size() ).first;
                fs->textures.push_back( t );
            }
            o.texture = it->second;
        }
        fs->materials.push_back( o );
    }

    int32_t env_light = -1;
    const IBL* env    = nullptr;
    for ( size_t i = 0; i < scene->lights.size(); i++ ) {
        mrt_light l {};
        put3( l.color, scene->lights[i]->m_color );
        if ( auto* sl = dynamic_cast<const ShapeLight*>( scene->lights[i] ) ) {
            const Object* o = sl->m_shape;
            l.kind          = MRT_LIGHT_SHAPE;
            l.shape_kind    = (int32_t)o->geom.type;
            l.transform     = tt.intern( o->transform );
            if ( o->geom.type == TRIANGLE ) {
                put3( l.tri_p, o->geom.triangle.p1 );
                put3( l.tri_p + 3, o->geom.triangle.p2 );
                put3( l.tri_p + 6, o->geom.triangle.p3 );
                put3( l.tri_n, o->geom.triangle.n1 );
                put3( l.tri_n + 3, o->geom.triangle.n2 );
                put3( l.tri_n + 6, o->geom.triangle.n3 );
            }
            else if ( o->geom.type == SPHERE ) {
                put3( l.sphere, o->geom.sphere.center );
                l.sphere[3] = o->geom.sphere.radius;
            }
            else { throw std::runtime_error( "flatten: a ShapeLight needs a triangle or a sphere" ); }
        }
        else if ( auto* ibl = dynamic_cast<const IBL*>( scene->lights[i] ) ) {
            l.kind = MRT_LIGHT_IBL;
            for ( const Light* e : scene->envLights )
                if ( e == ibl ) {
                    env_light = (int32_t)i;
                    env       = ibl;
                }
            if ( !env ) env = ibl;
        }
        else { throw std::runtime_error( "flatten: unsupported light type" ); }
        fs->lights.push_back( l );
    }
    if ( !env && !scene->envLights.empty() )
        throw std::runtime_error( "flatten: an environment light must also be in Scene::lights (as in every reference scene)" );

    Emitter em { fs->nodes, fs->refs };
    if ( tree->root ) {
        if ( layout == LAYOUT_PREORDER_PAIRS ) em.preorder( tree->root );
        else em.bfs_top( tree->root, 1u << 13 );
    }
    fs->tree_depth = em.depth;
    for ( int o : tree->outOfTree )
        fs->out_of_tree.push_back( (uint32_t)o );

    mrt_scene_desc& d = fs->desc;
    d.struct_size     = sizeof( mrt_scene_desc );
    d.n_objects       = (uint32_t)n;
    d.obj_kind        = fs->obj_kind.data();
    d.obj_material    = fs->obj_material.data();
    d.obj_transform   = fs->obj_transform.data();
    d.tri_p           = fs->tri_p.data();
    d.tri_n           = fs->tri_n.data();
    d.tri_uv          = fs->tri_uv.data();
    d.sphere          = fs->sphere.data();
    d.n_transforms    = (uint32_t)fs->transforms.size();
    d.transforms      = fs->transforms.data();
    d.n_materials     = (uint32_t)fs->materials.size();
    d.materials       = fs->materials.data();
    d.n_textures      = (uint32_t)fs->textures.size();
    d.textures        = fs->textures.data();
    d.n_texels        = fs->texels.size();
    d.texels          = fs->texels.data();
    d.n_nodes         = (uint32_t)fs->nodes.size();
    d.nodes           = fs->nodes.data();
    d.n_refs          = (uint32_t)fs->refs.size();
    d.refs            = fs->refs.data();
    if ( tree->root ) {
        put3( d.root_min, tree->root->min );
        put3( d.root_max, tree->root->max );
    }
    d.n_out_of_tree = (uint32_t)fs->out_of_tree.size();
    d.out_of_tree   = fs->out_of_tree.data();
    d.n_lights      = (uint32_t)fs->lights.size();
    d.lights        = fs->lights.data();
    d.env_light     = env_light;
    mat3 ident;
    put_m3( d.env_inv, ident );
    if ( env ) {
        fs->env_rgb  = env->m_pixels;
        d.env_width  = (uint32_t)env->m_width;
        d.env_height = (uint32_t)env->m_height;
        d.env_rgb    = fs->env_rgb.data();
        put_m3( d.env_inv, env->m_transform.getInvTransform() );
    }
    if ( auto* cam = dynamic_cast<const SimpleCamera*>( scene->cam ) ) {
        d.has_camera = 1;
        put3( d.camera.pos, cam->pos );
        put3( d.camera.near_top_left, cam->nearPlaneTopLeft );
        put3( d.camera.dx_pixel, cam->dXPixel );
        put3( d.camera.dy_pixel, cam->dYPixel );
        put3( d.camera.x_dir, cam->xDir );
        put3( d.camera.up, cam->up );
        d.camera.lens_radius = cam->m_lensRadius;
    }
    return fs.release();
}

} // namespace mrt
