// See flatten_ref.h.  Field by field what mrt.h's mrt_scene_desc documents.
#include "flatten_ref.h"

#include "../Camera.h"
#include "../Object.h"
#include "../image.h"

#include <cstring>
#include <map>
#include <stdexcept>

namespace {

// image_texture keeps its RenderImage in a private member (textures.hpp:123-124) and offers no
// accessor, and scene 18 -- the one main.cpp renders -- attaches one (example_scenes.cpp:1244).  To
// leave the reference's headers untouched, the member is reached through an explicit template
// instantiation, which the language exempts from access checking ([temp.spec]/6); a maintainer who
// prefers a one-line accessor in image_texture can replace imageOf() by it.
template <class Tag, typename Tag::type Member>
struct PrivateMember {
    friend typename Tag::type memberPointer( Tag ) { return Member; }
};
struct ImageTextureImage {
    typedef RenderImage* image_texture::*type;
    friend type memberPointer( ImageTextureImage );
};
template struct PrivateMember<ImageTextureImage, &image_texture::m_image>;
const RenderImage* imageOf( const image_texture* t ) {
    return t->*memberPointer( ImageTextureImage() );
}

void put3( float* d, const vec3& v ) {
    d[0] = v.x, d[1] = v.y, d[2] = v.z;
}
void putM3( float* d, const glm::mat3& m ) { // glm is column-major: m[col][row] -> d[3 * col + row]
    for ( int c = 0; c < 3; c++ )
        for ( int r = 0; r < 3; r++ )
            d[3 * c + r] = m[c][r];
}
mrt_transform toAbi( const Transform& t ) {
    mrt_transform o;
    putM3( o.m, t.getTransform() );
    putM3( o.inv, t.getInvTransform() );
    put3( o.t, t.getTranslation() );
    return o;
}
bool sameTransform( const Transform& a, const Transform& b ) {
    mrt_transform x = toAbi( a ), y = toAbi( b );
    return std::memcmp( &x, &y, sizeof( x ) ) == 0;
}

// Children of an inner node occupy two adjacent slots (right = left + 1), parents come before
// their children: a pre-order walk that reserves both slots when it meets the parent.
void emitNode( const KdTreeNode* n, uint32_t at, uint32_t depth, MrtFlat& out ) {
    if ( depth > out.tree_depth ) out.tree_depth = depth;
    if ( n->leaf ) {
        out.nodes[at].w0 = (uint32_t)out.refs.size();
        out.nodes[at].w1 = 3u | ( (uint32_t)n->objects.size() << 2 );
        for ( int o : n->objects ) // stored order decides ties (kdtree.cpp:515-546)
            out.refs.push_back( (uint32_t)o );
        return;
    }
    const uint32_t left = (uint32_t)out.nodes.size();
    out.nodes.resize( out.nodes.size() + 2 );
    std::memcpy( &out.nodes[at].w0, &n->split, 4 );
    out.nodes[at].w1 = (uint32_t)n->axis | ( left << 2 );
    emitNode( n->left, left, depth + 1, out );
    emitNode( n->right, left + 1, depth + 1, out );
}

} // namespace

uint32_t MrtFlat::internTransform( const Transform& t ) {
    // meshes share one transform per addObjectsFromFile call: look at the most recent first
    for ( size_t i = transform_values.size(); i-- > 0; )
        if ( sameTransform( transform_values[i], t ) ) return (uint32_t)i;
    transform_values.push_back( t );
    transforms.push_back( toAbi( t ) );
    return (uint32_t)transform_values.size() - 1;
}

void MrtFlat::bind() {
    desc.struct_size   = sizeof( mrt_scene_desc );
    desc.n_objects     = (uint32_t)obj_kind.size();
    desc.obj_kind      = obj_kind.data();
    desc.obj_material  = obj_material.data();
    desc.obj_transform = obj_transform.data();
    desc.tri_p         = tri_p.data();
    desc.tri_n         = tri_n.data();
    desc.tri_uv        = tri_uv.data();
    desc.sphere        = sphere.data();
    desc.n_transforms  = (uint32_t)transforms.size();
    desc.transforms    = transforms.data();
    desc.n_materials   = (uint32_t)materials.size();
    desc.materials     = materials.data();
    desc.n_textures    = (uint32_t)textures.size();
    desc.textures      = textures.data();
    desc.n_texels      = texels.size();
    desc.texels        = texels.data();
    desc.n_nodes       = (uint32_t)nodes.size();
    desc.nodes         = nodes.data();
    desc.n_refs        = (uint32_t)refs.size();
    desc.refs          = refs.data();
    desc.n_out_of_tree = (uint32_t)out_of_tree.size();
    desc.out_of_tree   = out_of_tree.data();
    desc.n_lights      = (uint32_t)lights.size();
    desc.lights        = lights.data();
    desc.env_rgb       = env_rgb.empty() ? nullptr : env_rgb.data();
}

void mrtFlattenGeometry( Scene* scene, KdTree* tree, MrtFlat& out ) {
    out            = MrtFlat {};
    const size_t n = scene->objects.size();
    out.obj_kind.resize( n );
    out.obj_material.resize( n );
    out.obj_transform.resize( n );
    out.tri_p.assign( n * 9, 0.f );
    out.tri_n.assign( n * 9, 0.f );
    out.tri_uv.assign( n * 6, 0.f );
    out.sphere.assign( n * 4, 0.f );
    out.internTransform( Transform() ); // index 0 must be the identity

    for ( size_t i = 0; i < n; i++ ) {
        const Object* o      = scene->objects[i];
        out.obj_kind[i]      = (uint8_t)o->geom.type; // Etype values are the MRT_* object kinds
        out.obj_material[i]  = (uint32_t)o->m_MaterialIndex;
        out.obj_transform[i] = out.internTransform( o->transform );
        if ( o->m_MaterialIndex >= scene->m_Materials.size() ) throw std::runtime_error( "flatten: material index out of range" );
        if ( o->geom.type == TRIANGLE ) {
            const auto& t = o->geom.triangle;
            put3( &out.tri_p[9 * i], t.p1 ), put3( &out.tri_p[9 * i + 3], t.p2 ), put3( &out.tri_p[9 * i + 6], t.p3 );
            put3( &out.tri_n[9 * i], t.n1 ), put3( &out.tri_n[9 * i + 3], t.n2 ), put3( &out.tri_n[9 * i + 6], t.n3 );
            for ( int k = 0; k < 3; k++ ) {
                out.tri_uv[6 * i + 2 * k]     = t.tex[k].x;
                out.tri_uv[6 * i + 2 * k + 1] = t.tex[k].y;
            }
        }
        else if ( o->geom.type == SPHERE ) {
            put3( &out.sphere[4 * i], o->geom.sphere.center );
            out.sphere[4 * i + 3] = o->geom.sphere.radius;
        }
        else if ( o->geom.type != PLANE ) { throw std::runtime_error( "flatten: cube objects are not supported by the GPU backend" ); }
    }

    std::map<const texture*, int32_t> texIndex;
    for ( const Material& m : scene->m_Materials ) {
        mrt_material o {};
        put3( o.albedo, m.m_albedo );
        put3( o.emission, m.m_emission );
        o.ior       = m.m_IOR;
        o.roughness = m.m_roughness;
        o.metalness = m.m_metalness;
        o.type      = (int32_t)m.m_MatType; // DIFFUSE / SPECULAR / TRANSPARENT = MRT_MAT_*
        o.texture   = -1;
        if ( m.m_texture && m.m_MatModel == MaterialModel::COOK_TORRANCE ) {
            auto it = texIndex.find( m.m_texture );
            if ( it == texIndex.end() ) {
                mrt_texture t {};
                t.texel_offset = out.texels.size(); // where an image's texels start; unused by a checker
                if ( auto* chk = dynamic_cast<const checker_texture*>( m.m_texture ) ) {
                    // the two solid colours, read through the public interface
                    t.kind = MRT_TEX_CHECKER;
                    put3( t.even, chk->even->value( 0.f, 0.f ) );
                    put3( t.odd, chk->odd->value( 0.f, 0.f ) );
                }
                else if ( auto* img = dynamic_cast<const image_texture*>( m.m_texture ) ) {
                    const RenderImage* px = imageOf( img );
                    if ( !px ) throw std::runtime_error( "flatten: an image_texture whose file could not be loaded" ); // loadPng returned nullptr
                    t.kind         = MRT_TEX_IMAGE;
                    t.width        = (uint32_t)px->width;
                    t.height       = (uint32_t)px->height;
                    for ( size_t k = 0; k < px->width * px->height; k++ ) {
                        out.texels.push_back( px->data[k].x );
                        out.texels.push_back( px->data[k].y );
                        out.texels.push_back( px->data[k].z );
                    }
                }
                else { throw std::runtime_error( "flatten: only checker_texture and image_texture are supported by the GPU backend" ); }
                t.xf = toAbi( m.m_texture->m_transform );
                it   = texIndex.emplace( m.m_texture, (int32_t)out.textures.size() ).first;
                out.textures.push_back( t );
            }
            o.texture = it->second;
        }
        out.materials.push_back( o );
    }

    if ( tree->root ) {
        out.nodes.resize( 1 );
        emitNode( tree->root, 0, 0, out );
        put3( out.desc.root_min, tree->root->min ); // already padded by 1 (kdtree.cpp:143-146)
        put3( out.desc.root_max, tree->root->max );
    }
    for ( int o : tree->outOfTree )
        out.out_of_tree.push_back( (uint32_t)o );

    out.desc.env_light = -1;
    const glm::mat3 ident( 1.f );
    putM3( out.desc.env_inv, ident );
    if ( auto* cam = dynamic_cast<const SimpleCamera*>( scene->cam ) ) {
        out.desc.has_camera = 1;
        put3( out.desc.camera.pos, cam->pos );
        put3( out.desc.camera.near_top_left, cam->nearPlaneTopLeft );
        put3( out.desc.camera.dx_pixel, cam->dXPixel );
        put3( out.desc.camera.dy_pixel, cam->dYPixel );
        put3( out.desc.camera.x_dir, cam->xDir );
        put3( out.desc.camera.up, cam->up );
        out.desc.camera.lens_radius = cam->m_lensRadius;
    }
    else if ( scene->cam ) { throw std::runtime_error( "flatten: the GPU backend generates rays for SimpleCamera only" ); }
    out.bind();
}
