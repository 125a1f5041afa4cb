// See flatten_ref.h.  This unit includes Light.h, which needs <stb_image.h> (vendored by the
// Walnut submodule in a full checkout of the reference).
//
// ShapeLight keeps its shape private (Light.h:196-197).  The one-line accessor this file relies on:
//     const Object* shape() const { return m_shape; }        // in class ShapeLight
#include "flatten_ref.h"

#include "../Light.h"
#include "../Object.h"

#include <stdexcept>

namespace {
void put3( float* d, const vec3& v ) {
    d[0] = v.x, d[1] = v.y, d[2] = v.z;
}
} // namespace

void mrtFlattenLights( Scene* scene, MrtFlat& out ) {
    out.lights.clear();
    out.env_rgb.clear();
    out.desc.env_light = -1;
    const IBL* env     = nullptr;
    for ( size_t i = 0; i < scene->lights.size(); i++ ) {
        mrt_light l {};
        put3( l.color, scene->lights[i]->m_color );
        if ( auto* sl = dynamic_cast<const ShapeLight*>( scene->lights[i] ) ) {
            const Object* o = sl->shape();
            l.kind          = MRT_LIGHT_SHAPE;
            l.shape_kind    = (int32_t)o->geom.type;
            l.transform     = out.internTransform( o->transform );
            if ( o->geom.type == TRIANGLE ) {
                put3( l.tri_p, o->geom.triangle.p1 ), put3( l.tri_p + 3, o->geom.triangle.p2 ), put3( l.tri_p + 6, o->geom.triangle.p3 );
                put3( l.tri_n, o->geom.triangle.n1 ), put3( l.tri_n + 3, o->geom.triangle.n2 ), put3( l.tri_n + 6, o->geom.triangle.n3 );
            }
            else if ( o->geom.type == SPHERE ) {
                put3( l.sphere, o->geom.sphere.center );
                l.sphere[3] = o->geom.sphere.radius;
            }
            else { throw std::runtime_error( "flatten: a ShapeLight needs a triangle or a sphere" ); }
        }
        else if ( auto* ibl = dynamic_cast<const IBL*>( scene->lights[i] ) ) {
            l.kind = MRT_LIGHT_IBL;
            for ( const Light* e : scene->envLights ) // the same IBL sits in both lists (example_scenes.cpp:1183-1184)
                if ( e == ibl ) out.desc.env_light = (int32_t)i;
            if ( !env || out.desc.env_light == (int32_t)i ) env = ibl;
        }
        else { throw std::runtime_error( "flatten: the Pathtracer reaches ShapeLight and IBL only" ); }
        out.lights.push_back( l );
    }
    if ( !env && !scene->envLights.empty() ) throw std::runtime_error( "flatten: an environment light must also be in Scene::lights" );
    if ( env ) {
        out.env_rgb.assign( env->m_pixels, env->m_pixels + 3 * (size_t)env->m_width * env->m_height );
        out.desc.env_width  = (uint32_t)env->m_width;
        out.desc.env_height = (uint32_t)env->m_height;
        const glm::mat3 inv = env->m_transform.getInvTransform();
        for ( int c = 0; c < 3; c++ )
            for ( int r = 0; r < 3; r++ )
                out.desc.env_inv[3 * c + r] = inv[c][r];
    }
    out.bind();
}
