// TEST INFRASTRUCTURE ONLY (CPU oracle) -- see mrt_oracle.h.
//
// Object / Ray / Intersection views over the flat scene description plus the restated
// Object::intersect family and intersectKdTree.
#pragma once
#include "../include/mrt.h"
#include "orc_math.h"

#include <cstring>
#include <vector>

namespace orc {

const float acne_eps = 1e-4f; // defines.h:36

// Ray (ray.h:5-30, ray.cpp:5-16)
struct Ray {
    V3 orig, dir;
    float tmax, tmin;
    int sign[3];
    V3 invdir;
    bool shadow           = false;
    bool hasDifferentials = true;
    Ray() {}
    Ray( V3 o, V3 d, float tmin_ = 0, float tmax_ = 100000 ) : orig( o ), dir( d ), tmax( tmax_ ), tmin( tmin_ ) {
        invdir  = v3( 1.f ) / d;
        sign[0] = d.x >= 0 ? 0 : 1;
        sign[1] = d.y >= 0 ? 0 : 1;
        sign[2] = d.z >= 0 ? 0 : 1;
    }
};

// The fields of Intersection (intersection.h:6-43) the path tracer reads.
struct Isect {
    V3 normal { 0, 0, 0 };
    V3 position { 0, 0, 0 };
    uint32_t materialIndex = 0;
    int32_t prim           = -1; // object index (not in the reference; used for parity checks)
    bool isOutside         = false;
    float u = 0, v = 0;   // interpolated texture coordinates (triangle.cpp:46-49)
    float bu = 0, bv = 0; // barycentric weights of p2, p3
};

struct Shape { // one Object (Object.h:10-26) or a ShapeLight's private shape
    int kind;
    uint32_t material;
    Xf xf;
    V3 p1, p2, p3, n1, n2, n3;
    V2 tex[3];
    V3 center;
    float radius;
};

inline Xf xf_from( const mrt_transform& t ) {
    Xf x;
    x.m   = m3_from( t.m );
    x.inv = m3_from( t.inv );
    x.t   = v3( t.t );
    return x;
}

inline Shape object_shape( const mrt_scene_desc* d, uint32_t i ) {
    Shape s {};
    s.kind     = d->obj_kind[i];
    s.material = d->obj_material[i];
    s.xf       = xf_from( d->transforms[d->obj_transform[i]] );
    if ( s.kind == MRT_TRIANGLE ) {
        const float* p = d->tri_p + 9 * (size_t)i;
        const float* n = d->tri_n + 9 * (size_t)i;
        const float* t = d->tri_uv + 6 * (size_t)i;
        s.p1           = v3( p );
        s.p2           = v3( p + 3 );
        s.p3           = v3( p + 6 );
        s.n1           = v3( n );
        s.n2           = v3( n + 3 );
        s.n3           = v3( n + 6 );
        for ( int k = 0; k < 3; k++ )
            s.tex[k] = V2 { t[2 * k], t[2 * k + 1] };
    }
    else if ( s.kind == MRT_SPHERE ) {
        s.center = v3( d->sphere + 4 * (size_t)i );
        s.radius = d->sphere[4 * (size_t)i + 3];
    }
    return s;
}

inline Shape light_shape( const mrt_scene_desc* d, uint32_t li ) {
    const mrt_light& l = d->lights[li];
    Shape s {};
    s.kind     = l.shape_kind;
    s.material = 0;
    s.xf       = xf_from( d->transforms[l.transform] );
    s.p1       = v3( l.tri_p );
    s.p2       = v3( l.tri_p + 3 );
    s.p3       = v3( l.tri_p + 6 );
    s.n1       = v3( l.tri_n );
    s.n2       = v3( l.tri_n + 3 );
    s.n3       = v3( l.tri_n + 6 );
    s.center   = v3( l.sphere );
    s.radius   = l.sphere[3];
    return s;
}

// Object::transformRay (Object.cpp:8-11)
inline void transform_ray( const Shape& s, V3& origin, V3& direction ) {
    origin    = s.xf.to( origin );
    direction = mul( s.xf.inv, direction );
}

// Triangle::intersect (shapes/triangle.cpp:4-84), differentials omitted (their results are unused).
inline bool triangle_intersect( const Shape& s, Ray* ray, Isect* isect ) {
    V3 origin = ray->orig;
    V3 dir    = ray->dir;
    transform_ray( s, origin, dir );

    V3 v1v2 = s.p2 - s.p1;
    V3 v1v3 = s.p3 - s.p1;

    V3 cross_rayDir_v1v3 = cross( dir, v1v3 );
    float det            = dot( v1v2, cross_rayDir_v1v3 );
    if ( det > -acne_eps && det < acne_eps ) return false;
    float inv_det = 1.f / det;

    V3 o_minus_p1 = origin - s.p1;
    float u       = dot( o_minus_p1, cross_rayDir_v1v3 ) * inv_det;
    if ( u < 0.f || u > 1.f ) return false;

    V3 cross_oMinusp1_v1v2 = cross( o_minus_p1, v1v2 );
    float v                = dot( dir, cross_oMinusp1_v1v2 ) * inv_det;
    if ( v < 0.f || u + v > 1.f ) return false;

    float t = dot( v1v3, cross_oMinusp1_v1v2 ) * inv_det;
    if ( t >= ray->tmin && t <= ray->tmax ) {
        ray->tmax            = t;
        isect->position      = ray->orig + ( t * ray->dir );
        isect->materialIndex = s.material;
        isect->isOutside     = !( det < 0.f );
        isect->bu            = u;
        isect->bv            = v;
        if ( ray->shadow ) return true;

        V3 objectNormal = normalize( s.n2 * u + s.n3 * v + s.n1 * ( 1.f - u - v ) );
        V3 normal       = s.xf.vec_from( objectNormal );
        isect->normal   = normalize( normal );

        float w  = 1.f - u - v;
        isect->u = s.tex[1].x * u + s.tex[2].x * v + s.tex[0].x * w;
        isect->v = s.tex[1].y * u + s.tex[2].y * v + s.tex[0].y * w;
        return true;
    }
    return false;
}

// The hit bookkeeping shared by both branches of Sphere::intersect (shapes/sphere.cpp:71-99, 134-158).
inline void sphere_fill( const Shape& s, Ray* ray, Isect* isect, V3 dir, V3 objectPoint, V3 objectNormal ) {
    V3 normal        = s.xf.vec_from( objectNormal );
    isect->isOutside = dot( dir, objectNormal ) < 0;
    isect->normal    = normalize( normal );

    float pi     = (float)3.14159265358979323846;
    float radius = length( objectPoint );
    float phi    = std::atan2( objectPoint.y, objectPoint.x );
    if ( phi < 0 ) phi += 2.f * pi;
    // PHIMAX = (2.f * M_PI) and THETAMAX = M_PI are double expressions (sphere.cpp:4-6)
    float u     = (float)( 1.f - (double)phi / ( 2.f * 3.14159265358979323846 ) );
    float theta = std::acos( clampf( objectPoint.z / radius, -1.f, 1.f ) );
    float v     = (float)( (double)( theta - 0.f ) / ( 3.14159265358979323846 - 0.f ) );
    isect->u    = u;
    isect->v    = v;
}

// Sphere::intersect (shapes/sphere.cpp:52-180)
inline bool sphere_intersect( const Shape& s, Ray* ray, Isect* isect ) {
    V3 origin = ray->orig;
    V3 dir    = ray->dir;
    transform_ray( s, origin, dir );

    V3 oc   = origin - v3( 0.f );
    float a = dot( dir, dir );
    float b = 2.f * dot( dir, oc );
    float c = dot( oc, oc ) - 1.f;

    float delta = b * b - 4.f * a * c;

    if ( delta == 0 ) {
        float t = -b / ( 2.f * a );
        if ( t >= ray->tmin && t <= ray->tmax ) {
            ray->tmax            = t;
            isect->position      = ray->orig + ( t * ray->dir );
            isect->materialIndex = s.material;
            if ( ray->shadow ) return true;
            V3 objectPoint  = s.xf.to( isect->position );
            V3 objectNormal = objectPoint - v3( 0.f );
            sphere_fill( s, ray, isect, dir, objectPoint, objectNormal );
            return true;
        }
    }
    else if ( delta > 0 ) {
        float t1 = ( -b + sqrtf( delta ) ) / ( 2.f * a );
        float t2 = ( -b - sqrtf( delta ) ) / ( 2.f * a );
        float t;
        if ( t1 >= ray->tmin && t1 < ray->tmax && t2 >= ray->tmin && t2 < ray->tmax ) {
            t = ( t2 < t1 ) ? t2 : t1; // std::min(t1, t2)
        }
        else if ( t1 >= ray->tmin && t1 < ray->tmax ) { t = t1; }
        else if ( t2 >= ray->tmin && t2 < ray->tmax ) { t = t2; }
        else { return false; }
        ray->tmax            = t;
        isect->position      = ray->orig + ( t * ray->dir );
        isect->materialIndex = s.material;
        if ( ray->shadow ) return true;

        V3 objectPoint  = origin + ( t * dir );
        V3 objectNormal = normalize( objectPoint - v3( 0.f ) );
        sphere_fill( s, ray, isect, dir, objectPoint, objectNormal );
        return true;
    }
    return false;
}

// Plane::intersect (shapes/plane.cpp:6-68), differentials omitted.
inline bool plane_intersect( const Shape& s, Ray* ray, Isect* isect ) {
    V3 origin = ray->orig;
    V3 dir    = ray->dir;
    transform_ray( s, origin, dir );

    float rayPz = origin.z;
    float rayDz = dir.z;
    float t     = -rayPz / rayDz;
    if ( t < 0 || t > ray->tmax ) return false;
    V3 x = origin + t * dir;
    if ( x.x < -1 || x.x > 1 || x.y < -1 || x.y > 1 ) { return false; }

    ray->tmax            = t;
    isect->position      = ray->orig + t * ray->dir;
    isect->materialIndex = s.material;
    if ( ray->shadow ) return true;

    V3 objectNormal  = v3( 0.f, 0.f, 1.f );
    V3 normal        = s.xf.vec_from( objectNormal );
    isect->isOutside = true;
    isect->normal    = normalize( normal );
    isect->u         = ( 1.f + x.x ) * .5f;
    isect->v         = ( 1.f + x.y ) * .5f;
    return true;
}

inline bool shape_intersect( const Shape& s, Ray* ray, Isect* isect ) {
    switch ( s.kind ) {
    case MRT_TRIANGLE:
        return triangle_intersect( s, ray, isect );
    case MRT_SPHERE:
        return sphere_intersect( s, ray, isect );
    case MRT_PLANE:
        return plane_intersect( s, ray, isect );
    }
    return false;
}

// intersectAabb (kdtree.cpp:162-180)
inline bool intersect_aabb( Ray* r, V3 mn, V3 mx ) {
    float tmin, tmax, tymin, tymax, tzmin, tzmax;
    V3 bounds[2] = { mn, mx };
    tmin         = ( bounds[r->sign[0]].x - r->orig.x ) * r->invdir.x;
    tmax         = ( bounds[1 - r->sign[0]].x - r->orig.x ) * r->invdir.x;
    tymin        = ( bounds[r->sign[1]].y - r->orig.y ) * r->invdir.y;
    tymax        = ( bounds[1 - r->sign[1]].y - r->orig.y ) * r->invdir.y;
    if ( ( tmin > tymax ) || ( tymin > tmax ) ) return false;
    if ( tymin > tmin ) tmin = tymin;
    if ( tymax < tmax ) tmax = tymax;
    tzmin = ( bounds[r->sign[2]].z - r->orig.z ) * r->invdir.z;
    tzmax = ( bounds[1 - r->sign[2]].z - r->orig.z ) * r->invdir.z;
    if ( ( tmin > tzmax ) || ( tzmin > tmax ) ) return false;
    if ( tzmin > tmin ) tmin = tzmin;
    if ( tzmax < tmax ) tmax = tzmax;
    if ( tmin > r->tmin ) r->tmin = tmin;
    if ( tmax < r->tmax ) r->tmax = tmax;
    return true;
}

// A scene prepared for tracing: shapes materialised once, tree arrays borrowed from the desc.
struct TraceScene {
    const mrt_scene_desc* d;
    std::vector<Shape> shapes;
    std::vector<uint8_t> mat_emissive;     // !(r == 0 && g == 0 && b == 0)  (kdtree.cpp:533-534)
    std::vector<uint8_t> mat_emissive_sum; // !isBlack(m_emission)           (integrator.cpp:110,147)
    explicit TraceScene( const mrt_scene_desc* desc ) : d( desc ) {
        shapes.reserve( desc->n_objects );
        for ( uint32_t i = 0; i < desc->n_objects; i++ )
            shapes.push_back( object_shape( desc, i ) );
        mat_emissive.resize( desc->n_materials );
        for ( uint32_t i = 0; i < desc->n_materials; i++ ) {
            const float* e  = desc->materials[i].emission;
            mat_emissive[i] = !( e[0] == 0.f && e[1] == 0.f && e[2] == 0.f );
            mat_emissive_sum.push_back( !( e[0] + e[1] + e[2] == 0. ) );
        }
    }
};

struct StackNode { // kdtree.cpp:39-43
    float tmin, tmax;
    uint32_t node;
};

// traverse (kdtree.cpp:467-553) over the flat node array.
inline bool traverse( const TraceScene& sc, Ray* ray, Isect* intersection, mrt_trace_stats* st ) {
    const mrt_kdnode* nodes = sc.d->nodes;
    const uint32_t* refs    = sc.d->refs;
    StackNode stack[256];
    int sp      = 0;
    stack[sp++] = StackNode { ray->tmin, ray->tmax, 0u };
    float a, b, t;
    while ( sp > 0 ) {
        StackNode cur = stack[--sp];
        a             = cur.tmin;
        b             = cur.tmax;
        uint32_t node = cur.node;
        while ( ( nodes[node].w1 & 3u ) != 3u ) {
            st->inner++;
            int axis = nodes[node].w1 & 3u;
            float split;
            std::memcpy( &split, &nodes[node].w0, 4 );
            uint32_t left  = nodes[node].w1 >> 2;
            uint32_t right = left + 1;
            float diff     = split - ray->orig[axis];
            t              = diff / ray->dir[axis];
            uint32_t near_, far_;
            if ( diff > 0.0 ) {
                near_ = left;
                far_  = right;
            }
            else {
                near_ = right;
                far_  = left;
            }
            if ( ( t > b ) || ( t < 0.0 ) ) { node = near_; }
            else {
                if ( t < a ) { node = far_; }
                else {
                    stack[sp++] = StackNode { t, b, far_ };
                    node        = near_;
                    b           = t;
                }
            }
        }
        st->leaves++;
        ray->tmin            = a;
        ray->tmax            = b;
        bool hasIntersection = false;
        float dist           = 0;
        uint32_t off         = nodes[node].w0;
        uint32_t cnt         = nodes[node].w1 >> 2;
        for ( uint32_t i = 0; i < cnt; i++ ) {
            Isect temp;
            uint32_t obj = refs[off + i];
            st->refs++;
            st->prim_tests++;
            if ( shape_intersect( sc.shapes[obj], ray, &temp ) ) {
                temp.prim       = (int32_t)obj;
                float temp_dist = ray->tmax;
                if ( hasIntersection ) {
                    if ( temp_dist < dist ) {
                        dist          = temp_dist;
                        *intersection = temp;
                    }
                }
                else {
                    if ( ray->shadow ) {
                        if ( !sc.mat_emissive[temp.materialIndex] ) {
                            ray->tmax     = temp_dist;
                            *intersection = temp; // not in the reference (output unused there)
                            return true;
                        }
                    }
                    else {
                        hasIntersection = true;
                        *intersection   = temp;
                        dist            = temp_dist;
                    }
                }
            }
        }
        if ( hasIntersection ) {
            ray->tmax = dist;
            return true;
        }
    }
    return false;
}

// intersectKdTree (kdtree.cpp:555-613)
inline bool intersect_kdtree( const TraceScene& sc, Ray* ray, Isect* intersection, mrt_trace_stats* st ) {
    bool hasIntersection = false;
    float dist           = 0;
    const mrt_scene_desc* d = sc.d;

    Ray ray_backup;
    if ( d->n_out_of_tree > 0 ) {
        ray_backup = Ray( ray->orig, ray->dir, ray->tmin, ray->tmax );
        // the copy constructor call of kdtree.cpp:566 does not carry Ray::shadow over:
        // the backup ray is always a closest-hit ray
    }

    if ( d->n_nodes > 0 && intersect_aabb( ray, v3( d->root_min ), v3( d->root_max ) ) ) {
        hasIntersection = traverse( sc, ray, intersection, st );
    }

    if ( hasIntersection ) {
        if ( ray->shadow ) { return true; }
        ray_backup.tmax = ray->tmax;
        dist            = ray->tmax;
    }

    for ( uint32_t i = 0; i < d->n_out_of_tree; i++ ) {
        Isect temp;
        uint32_t obj = d->out_of_tree[i];
        st->prim_tests++;
        if ( shape_intersect( sc.shapes[obj], &ray_backup, &temp ) ) {
            temp.prim       = (int32_t)obj;
            float temp_dist = ray_backup.tmax;
            if ( hasIntersection ) {
                if ( temp_dist < dist ) {
                    dist          = temp_dist;
                    *intersection = temp;
                    ray->tmax     = dist;
                }
            }
            else {
                hasIntersection = true;
                *intersection   = temp;
                dist            = temp_dist;
                ray->tmax       = dist;
                if ( ray->shadow ) { return true; }
            }
        }
    }
    return hasIntersection;
}

} // namespace orc
