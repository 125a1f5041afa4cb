// TEST INFRASTRUCTURE ONLY (container only; see eval_order_sample.cpp).
#include "Material.h"
#include "intersection.h"
#include "ray.h"

extern int g_eval_order_calls;

// Returns 1 when the cosine was taken from the direction wi held BEFORE sample() wrote it (the right
// operand of operator* evaluated first), 0 when from the direction sample() wrote, -1 if sample() did
// not run exactly once.  The statement is integrator.cpp:96-98 with the same names and types.
extern "C" int ref_eval_order_probe( void ) {
    alignas( Material ) static unsigned char image[sizeof( Material )] = { 0 };
    const Material& intersectedMaterial = *reinterpret_cast<const Material*>( image );
    alignas( Intersection ) static unsigned char isect_image[sizeof( Intersection )] = { 0 };
    Intersection* intersection = reinterpret_cast<Intersection*>( isect_image );
    intersection->normal       = vec3( 0.f, 0.f, 1.f );
    Ray* ray                   = nullptr;
    point2 uScattering( 0.5f, 0.5f );
    vec3 wi( 1.f, 0.f, 0.f ); // the light-sample direction left in wi: perpendicular to the normal, cosine 0
    float scatteringPdf = 0.f;
    g_eval_order_calls  = 0;
    color3 bsdf =
        intersectedMaterial.sample( ray, intersection, uScattering, &wi, &scatteringPdf ) *
        max( dot( intersection->normal, wi ), 0.f );
    if ( g_eval_order_calls != 1 ) return -1;
    return bsdf.x == 0.f ? 1 : 0;
}
