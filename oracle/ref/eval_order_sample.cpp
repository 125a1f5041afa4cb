// TEST INFRASTRUCTURE ONLY (container only; part of oracle/_ref/libmrt_ref.so).
//
// Compiler evidence for one expression of the reference, not parity: integrator.cpp:96-98 reads
//     color3 bsdf = intersectedMaterial.sample( ray, intersection, uScattering, &wi, &scatteringPdf ) *
//                   max( dot( intersection->normal, wi ), 0.f );
// The left operand WRITES wi, the right operand READS it, and C++ does not sequence the two operands of
// operator*.  Material.cpp, where Material::sample lives, cannot be compiled here (it includes Light.h,
// which needs a header the image lacks), so this file gives the class of the reference's own Material.h an
// INSTRUMENTED sample() -- it only records that it ran and writes a known wi -- in a translation unit of
// its own, as the real one is, and eval_order_expr.cpp evaluates an expression of exactly that shape
// against it with the flags of oracle/Makefile.  What comes out is which operand g++ evaluates first.
#include "Material.h"

int g_eval_order_calls = 0;

color3 Material::sample( Ray*, Intersection*, const vec2&, vec3* wi, float* scatteringPdf ) const {
    g_eval_order_calls++;
    *wi            = vec3( 0.f, 0.f, 1.f ); // along the normal the probe uses: the cosine of the NEW direction is 1
    *scatteringPdf = 1.f;
    return color3( 1.f, 1.f, 1.f );
}
