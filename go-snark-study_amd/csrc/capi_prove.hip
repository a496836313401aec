// TEMPORARY stubs (replaced by the poly / prove engines).
#include "runtime.h"
using namespace gs;
#define NI(name) return fail(GS_ERR_ARG, name ": not implemented yet")
extern "C" {
int gs_poly_mul(const uint64_t*, size_t, const uint64_t*, size_t, uint64_t*) { NI("gs_poly_mul"); }
int gs_poly_div(const uint64_t*, size_t, const uint64_t*, size_t, uint64_t*, uint64_t*) { NI("gs_poly_div"); }
int gs_poly_add(const uint64_t*, size_t, const uint64_t*, size_t, uint64_t*) { NI("gs_poly_add"); }
int gs_poly_sub(const uint64_t*, size_t, const uint64_t*, size_t, uint64_t*) { NI("gs_poly_sub"); }
int gs_poly_eval(const uint64_t*, size_t, const uint64_t*, uint64_t*) { NI("gs_poly_eval"); }
int gs_lagrange_interpolation(const uint64_t*, size_t, uint64_t*) { NI("gs_lagrange_interpolation"); }
int gs_zpoly(size_t, uint64_t*) { NI("gs_zpoly"); }
int gs_r1cs_to_px(size_t, size_t, const uint32_t*, const uint32_t*, const uint64_t*, const uint32_t*, const uint32_t*, const uint64_t*,
                  const uint32_t*, const uint32_t*, const uint64_t*, const uint64_t*, uint64_t*, uint64_t*, uint64_t*, uint64_t*) { NI("gs_r1cs_to_px"); }
int gs_groth16_pk_create(gs_handle, gs_handle, gs_handle, gs_handle, gs_handle, const uint64_t*, const uint64_t*, const uint64_t*,
                         const uint64_t*, const uint64_t*, const uint64_t*, size_t, size_t, size_t, gs_handle*) { NI("gs_groth16_pk_create"); }
int gs_groth16_prove(gs_handle, const uint64_t*, size_t, const uint64_t*, size_t, const uint64_t*, const uint64_t*, uint64_t*, int*) { NI("gs_groth16_prove"); }
int gs_groth16_prove_resident(gs_handle, gs_handle, gs_handle, const uint64_t*, const uint64_t*, uint64_t*, int*) { NI("gs_groth16_prove_resident"); }
int gs_pinocchio_pk_create(gs_handle, gs_handle, gs_handle, gs_handle, gs_handle, gs_handle, gs_handle, gs_handle, const uint64_t*,
                           size_t, size_t, size_t, gs_handle*) { NI("gs_pinocchio_pk_create"); }
int gs_pinocchio_prove(gs_handle, const uint64_t*, size_t, const uint64_t*, size_t, uint64_t*, int*) { NI("gs_pinocchio_prove"); }
}
