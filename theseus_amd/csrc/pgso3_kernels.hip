// SO3 pose-graph kernels (rotation-only graphs: th.Between / th.Difference on th.SO3 variables): the generic 3-dof kernels
// of pg3_generic.cuh instantiated with the SO3 group functor of lie_so3.cuh (3x3 records, 3x3 blocks).
// Replaces, for SO3 variables, what pg_kernels.hip replaces for SE3: torchlie's SO3 closed forms
// (torchlie/torchlie/functional/so3_impl.py:220-479) under theseus/geometry/so3.py, Between / Local
// (embodied/measurements/between.py:38-45, embodied/misc/local_cost_fn.py:42-61), retraction (geometry/lie_group.py:197-198).
#include "common.cuh"
#include "lie_so3.cuh"
#include "pg3_generic.cuh"

using namespace thx;

static inline Eps<double> make_eps_so3(const thx_lie_eps* e, int dtype) {
  // the reference compares an fp32 angle with the fp32-rounded threshold
  return dtype == THX_F32 ? Eps<double>{(double)(float)e->near_zero, (double)(float)e->d_near_zero, (double)(float)e->near_pi}
                          : Eps<double>{e->near_zero, e->d_near_zero, e->near_pi};
}

extern "C" {

int thx_pgso3_assemble(const thx_pg_structure* s, const thx_pg_data* d, void* H, int64_t ld, void* g, int dtype,
                       const thx_lie_eps* eps, void* stream) {
  if (!eps) return fail("null eps");
  return pg3_assemble<GroupSO3>(s, d, H, ld, g, dtype, make_eps_so3(eps, dtype), stream, "thx_pgso3_assemble");
}

int thx_pgso3_error(const thx_pg_structure* s, const thx_pg_data* d, void* partials, void* err, int dtype,
                    const thx_lie_eps* eps, void* stream) {
  if (!eps) return fail("null eps");
  return pg3_error<GroupSO3>(s, d, partials, err, dtype, make_eps_so3(eps, dtype), stream, "thx_pgso3_error");
}

int thx_pgso3_jacobians(const thx_pg_structure* s, const thx_pg_data* d, void* J0, void* J1, void* eb, void* Jp, void* ep,
                        int dtype, const thx_lie_eps* eps, void* stream) {
  if (!eps) return fail("null eps");
  return pg3_jacobians<GroupSO3>(s, d, J0, J1, eb, Jp, ep, dtype, make_eps_so3(eps, dtype), stream, "thx_pgso3_jacobians");
}

int thx_so3_retract(const void* poses, const void* delta, int64_t ldd, double step, const uint8_t* ignore_mask, void* out,
                    int32_t P, int32_t B, int dtype, const thx_lie_eps* eps, void* stream) {
  if (!eps) return fail("bad retract args");
  return g3_retract<GroupSO3>(poses, delta, ldd, step, ignore_mask, out, P, B, dtype, make_eps_so3(eps, dtype), stream,
                              "thx_so3_retract");
}

int thx_so3_op(int op, const void* a, const void* b, void* out, void* jac, int64_t N, int dtype, const thx_lie_eps* eps,
               void* stream) {
  if (N > 0 && !eps) return fail("thx_so3_op: bad arguments");
  return g3_op<GroupSO3>(op, a, b, out, jac, N, dtype, N > 0 ? make_eps_so3(eps, dtype) : Eps<double>{0, 0, 0}, stream,
                         "thx_so3_op");
}

}  // extern "C"
