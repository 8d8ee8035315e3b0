// SE2 pose-graph kernels (2-D SLAM): the generic 3-dof kernels of pg3_generic.cuh instantiated with the SE2 group functor
// of lie_se2.cuh (4-element records [x, y, cos, sin], 3x3 blocks).
#include "common.cuh"
#include "lie_se2.cuh"
#include "pg3_generic.cuh"

namespace thx {

struct GroupSE2 {
  static constexpr int REC = 4, DOF = 3;
  using X = SE2<double>;
  using Eps = Eps2<double>;
  template <typename T>
  static __device__ __forceinline__ X load(const T* __restrict__ p) { return se2_load(p); }
  template <typename T>
  static __device__ __forceinline__ void store(T* __restrict__ p, const X& x) { se2_store(p, x); }
  static __device__ __forceinline__ void inv(const X& a, X& y) { se2_inv(a, y); }
  static __device__ __forceinline__ void mul(const X& a, const X& b, X& z) { se2_mul(a, b, z); }
  static __device__ __forceinline__ void exp(const double* xi, const Eps& eps, X& x, double* J) { se2_exp<double>(xi, eps, x, J); }
  static __device__ __forceinline__ void log_jlog(const X& x, const Eps& eps, double* xi, double* J, bool want_jac) {
    se2_log_jlog(x, eps, xi, J, want_jac);
  }
  static __device__ __forceinline__ void adjoint(const X& x, double* A) { se2_adjoint(x, A); }
};

static inline Eps2<double> make_eps2(const thx_se2_eps* e, int dtype) {
  // the reference compares an fp32 angle with the fp32-rounded threshold
  return dtype == THX_F32 ? Eps2<double>{(double)(float)e->near_zero, (double)(float)e->d_near_zero}
                          : Eps2<double>{e->near_zero, e->d_near_zero};
}

}  // namespace thx

using namespace thx;

extern "C" {

int thx_pg2_assemble(const thx_pg_structure* s, const thx_pg_data* d, void* H, int64_t ld, void* g, int dtype,
                     const thx_se2_eps* eps, void* stream) {
  if (!eps) return fail("null eps");
  return pg3_assemble<GroupSE2>(s, d, H, ld, g, dtype, make_eps2(eps, dtype), stream, "thx_pg2_assemble");
}

int thx_pg2_error(const thx_pg_structure* s, const thx_pg_data* d, void* partials, void* err, int dtype,
                  const thx_se2_eps* eps, void* stream) {
  if (!eps) return fail("null eps");
  return pg3_error<GroupSE2>(s, d, partials, err, dtype, make_eps2(eps, dtype), stream, "thx_pg2_error");
}

int thx_pg2_jacobians(const thx_pg_structure* s, const thx_pg_data* d, void* J0, void* J1, void* eb, void* Jp, void* ep,
                      int dtype, const thx_se2_eps* eps, void* stream) {
  if (!eps) return fail("null eps");
  return pg3_jacobians<GroupSE2>(s, d, J0, J1, eb, Jp, ep, dtype, make_eps2(eps, dtype), stream, "thx_pg2_jacobians");
}

int thx_se2_retract(const void* poses, const void* delta, int64_t ldd, double step, const uint8_t* ignore_mask, void* out,
                    int32_t P, int32_t B, int dtype, const thx_se2_eps* eps, void* stream) {
  if (!eps) return fail("bad retract args");
  return g3_retract<GroupSE2>(poses, delta, ldd, step, ignore_mask, out, P, B, dtype, make_eps2(eps, dtype), stream,
                              "thx_se2_retract");
}

int thx_se2_op(int op, const void* a, const void* b, void* out, void* jac, int64_t N, int dtype, const thx_se2_eps* eps,
               void* stream) {
  if (N > 0 && !eps) return fail("thx_se2_op: bad arguments");
  return g3_op<GroupSE2>(op, a, b, out, jac, N, dtype, N > 0 ? make_eps2(eps, dtype) : Eps2<double>{0, 0}, stream, "thx_se2_op");
}

}  // extern "C"
