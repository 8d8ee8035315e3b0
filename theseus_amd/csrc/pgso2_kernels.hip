// SO2 pose-graph kernels (planar rotation-only graphs: th.Between / th.Difference on th.SO2 variables): the generic small-group
// kernels of pg3_generic.cuh instantiated with the SO2 group functor below -- records [cos, sin], ONE degree of freedom, 1x1
// blocks.  Written from the closed forms of theseus/geometry/so2.py: exp = (cos theta, sin theta) with Jacobian 1 (:167-186,
// update_from_angle :96-100), log = atan2(sin, cos) with Jacobian 1 (:206-223), adjoint 1 (:116-117), compose by the angle-
// addition formulas (:225-231), inverse (cos, -sin) (:233-235).  No Taylor switches: the group has no thresholds on this path.
// NO re-normalisation of the record anywhere (the reference's constructors pass tensors through unchanged: SURVEY.md App. A).
#include "common.cuh"
#include "lie.cuh"
#include "pg3_generic.cuh"

namespace thx {

struct SO2c {
  double c, s;
};
struct NoEps {};

struct GroupSO2 {
  static constexpr int REC = 2, DOF = 1;
  using X = SO2c;
  using Eps = NoEps;
  template <typename T>
  static __device__ __forceinline__ X load(const T* __restrict__ p) { return X{(double)p[0], (double)p[1]}; }
  template <typename T>
  static __device__ __forceinline__ void store(T* __restrict__ p, const X& x) {
    p[0] = (T)x.c;
    p[1] = (T)x.s;
  }
  static __device__ __forceinline__ void inv(const X& a, X& y) { y = X{a.c, -a.s}; }
  static __device__ __forceinline__ void mul(const X& a, const X& b, X& z) {
    z.c = a.c * b.c - a.s * b.s;
    z.s = a.s * b.c + a.c * b.s;
  }
  static __device__ __forceinline__ void exp(const double* xi, const Eps&, X& x, double* J) {
    x.c = t_cos(xi[0]);
    x.s = t_sin(xi[0]);
    if (J) J[0] = 1.0;
  }
  static __device__ __forceinline__ void log_jlog(const X& x, const Eps&, double* xi, double* J, bool want_jac) {
    xi[0] = t_atan2(x.s, x.c);
    if (want_jac) J[0] = 1.0;
  }
  static __device__ __forceinline__ void adjoint(const X&, double* A) { A[0] = 1.0; }
};

}  // namespace thx

using namespace thx;

extern "C" {

int thx_pgso2_assemble(const thx_pg_structure* s, const thx_pg_data* d, void* H, int64_t ld, void* g, int dtype, void* stream) {
  return pg3_assemble<GroupSO2>(s, d, H, ld, g, dtype, NoEps{}, stream, "thx_pgso2_assemble");
}

int thx_pgso2_error(const thx_pg_structure* s, const thx_pg_data* d, void* partials, void* err, int dtype, void* stream) {
  return pg3_error<GroupSO2>(s, d, partials, err, dtype, NoEps{}, stream, "thx_pgso2_error");
}

int thx_pgso2_jacobians(const thx_pg_structure* s, const thx_pg_data* d, void* J0, void* J1, void* eb, void* Jp, void* ep,
                        int dtype, void* stream) {
  return pg3_jacobians<GroupSO2>(s, d, J0, J1, eb, Jp, ep, dtype, NoEps{}, stream, "thx_pgso2_jacobians");
}

int thx_so2_retract(const void* poses, const void* delta, int64_t ldd, double step, const uint8_t* ignore_mask, void* out,
                    int32_t P, int32_t B, int dtype, void* stream) {
  return g3_retract<GroupSO2>(poses, delta, ldd, step, ignore_mask, out, P, B, dtype, NoEps{}, stream, "thx_so2_retract");
}

int thx_so2_op(int op, const void* a, const void* b, void* out, void* jac, int64_t N, int dtype, void* stream) {
  return g3_op<GroupSO2>(op, a, b, out, jac, N, dtype, NoEps{}, stream, "thx_so2_op");
}

}  // extern "C"
