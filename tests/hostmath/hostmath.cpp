// TEST-ONLY: the device maths of theseus_amd/csrc/unroll_se3.cuh compiled for the host (tests/test_unroll_math_host.py).
#include "unroll_se3.cuh"

using namespace thx;

static void load(const double* p, SE3<double>& X) {
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) X.R[3 * i + j] = p[4 * i + j];
    X.t[i] = p[4 * i + 3];
  }
}

extern "C" {

// out: gXi[12] gXj[12] gZ[12] gs[6] glr[1]
void hm_edge_vjp(const double* Xi, const double* Xj, const double* Z, const double* s, const double* wi, const double* wj,
                 const double* di, const double* dj, const double* eps, double lam, int loss, double log_radius, double* out) {
  SE3<double> A, B, C;
  load(Xi, A);
  load(Xj, B);
  load(Z, C);
  const Eps<double> e{eps[0], eps[1], eps[2]};
  unroll_edge_vjp(A, B, C, s, wi, wj, di, dj, e, out, out + 12, out + 24, out + 36, lam, loss, log_radius, out + 42);
}

// out: gX[12] gT[12] gs[6] glr[1]
void hm_prior_vjp(const double* X, const double* T, const double* s, const double* w, const double* d, const double* eps,
                  double lam, int loss, double log_radius, double* out) {
  SE3<double> A, B;
  load(X, A);
  load(T, B);
  const Eps<double> e{eps[0], eps[1], eps[2]};
  unroll_prior_vjp(A, B, s, w, d, e, out, out + 12, out + 24, lam, loss, log_radius, out + 30);
}
}
