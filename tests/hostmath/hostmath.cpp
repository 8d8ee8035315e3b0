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

// ---- 3-dof groups (theseus_amd/csrc/unroll_g3.cuh): group 2 = SE2 (raw 4), 3 = SO3 (raw 9) ----
#include "unroll_g3.cuh"

extern "C" {

// out: g[3 * NR] (Xi, Xj, Z) gs[3] glr[1]; prior (edge = 0): the Xi third stays 0, Xj = the variable, Z = the target
void hm_g3_vjp(int group, int edge, const double* Xi, const double* Xj, const double* Z, const double* s, const double* wi,
               const double* wj, const double* di, const double* dj, const double* eps, double lam, int loss, double log_radius,
               double* out) {
  if (group == 2) {
    const Eps2<double> e{eps[0], eps[1]};
    if (edge) unroll3_vjp<UG_SE2, true>(Xi, Xj, Z, s, wi, wj, di, dj, e, lam, loss, log_radius, out, out + 12, out + 15);
    else unroll3_vjp<UG_SE2, false>(Xi, Xj, Z, s, wi, wj, di, dj, e, lam, loss, log_radius, out, out + 12, out + 15);
  } else {
    const Eps<double> e{eps[0], eps[1], eps[2]};
    if (edge) unroll3_vjp<UG_SO3, true>(Xi, Xj, Z, s, wi, wj, di, dj, e, lam, loss, log_radius, out, out + 27, out + 30);
    else unroll3_vjp<UG_SO3, false>(Xi, Xj, Z, s, wi, wj, di, dj, e, lam, loss, log_radius, out, out + 27, out + 30);
  }
}
}

// ---- bundle adjustment (theseus_amd/csrc/unroll_ba.cuh) ----
#include "unroll_ba.cuh"

extern "C" {

// calib = [focal, k1, k2]; out: gcam[12] gX[3] gfeat[2] gs[2] gcal[3] glr[1]
void hm_reproj_vjp(const double* cam, const double* X, const double* feat, const double* calib, const double* s, const double* wc,
                   const double* wp, const double* dc, const double* dp, double lam, int loss, double log_radius, double* out) {
  SE3<double> C;
  load(cam, C);
  unroll_reproj_vjp(C, X, feat, calib[0], calib[1], calib[2], s, wc, wp, dc, dp, lam, loss, log_radius, out, out + 12, out + 15,
                    out + 17, out + 19, out + 22);
}

// out: gX[3] gT[3] gs[3]
void hm_pt_prior_vjp(const double* X, const double* t, const double* s, const double* w, const double* d, double lam, double* out) {
  unroll_pt_prior_vjp(X, t, s, w, d, lam, out, out + 3, out + 6);
}
}
