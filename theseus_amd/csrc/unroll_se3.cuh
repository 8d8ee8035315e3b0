// BackwardMode.UNROLL / TRUNCATED on SE3 pose graphs (theseus/optimizer/nonlinear/nonlinear_least_squares.py:223-292): the
// reference differentiates THROUGH every iteration delta = (H + D)^-1 g, H = A^T A and g = A^T b both in the graph.  With
// w = (H + D)^-1 dL/d delta (one solve with that iteration's factor) the iteration's contribution to every gradient is the
// gradient of the SCALAR
//     psi = w^T g - w^T (H + D) delta,   D = lambda I (spherical: constant) or lambda diag(H) + eps (ellipsoidal, dense_solver.py:38-64)
//         = - sum_costs (J w)_c . (r_c + (J delta)_c)  -  [ellipsoidal]  lambda sum_i w_i delta_i H_ii        (w, delta held constant)
// and, cost by cost (Between: E = Z^-1 Xi^-1 Xj, r = s * log E, J_j = s * Jlog(E), J_i = -s * Jlog(E) Ad((Xi^-1 Xj)^-1)),
//     phi = - sum_r s_r^2 (Jlog q_w)_r (log(E)_r + (Jlog q_delta)_r),   q_v = v_j - Ad(D^-1) v_i,   D = Xi^-1 Xj
//           - [ellipsoidal] lambda sum_r s_r^2 sum_k ((J_j)_rk^2 w_jk delta_jk + (J_i)_rk^2 w_ik delta_ik)       (H_ii = sum of squares of column i)
// (Difference / Local priors: E = T^-1 X, q_v = v, one Jacobian).  This header evaluates phi and its derivative along ONE direction of the
// raw 3 x 4 entries of Xi, Xj, Z in forward mode (Dual<double> through the closed forms of lie.cuh): inverse / compose / adjoint /
// the Jlog closed forms have plain autograd graphs in the reference, log(E)'s own derivative is torchlie's passthrough backward
// (se3_impl.py:487-493): d log = Jlog [E_R^T dE_t ; vee(E_R^T dE_R) / 2] -- the convention thx_pg_vjp (vjp_se3.cuh) already
// follows for the implicit mode, of which this is the extension to the poses and to the Hessian term.
// Plain C++ templates over lie.cuh / dual.cuh: also compiled for the HOST by tests/hostmath (the maths is checked there against
// torch autograd through the oracle, without a GPU).
#pragma once
#include "dual.cuh"
#include "lie.cuh"
#include "robust.cuh"

namespace thx {

using UD = Dual<double>;

// q = vj - Ad(Dinv) vi,  Ad(X) = [[R, hat(t) R], [0, R]]  (tangent order [lin, ang])
template <typename S>
__device__ __forceinline__ void unroll_q(const SE3<S>& Dinv, const double* vi, const double* vj, S* q) {
  S vl[3] = {S(vi[0]), S(vi[1]), S(vi[2])}, va[3] = {S(vi[3]), S(vi[4]), S(vi[5])};
  S Rl[3], Ra[3], tx[3];
  mat3_vec(Dinv.R, vl, Rl);
  mat3_vec(Dinv.R, va, Ra);
  cross3(Dinv.t, Ra, tx);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    q[k] = S(vj[k]) - (Rl[k] + tx[k]);
    q[3 + k] = S(vj[3 + k]) - Ra[k];
  }
}

// a = Jlog q,  Jlog = [[Jr, Jt], [0, Jr]]
template <typename S>
__device__ __forceinline__ void unroll_jlog_apply(const S* Jr, const S* Jt, const S* q, S* a) {
  S t0[3], t1[3], t2[3];
  mat3_vec(Jr, q, t0);
  mat3_vec(Jt, q + 3, t1);
  mat3_vec(Jr, q + 3, t2);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    a[i] = t0[i] + t1[i];
    a[3 + i] = t2[i];
  }
}

// log(E) and Jlog(E) in dual arithmetic with the reference's passthrough backward for log: xi.d := Jlog [E_R^T dE_t ; vee(E_R^T dE_R)/2]
__device__ __forceinline__ void unroll_log_jlog(const SE3<UD>& E, const Eps<UD>& eps, UD* xi, UD* Jr, UD* Jt) {
  se3_log_jlog(E, eps, xi, Jr, Jt, true);
  double R[9], dR[9], dt[3], M[9], u[6], jr[9], jt[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    R[i] = E.R[i].v;
    dR[i] = E.R[i].d;
    jr[i] = Jr[i].v;
    jt[i] = Jt[i].v;
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) dt[i] = E.t[i].d;
  mat3_tmul(R, dR, M);
  mat3_tvec(R, dt, u);
  u[3] = 0.5 * (M[7] - M[5]);
  u[4] = 0.5 * (M[2] - M[6]);
  u[5] = 0.5 * (M[3] - M[1]);
  double t0[3], t1[3], t2[3];
  mat3_vec(jr, u, t0);
  mat3_vec(jt, u + 3, t1);
  mat3_vec(jr, u + 3, t2);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    xi[i].d = t0[i] + t1[i];
    xi[3 + i].d = t2[i];
  }
}

// phi of a Between cost (value and directional derivative); a_out / b_out (may be null): (Jlog q_w)_r and log(E)_r + (Jlog q_delta)_r;
// lam != 0: ellipsoidal damping's term; the weight gradient of that term goes to ell_gs (may be null): d/ds_r = 2 s_r (...)_r
// phi_r[6] / x_r[6]: the rows of phi (phi = sum_r phi_r: what a plain cost contributes) and the squared weighted errors
// x_r = (s_r log(E)_r)^2 a robust loss looks at -- value and directional derivative.
__device__ __forceinline__ void unroll_edge_phi(const SE3<UD>& Xi, const SE3<UD>& Xj, const SE3<UD>& Z, const double* s,
                                                const double* wi, const double* wj, const double* di, const double* dj,
                                                const Eps<UD>& eps, double lam, UD* phi_r, UD* x_r, double* a_out, double* b_out,
                                                double* ell_rows, double* xi_out) {
  SE3<UD> Xii, D, Zi, E, Dinv;
  se3_inv(Xi, Xii);
  se3_mul(Xii, Xj, D);
  se3_inv(Z, Zi);
  se3_mul(Zi, D, E);
  se3_inv(D, Dinv);
  UD xi[6], Jr[9], Jt[9], qw[6], qd[6], a[6], c[6];
  unroll_log_jlog(E, eps, xi, Jr, Jt);
  unroll_q(Dinv, wi, wj, qw);
  unroll_q(Dinv, di, dj, qd);
  unroll_jlog_apply(Jr, Jt, qw, a);
  unroll_jlog_apply(Jr, Jt, qd, c);
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    const UD bsum = xi[r] + c[r];
    phi_r[r] = UD(0.0) - UD(s[r] * s[r]) * (a[r] * bsum);
    x_r[r] = UD(s[r] * s[r]) * (xi[r] * xi[r]);
    if (a_out) a_out[r] = a[r].v;
    if (b_out) b_out[r] = bsum.v;
    if (xi_out) xi_out[r] = xi[r].v;
  }
  if (lam != 0.0) {
    // J_j = [[Jr, Jt], [0, Jr]];  J_i = -J_j Ad(Dinv) = -[[Jr R, Jr hat(t) R + Jt R], [0, Jr R]]  (signs drop out of the squares)
    UD JrR[9], hR[9], JrhR[9], JtR[9], TR[9];
    mat3_mul(Jr, Dinv.R, JrR);
#pragma unroll
    for (int k = 0; k < 3; ++k) {   // column k of hat(t) R = t x R[:, k]
      UD col[3] = {Dinv.R[k], Dinv.R[3 + k], Dinv.R[6 + k]}, cr[3];
      cross3(Dinv.t, col, cr);
      hR[k] = cr[0];
      hR[3 + k] = cr[1];
      hR[6 + k] = cr[2];
    }
    mat3_mul(Jr, hR, JrhR);
    mat3_mul(Jt, Dinv.R, JtR);
#pragma unroll
    for (int k = 0; k < 9; ++k) TR[k] = JrhR[k] + JtR[k];
    // row r of the ellipsoidal term (without s_r^2): sum_k (J_j)_rk^2 w_jk delta_jk + (J_i)_rk^2 w_ik delta_ik
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      UD top(0.0), bot(0.0);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        top = top + Jr[3 * r + k] * Jr[3 * r + k] * UD(wj[k] * dj[k]) + Jt[3 * r + k] * Jt[3 * r + k] * UD(wj[3 + k] * dj[3 + k]) +
              JrR[3 * r + k] * JrR[3 * r + k] * UD(wi[k] * di[k]) + TR[3 * r + k] * TR[3 * r + k] * UD(wi[3 + k] * di[3 + k]);
        bot = bot + Jr[3 * r + k] * Jr[3 * r + k] * UD(wj[3 + k] * dj[3 + k]) + JrR[3 * r + k] * JrR[3 * r + k] * UD(wi[3 + k] * di[3 + k]);
      }
      phi_r[r] = phi_r[r] - UD(lam * s[r] * s[r]) * top;
      phi_r[3 + r] = phi_r[3 + r] - UD(lam * s[3 + r] * s[3 + r]) * bot;
      if (ell_rows) {
        ell_rows[r] = top.v;
        ell_rows[3 + r] = bot.v;
      }
    }
  }
}

// phi of a Difference / Local prior: E = T^-1 X, J = Jlog(E)
__device__ __forceinline__ void unroll_prior_phi(const SE3<UD>& X, const SE3<UD>& T, const double* s, const double* w,
                                                 const double* d, const Eps<UD>& eps, double lam, UD* phi_r, UD* x_r, double* a_out,
                                                 double* b_out, double* ell_rows, double* xi_out) {
  SE3<UD> Ti, E;
  se3_inv(T, Ti);
  se3_mul(Ti, X, E);
  UD xi[6], Jr[9], Jt[9], qw[6], qd[6], a[6], c[6];
  unroll_log_jlog(E, eps, xi, Jr, Jt);
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    qw[k] = UD(w[k]);
    qd[k] = UD(d[k]);
  }
  unroll_jlog_apply(Jr, Jt, qw, a);
  unroll_jlog_apply(Jr, Jt, qd, c);
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    const UD bsum = xi[r] + c[r];
    phi_r[r] = UD(0.0) - UD(s[r] * s[r]) * (a[r] * bsum);
    x_r[r] = UD(s[r] * s[r]) * (xi[r] * xi[r]);
    if (a_out) a_out[r] = a[r].v;
    if (b_out) b_out[r] = bsum.v;
    if (xi_out) xi_out[r] = xi[r].v;
  }
  if (lam != 0.0) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      UD top(0.0), bot(0.0);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        top = top + Jr[3 * r + k] * Jr[3 * r + k] * UD(w[k] * d[k]) + Jt[3 * r + k] * Jt[3 * r + k] * UD(w[3 + k] * d[3 + k]);
        bot = bot + Jr[3 * r + k] * Jr[3 * r + k] * UD(w[3 + k] * d[3 + k]);
      }
      phi_r[r] = phi_r[r] - UD(lam * s[r] * s[r]) * top;
      phi_r[3 + r] = phi_r[3 + r] - UD(lam * s[3 + r] * s[3 + r]) * bot;
      if (ell_rows) {
        ell_rows[r] = top.v;
        ell_rows[3 + r] = bot.v;
      }
    }
  }
}

// SE3<UD> from values with the raw entry k (0..11, row major 3 x 4) seeded (k < 0: no seed)
__device__ __forceinline__ void unroll_seed(const SE3<double>& X, int k, SE3<UD>& Y) {
  const int kr = k >> 2, kc = k & 3;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) Y.R[3 * i + j] = UD(X.R[3 * i + j], (k >= 0 && i == kr && j == kc) ? 1.0 : 0.0);
    Y.t[i] = UD(X.t[i], (k >= 0 && i == kr && kc == 3) ? 1.0 : 0.0);
  }
}

// Robust combination (robust_cost_function.py:115-135: (J, e) <- sqrt(rho'(x) + eps) (J, e), NOT detached): with m_r = rho'(X_r) + eps
// the cost contributes Phi = sum_r m_r phi_r;  dPhi = sum_r [ m_r dphi_r + P_r m_x,r dx_r ],  P_r = phi_r (flatten_dims) | sum phi.
struct UnrollRobust {
  RobustTerms<6> rt;
  double P[6];
  __device__ __forceinline__ void setup(int loss, double log_radius, const UD* phi_r, const UD* x_r) {
    double xv[6], pv[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      xv[r] = x_r[r].v;
      pv[r] = phi_r[r].v;
    }
    rt.eval(loss, xv, log_radius);
    rt.group(pv, P);
  }
  __device__ __forceinline__ double dot(const UD* phi_r, const UD* x_r) const {
    double g = 0.0;
#pragma unroll
    for (int r = 0; r < 6; ++r) g += rt.m[r] * phi_r[r].d + P[r] * rt.m_x[r] * x_r[r].d;
    return g;
  }
  // weight gradients: phi_r = -s_r^2 (a_r b_r + lambda ell_r) and x_r = s_r^2 xi_r^2
  __device__ __forceinline__ void weights(const double* s, const UD* phi_r, const double* a, const double* b, const double* ell,
                                          double lam, const double* xi, double* gs, double* glr) const {
    double gl = 0.0;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      gs[r] = rt.m[r] * (-2.0 * s[r] * (a[r] * b[r] + lam * ell[r])) + P[r] * rt.m_x[r] * (2.0 * s[r] * xi[r] * xi[r]);
      gl += phi_r[r].v * rt.m_l[r];
    }
    *glr = gl;
  }
};

// Gradients of a Between cost's Phi: gXi, gXj, gZ (12 raw entries each), gs (6 weights), glr (log_loss_radius; 0 for a plain cost)
__device__ __forceinline__ void unroll_edge_vjp(const SE3<double>& Xi, const SE3<double>& Xj, const SE3<double>& Z, const double* s,
                                                const double* wi, const double* wj, const double* di, const double* dj,
                                                const Eps<double>& eps, double* gXi, double* gXj, double* gZ, double* gs,
                                                double lam = 0.0, int loss = THX_LOSS_NONE, double log_radius = 0.0,
                                                double* glr = nullptr) {
  const Eps<UD> epsd{UD(eps.nz), UD(eps.dnz), UD(eps.npi)};
  SE3<UD> A, Bv, C;
  UD phi_r[6], x_r[6];
  double xi[6], a[6], b[6], ell[6] = {0, 0, 0, 0, 0, 0}, gl = 0.0;
  unroll_seed(Xi, -1, A);
  unroll_seed(Xj, -1, Bv);
  unroll_seed(Z, -1, C);
  unroll_edge_phi(A, Bv, C, s, wi, wj, di, dj, epsd, lam, phi_r, x_r, a, b, ell, xi);
  UnrollRobust rb;
  rb.setup(loss, log_radius, phi_r, x_r);
  rb.weights(s, phi_r, a, b, ell, lam, xi, gs, &gl);
  if (glr) *glr = gl;
  for (int k = 0; k < 36; ++k) {   // run-time loop: one dual evaluation per raw entry of Xi, Xj, Z
    const int which = k / 12, e = k % 12;
    unroll_seed(Xi, which == 0 ? e : -1, A);
    unroll_seed(Xj, which == 1 ? e : -1, Bv);
    unroll_seed(Z, which == 2 ? e : -1, C);
    unroll_edge_phi(A, Bv, C, s, wi, wj, di, dj, epsd, lam, phi_r, x_r, nullptr, nullptr, nullptr, nullptr);
    const double g = rb.dot(phi_r, x_r);
    if (which == 0) gXi[e] = g;
    else if (which == 1) gXj[e] = g;
    else gZ[e] = g;
  }
}

__device__ __forceinline__ void unroll_prior_vjp(const SE3<double>& X, const SE3<double>& T, const double* s, const double* w,
                                                 const double* d, const Eps<double>& eps, double* gX, double* gT, double* gs,
                                                 double lam = 0.0, int loss = THX_LOSS_NONE, double log_radius = 0.0,
                                                 double* glr = nullptr) {
  const Eps<UD> epsd{UD(eps.nz), UD(eps.dnz), UD(eps.npi)};
  SE3<UD> A, Bv;
  UD phi_r[6], x_r[6];
  double xi[6], a[6], b[6], ell[6] = {0, 0, 0, 0, 0, 0}, gl = 0.0;
  unroll_seed(X, -1, A);
  unroll_seed(T, -1, Bv);
  unroll_prior_phi(A, Bv, s, w, d, epsd, lam, phi_r, x_r, a, b, ell, xi);
  UnrollRobust rb;
  rb.setup(loss, log_radius, phi_r, x_r);
  rb.weights(s, phi_r, a, b, ell, lam, xi, gs, &gl);
  if (glr) *glr = gl;
  for (int k = 0; k < 24; ++k) {
    const int which = k / 12, e = k % 12;
    unroll_seed(X, which == 0 ? e : -1, A);
    unroll_seed(T, which == 1 ? e : -1, Bv);
    unroll_prior_phi(A, Bv, s, w, d, epsd, lam, phi_r, x_r, nullptr, nullptr, nullptr, nullptr);
    const double g = rb.dot(phi_r, x_r);
    if (which == 0) gX[e] = g;
    else gT[e] = g;
  }
}

}  // namespace thx
