// Generic block assembly: H = A^T A (lower triangle) and g = A^T b from per-cost Jacobian blocks, for ANY cost
// function whose (weighted) Jacobians and errors are supplied as tensors -- the general case of
// DenseLinearization._linearize_jacobian_impl + _linearize_hessian_impl
// (theseus/optimizer/dense_linearization.py:29-62) without ever forming the dense A.  The pose-graph kernels in
// pg_kernels.hip are the fused special case; this path serves everything else (AutoDiffCostFunction, user costs).
//
// Deterministic: contributions are grouped by target block on the host (CSR), one lane owns one output element of
// one problem and sums its contributions in a fixed order -- no atomics.  Sums are carried in fp64 registers.
#include "common.cuh"

namespace thx {

template <typename T>
__global__ void __launch_bounds__(64)
block_hessian_kernel(const thx_block_target* __restrict__ targets, const thx_block_term* __restrict__ terms,
                     const int32_t* __restrict__ elem2target, int n_elems, T* __restrict__ H, int64_t ld, int B) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  const int el = blockIdx.z * gridDim.y + blockIdx.y;  // the element index is folded over (y, z): grid.y <= 65535
  if (b >= B || el >= n_elems) return;
  const thx_block_target tg = targets[elem2target[el]];
  const int q = el - tg.elem_begin, i = q / tg.dof_b, j = q % tg.dof_b;
  if (tg.row0 == tg.col0 && j > i) return;  // diagonal block: lower triangle only
  double acc = 0.0;
  for (int t = tg.term_begin; t < tg.term_end; ++t) {
    const thx_block_term tm = terms[t];
    const T* Ja = static_cast<const T*>(tm.Ja) + (int64_t)b * tm.bstride_a + i;
    const T* Jb = static_cast<const T*>(tm.Jb) + (int64_t)b * tm.bstride_b + j;
    for (int r = 0; r < tm.dim; ++r) acc += (double)Ja[(int64_t)r * tm.dof_a] * (double)Jb[(int64_t)r * tm.dof_b];
  }
  H[(int64_t)b * ld * ld + (int64_t)(tg.row0 + i) * ld + tg.col0 + j] = (T)acc;
}

template <typename T>
__global__ void __launch_bounds__(64)
block_gradient_kernel(const thx_grad_target* __restrict__ targets, const thx_grad_term* __restrict__ terms,
                      const int32_t* __restrict__ elem2target, int n_elems, T* __restrict__ g, int64_t ldg, int B) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  const int el = blockIdx.z * gridDim.y + blockIdx.y;
  if (b >= B || el >= n_elems) return;
  const thx_grad_target tg = targets[elem2target[el]];
  const int i = el - tg.elem_begin;
  double acc = 0.0;
  for (int t = tg.term_begin; t < tg.term_end; ++t) {
    const thx_grad_term tm = terms[t];
    const T* J = static_cast<const T*>(tm.J) + (int64_t)b * tm.bstride_j + i;
    const T* e = static_cast<const T*>(tm.e) + (int64_t)b * tm.bstride_e;
    for (int r = 0; r < tm.dim; ++r) acc += (double)J[(int64_t)r * tm.dof] * (double)e[r];
  }
  g[(int64_t)b * ldg + tg.col0 + i] = (T)(-acc);  // Atb = A^T b with b = -err (dense_linearization.py:55)
}

}  // namespace thx

using namespace thx;

extern "C" int thx_block_assemble(const thx_block_target* h_targets, const thx_block_term* h_terms,
                                  const int32_t* h_elem2target, int32_t n_h_elems, const thx_grad_target* g_targets,
                                  const thx_grad_term* g_terms, const int32_t* g_elem2target, int32_t n_g_elems,
                                  void* H, int64_t ld, void* g, int64_t ldg, int32_t B, int dtype, void* stream) {
  if (B <= 0) return fail("thx_block_assemble: empty batch");
  if (n_h_elems > 0 && (!h_targets || !h_terms || !h_elem2target || !H)) return fail("thx_block_assemble: null H table");
  if (n_g_elems > 0 && (!g_targets || !g_terms || !g_elem2target || !g)) return fail("thx_block_assemble: null g table");
  dim3 block(64);
  const unsigned gx = (B + 63) / 64;
  auto grid_for = [&](int n) { return dim3(gx, n < 65535 ? n : 65535, (n + 65534) / 65535); };
  THX_DISPATCH(dtype,
               {
                 if (n_h_elems > 0)
                   hipLaunchKernelGGL(block_hessian_kernel<float>, grid_for(n_h_elems), block, 0, as_stream(stream),
                                      h_targets, h_terms, h_elem2target, n_h_elems, (float*)H, ld, B);
                 if (n_g_elems > 0)
                   hipLaunchKernelGGL(block_gradient_kernel<float>, grid_for(n_g_elems), block, 0, as_stream(stream),
                                      g_targets, g_terms, g_elem2target, n_g_elems, (float*)g, ldg, B);
               },
               {
                 if (n_h_elems > 0)
                   hipLaunchKernelGGL(block_hessian_kernel<double>, grid_for(n_h_elems), block, 0, as_stream(stream),
                                      h_targets, h_terms, h_elem2target, n_h_elems, (double*)H, ld, B);
                 if (n_g_elems > 0)
                   hipLaunchKernelGGL(block_gradient_kernel<double>, grid_for(n_g_elems), block, 0, as_stream(stream),
                                      g_targets, g_terms, g_elem2target, n_g_elems, (double*)g, ldg, B);
               });
  return check_launch("thx_block_assemble");
}
