// The VPINN residual linear form, fused:
//     r_i = sum_T sum_q dx_q ( f(x_q) v_i(q) + s * grad v_i . g_q ),      s = -1 in the reference
// (examples/example_weak.py:64-75: `rhs(x, y) * v - v_grad @ gradient(points).mT` handed to
// integrate_linear_form, abstract_basis.py:95-112), and its adjoint in g and f -- the training
// step differentiates through it (example_weak.py:132-152).  The reference materialises the
// (N_T, Q, 3, 1) integrand with torch on every epoch; here one launch per direction reads g
// (N_T, Q, 2) and writes the element vectors (forward; summed per vertex by tfem_csr_gather, no
// atomics, the reference's accumulation order) or the cotangents of g and f (backward; one lane
// per element, nothing shared).  Operation order per element = the reference's: P1 gradients
// G @ inv(J) (element_tri.py:41), v_grad @ g^T summed over the two components in order, the
// integrand times dx_q summed over q in order.
#include <hip/hip_runtime.h>

#include <cstring>

#include "tfem_common.hpp"
#include "tfem_source.hpp"

namespace tfem {
namespace {

constexpr int kResBlock = 256;

template <typename T, typename I>
struct ResArgs {
  const T *coords;   // (n_verts, 2)
  const I *conn;     // (n_elems, 3)
  const T *fq;       // (n_elems, Q) source values, or NULL
  const T *flux;     // (n_elems, Q, 2), or NULL
  const T *cot;      // backward: (n_verts) cotangent of r
  T *out;            // forward: (3, n_elems) element vectors, entry-major
  T *grad_fq;        // backward: (n_elems, Q) or NULL
  T *grad_flux;      // backward: (n_elems, Q, 2) or NULL
  int64_t n_elems;
  T sign;
  T hw[kMaxQuad];
  T lam[3][kMaxQuad];
  SrcProgram<T> src;
};

template <typename T>
struct P1Geo {
  T x[3], y[3], det, vg[3][2];
};

// gather (abstract_mesh.py:257-262), J = X^T G (basis.py:87-88), signed det and (1/det) adj
// (element_tri.py:132-145), gradients G @ inv (element_tri.py:41)
template <typename T, typename I>
__device__ __forceinline__ void res_geometry(const ResArgs<T, I> &a, int64_t e, P1Geo<T> &g) {
  const I *c = a.conn + 3 * e;
  const int64_t v0 = c[0], v1 = c[1], v2 = c[2];
  g.x[0] = a.coords[2 * v0];
  g.y[0] = a.coords[2 * v0 + 1];
  g.x[1] = a.coords[2 * v1];
  g.y[1] = a.coords[2 * v1 + 1];
  g.x[2] = a.coords[2 * v2];
  g.y[2] = a.coords[2 * v2 + 1];
  const T ja = g.x[1] - g.x[0], jb = g.x[2] - g.x[0];
  const T jc = g.y[1] - g.y[0], jd = g.y[2] - g.y[0];
  g.det = ja * jd - jb * jc;
  const T r = T(1) / g.det;
  const T inv[2][2] = {{r * jd, r * (-jb)}, {r * (-jc), r * ja}};
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    g.vg[0][k] = (-inv[0][k]) + (-inv[1][k]);
    g.vg[1][k] = inv[0][k];
    g.vg[2][k] = inv[1][k];
  }
}

template <typename T, typename I, int Q, bool SRC>
__global__ __launch_bounds__(kResBlock) void k_p1_residual(const ResArgs<T, I> a) {
  const int64_t e = int64_t(blockIdx.x) * kResBlock + threadIdx.x;
  const int64_t ec = e < a.n_elems ? e : a.n_elems - 1;  // whole waves run the source program
  P1Geo<T> g;
  res_geometry(a, ec, g);
  T f[Q];
  if constexpr (SRC) {
    using Args = ResArgs<T, I>;
    const SrcLanes<T> prog = src_load_lanes<T>(src_in_kernarg<T>(__builtin_offsetof(Args, src)));
    T xq[Q], yq[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {  // bar^T @ X (basis.py:90-91)
      xq[q] = (a.lam[0][q] * g.x[0] + a.lam[1][q] * g.x[1]) + a.lam[2][q] * g.x[2];
      yq[q] = (a.lam[0][q] * g.y[0] + a.lam[1][q] * g.y[1]) + a.lam[2][q] * g.y[2];
    }
    src_run<T, Q>(prog, xq, yq, f);
  } else {
#pragma unroll
    for (int q = 0; q < Q; ++q) f[q] = a.fq ? a.fq[Q * ec + q] : T(0);
  }
  if (e >= a.n_elems) return;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    T acc = T(0);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      T term = f[q] * a.lam[i][q];
      if (a.flux) {
        const T *gq = a.flux + 2 * (Q * e + q);
        term = term + a.sign * (g.vg[i][0] * gq[0] + g.vg[i][1] * gq[1]);
      }
      acc = acc + term * (a.hw[q] * g.det);
    }
    a.out[int64_t(i) * a.n_elems + e] = acc;
  }
}

template <typename T, typename I, int Q>
__global__ __launch_bounds__(kResBlock) void k_p1_residual_backward(const ResArgs<T, I> a) {
  const int64_t e = int64_t(blockIdx.x) * kResBlock + threadIdx.x;
  if (e >= a.n_elems) return;
  P1Geo<T> g;
  res_geometry(a, e, g);
  const I *c = a.conn + 3 * e;
  const T ct[3] = {a.cot[int64_t(c[0])], a.cot[int64_t(c[1])], a.cot[int64_t(c[2])]};
  // d r_i / d g_{q,k} = s dx_q vg_ik ;  d r_i / d f_q = dx_q l_i(q)
  const T gx = (ct[0] * g.vg[0][0] + ct[1] * g.vg[1][0]) + ct[2] * g.vg[2][0];
  const T gy = (ct[0] * g.vg[0][1] + ct[1] * g.vg[1][1]) + ct[2] * g.vg[2][1];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const T dx = a.hw[q] * g.det;
    if (a.grad_flux) {
      a.grad_flux[2 * (Q * e + q)] = a.sign * (dx * gx);
      a.grad_flux[2 * (Q * e + q) + 1] = a.sign * (dx * gy);
    }
    if (a.grad_fq)
      a.grad_fq[Q * e + q] = dx * ((ct[0] * a.lam[0][q] + ct[1] * a.lam[1][q]) + ct[2] * a.lam[2][q]);
  }
}

template <typename T, typename I>
int fill_args(ResArgs<T, I> &a, const void *coords, const void *conn, int64_t n_elems, int quad_order,
              double sign, int *nq) {
  TriTables tables;
  if (!build_tri_tables(quad_order, int(sizeof(T)), &tables))
    return fail(TFEM_ERR_UNSUPPORTED, "Integration order not implemented");
  std::memset(&a, 0, sizeof(a));
  a.coords = static_cast<const T *>(coords);
  a.conn = static_cast<const I *>(conn);
  a.n_elems = n_elems;
  a.sign = T(sign);
  for (int q = 0; q < tables.nq; ++q) {
    a.hw[q] = T(tables.hw[q]);
    for (int i = 0; i < 3; ++i) a.lam[i][q] = T(tables.lam[q][i]);
  }
  *nq = tables.nq;
  return TFEM_OK;
}

#define TFEM_RES_DISPATCH_Q(nq, CALL)                       \
  switch (nq) {                                             \
    case 1: { constexpr int Q = 1; CALL; break; }           \
    case 3: { constexpr int Q = 3; CALL; break; }           \
    case 4: { constexpr int Q = 4; CALL; break; }           \
    case 6: { constexpr int Q = 6; CALL; break; }           \
    default: return fail(TFEM_ERR_UNSUPPORTED, "Integration order not implemented"); \
  }

template <typename T, typename I>
int run_forward(const void *coords, const void *conn, int64_t n_elems, int quad_order, const void *fq,
                const tfem_source_program *source, const void *flux, double sign, void *out,
                hipStream_t stream) {
  ResArgs<T, I> a;
  int nq = 0;
  if (int st = fill_args(a, coords, conn, n_elems, quad_order, sign, &nq)) return st;
  a.fq = static_cast<const T *>(fq);
  a.flux = static_cast<const T *>(flux);
  a.out = static_cast<T *>(out);
  if (source) {
    if (int st = src_convert<T>(source, &a.src)) return st;
  }
  if (n_elems == 0) return TFEM_OK;
  const dim3 grid(unsigned((n_elems + kResBlock - 1) / kResBlock)), block(kResBlock);
  if (source) {
    TFEM_RES_DISPATCH_Q(nq, hipLaunchKernelGGL((k_p1_residual<T, I, Q, true>), grid, block, 0, stream, a));
  } else {
    TFEM_RES_DISPATCH_Q(nq, hipLaunchKernelGGL((k_p1_residual<T, I, Q, false>), grid, block, 0, stream, a));
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(TFEM_ERR_HIP, "residual launch: %s", hipGetErrorString(e));
  return TFEM_OK;
}

template <typename T, typename I>
int run_backward(const void *coords, const void *conn, int64_t n_elems, int quad_order, const void *cot,
                 double sign, void *grad_fq, void *grad_flux, hipStream_t stream) {
  ResArgs<T, I> a;
  int nq = 0;
  if (int st = fill_args(a, coords, conn, n_elems, quad_order, sign, &nq)) return st;
  a.cot = static_cast<const T *>(cot);
  a.grad_fq = static_cast<T *>(grad_fq);
  a.grad_flux = static_cast<T *>(grad_flux);
  if (n_elems == 0) return TFEM_OK;
  const dim3 grid(unsigned((n_elems + kResBlock - 1) / kResBlock)), block(kResBlock);
  TFEM_RES_DISPATCH_Q(nq, hipLaunchKernelGGL((k_p1_residual_backward<T, I, Q>), grid, block, 0, stream, a));
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(TFEM_ERR_HIP, "residual backward launch: %s", hipGetErrorString(e));
  return TFEM_OK;
}

}  // namespace
}  // namespace tfem

extern "C" {

int tfem_p1_residual_local(const void *coords, int real_bytes, const void *conn, int idx_bytes,
                           int64_t n_elems, int64_t n_verts, int quad_order, const void *fq,
                           const tfem_source_program *source, const void *flux, double flux_sign,
                           void *out_local, void *stream) {
  using namespace tfem;
  if ((real_bytes != 4 && real_bytes != 8) || (idx_bytes != 4 && idx_bytes != 8))
    return fail(TFEM_ERR_INVALID_ARGUMENT, "real_bytes / idx_bytes must be 4 or 8");
  if (n_elems < 0 || n_verts < 0 || (n_elems > 0 && (!coords || !conn || !out_local)))
    return fail(TFEM_ERR_INVALID_ARGUMENT, "bad arguments");
  if (fq && source) return fail(TFEM_ERR_INVALID_ARGUMENT, "source values AND a source program");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (real_bytes == 8)
    return idx_bytes == 4 ? run_forward<double, int32_t>(coords, conn, n_elems, quad_order, fq, source, flux, flux_sign, out_local, s)
                          : run_forward<double, int64_t>(coords, conn, n_elems, quad_order, fq, source, flux, flux_sign, out_local, s);
  return idx_bytes == 4 ? run_forward<float, int32_t>(coords, conn, n_elems, quad_order, fq, source, flux, flux_sign, out_local, s)
                        : run_forward<float, int64_t>(coords, conn, n_elems, quad_order, fq, source, flux, flux_sign, out_local, s);
}

int tfem_p1_residual_backward(const void *coords, int real_bytes, const void *conn, int idx_bytes,
                              int64_t n_elems, int64_t n_verts, int quad_order, const void *cotangent,
                              double flux_sign, void *grad_fq, void *grad_flux, void *stream) {
  using namespace tfem;
  if ((real_bytes != 4 && real_bytes != 8) || (idx_bytes != 4 && idx_bytes != 8))
    return fail(TFEM_ERR_INVALID_ARGUMENT, "real_bytes / idx_bytes must be 4 or 8");
  if (n_elems < 0 || n_verts < 0 || (n_elems > 0 && (!coords || !conn || !cotangent)))
    return fail(TFEM_ERR_INVALID_ARGUMENT, "bad arguments");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (real_bytes == 8)
    return idx_bytes == 4 ? run_backward<double, int32_t>(coords, conn, n_elems, quad_order, cotangent, flux_sign, grad_fq, grad_flux, s)
                          : run_backward<double, int64_t>(coords, conn, n_elems, quad_order, cotangent, flux_sign, grad_fq, grad_flux, s);
  return idx_bytes == 4 ? run_backward<float, int32_t>(coords, conn, n_elems, quad_order, cotangent, flux_sign, grad_fq, grad_flux, s)
                        : run_backward<float, int64_t>(coords, conn, n_elems, quad_order, cotangent, flux_sign, grad_fq, grad_flux, s);
}

}  // extern "C"
