// Device side of libtfem_hip (gfx950 / CDNA4): element kernels and their C entry points.
//
// Baseline ("v1") numeric phase: one lane per element, gather of the three vertex
// coordinates fused into the kernel, per-element arithmetic in the reference's operation
// order, scatter through the precomputed int32 slot map with hardware fp64 atomics.
// The tile-plan kernels (tfem_tiles.hip) replace the scatter for the P1 headline path.
#include <hip/hip_runtime.h>

#include <cstring>

#include "tfem_common.hpp"

// The reference evaluates every product and sum as a separate rounded torch op; keep the
// compiler from fusing them into FMAs so the results track it to the last bits.
#pragma clang fp contract(off)

namespace tfem {

#define TFEM_HIP_CHECK(expr)                                                         \
  do {                                                                               \
    hipError_t err__ = (expr);                                                       \
    if (err__ != hipSuccess)                                                         \
      return fail(TFEM_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(err__));   \
  } while (0)

constexpr int kBlock = 256;

static inline unsigned blocks_for(int64_t work) { return unsigned((work + kBlock - 1) / kBlock); }

template <typename T>
__device__ __forceinline__ void atomic_add(T *address, T value) {
  unsafeAtomicAdd(address, value);  // global_atomic_add_f64 / _f32, no CAS loop
}

// ---------------------------------------------------------------------------------------
// kernel arguments
// ---------------------------------------------------------------------------------------
template <typename T, typename I>
struct TriArgs {
  const T *coords;      // (n_verts, 2)   [fracture: (F, coords_per_frac, 2)]
  const I *conn_geo;    // (n_elems, 3)
  int64_t n_elems;
  int64_t elems_per_frac;   // n_elems when not a fracture mesh
  int64_t coords_per_frac;
  const T *frac_pinv;   // (F, 2, 3) or nullptr
  const T *frac_det;    // (F) or nullptr
  T alpha, beta;
  const int32_t *slots;
  T *vals;
  const I *conn_dof;
  const T *fq;
  T *out;
  T *o_vgrad, *o_dx, *o_points, *o_inv;
  T hw[kMaxQuad];
  T lam[kMaxQuad][3];
};

template <typename T>
struct P2Tables {
  T phi[kMaxQuad][6];
  T rgrad[kMaxQuad][6][2];
};

template <typename T, typename I>
static void fill_common(TriArgs<T, I> &a, const TriTables &t) {
  for (int q = 0; q < kMaxQuad; ++q) {
    a.hw[q] = T(t.hw[q]);
    for (int i = 0; i < 3; ++i) a.lam[q][i] = T(t.lam[q][i]);
  }
}

// ---------------------------------------------------------------------------------------
// per-element geometry, reference operation order
// ---------------------------------------------------------------------------------------
template <typename T>
struct ElemGeo {
  T x[3], y[3];
  T det;              // signed, element_tri.py:139
  T inv[2][2];        // (1/det) * adj, element_tri.py:141-143
  T fdet;             // fracture area factor (1 when not a fracture)
};

template <typename T, typename I>
__device__ __forceinline__ void load_element(const TriArgs<T, I> &a, int64_t e, ElemGeo<T> &g,
                                             int64_t &frac) {
  const I *c = a.conn_geo + 3 * e;
  frac = e / a.elems_per_frac;
  const T *base = a.coords + 2 * frac * a.coords_per_frac;
  const int64_t v0 = c[0], v1 = c[1], v2 = c[2];
  g.x[0] = base[2 * v0];
  g.y[0] = base[2 * v0 + 1];
  g.x[1] = base[2 * v1];
  g.y[1] = base[2 * v1 + 1];
  g.x[2] = base[2 * v2];
  g.y[2] = base[2 * v2 + 1];
  // J = X^T G (basis.py:87-88): -x0 + x1 + 0*x2 rounds exactly like x1 - x0
  const T ja = g.x[1] - g.x[0], jb = g.x[2] - g.x[0];
  const T jc = g.y[1] - g.y[0], jd = g.y[2] - g.y[0];
  g.det = ja * jd - jb * jc;
  const T r = T(1) / g.det;
  g.inv[0][0] = r * jd;
  g.inv[0][1] = r * (-jb);
  g.inv[1][0] = r * (-jc);
  g.inv[1][1] = r * ja;
  g.fdet = a.frac_det ? a.frac_det[frac] : T(1);
}

// P1 physical gradients G @ inv (element_tri.py:41), optionally @ pinv_F (fracture_basis.py:20-22)
template <typename T, typename I, int D>
__device__ __forceinline__ void p1_gradients(const TriArgs<T, I> &a, const ElemGeo<T> &g,
                                             int64_t frac, T (&grad)[3][D]) {
  T g2[3][2];
  for (int c = 0; c < 2; ++c) {
    g2[0][c] = (-g.inv[0][c]) + (-g.inv[1][c]);
    g2[1][c] = g.inv[0][c];
    g2[2][c] = g.inv[1][c];
  }
  if constexpr (D == 2) {
    for (int i = 0; i < 3; ++i)
      for (int c = 0; c < 2; ++c) grad[i][c] = g2[i][c];
  } else {
    const T *p = a.frac_pinv + 6 * frac;  // (2,3) row-major
    for (int i = 0; i < 3; ++i)
      for (int c = 0; c < 3; ++c) grad[i][c] = g2[i][0] * p[c] + g2[i][1] * p[3 + c];
  }
}

template <typename T, int D>
__device__ __forceinline__ T dot(const T (&u)[D], const T (&v)[D]) {
  T s = u[0] * v[0];
  for (int c = 1; c < D; ++c) s = s + u[c] * v[c];
  return s;
}

// ---------------------------------------------------------------------------------------
// P1 bilinear:  sum_q (alpha * gi.gj + beta * li lj) * dx_q   -> CSR (atomic scatter)
// ---------------------------------------------------------------------------------------
template <typename T, typename I, int Q, int D>
__global__ __launch_bounds__(kBlock) void k_p1_bilinear_atomic(const TriArgs<T, I> a) {
  const int64_t e = int64_t(blockIdx.x) * kBlock + threadIdx.x;
  if (e >= a.n_elems) return;
  ElemGeo<T> g;
  int64_t frac;
  load_element(a, e, g, frac);
  T grad[3][D];
  p1_gradients<T, I, D>(a, g, frac, grad);
  T dx[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    dx[q] = a.hw[q] * g.det;
    if constexpr (D == 3) dx[q] = dx[q] * g.fdet;
  }
  T loc[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = i; j < 3; ++j) {
      const T s = a.alpha * dot<T, D>(grad[i], grad[j]);
      T acc = T(0);
#pragma unroll
      for (int q = 0; q < Q; ++q) acc = acc + (s + a.beta * (a.lam[q][i] * a.lam[q][j])) * dx[q];
      loc[i][j] = acc;
      loc[j][i] = acc;  // bitwise symmetric: same products, same order
    }
  }
  if (!a.slots) {  // local blocks, entry-major (9, n_elems): lanes store contiguously
#pragma unroll
    for (int k = 0; k < 9; ++k) a.vals[int64_t(k) * a.n_elems + e] = loc[k / 3][k % 3];
    return;
  }
  const int32_t *s = a.slots + 9 * e;
#pragma unroll
  for (int k = 0; k < 9; ++k) atomic_add(a.vals + s[k], loc[k / 3][k % 3]);
}

// ---------------------------------------------------------------------------------------
// P2 bilinear (6x6 blocks); reference-element tables staged in LDS
// ---------------------------------------------------------------------------------------
template <typename T, typename I, int Q>
__global__ __launch_bounds__(kBlock) void k_p2_bilinear_atomic(const TriArgs<T, I> a,
                                                                const P2Tables<T> tab) {
  __shared__ T s_phi[Q][6];
  __shared__ T s_rg[Q][6][2];
  for (int t = threadIdx.x; t < Q * 6; t += kBlock) {
    s_phi[t / 6][t % 6] = tab.phi[t / 6][t % 6];
    s_rg[t / 6][t % 6][0] = tab.rgrad[t / 6][t % 6][0];
    s_rg[t / 6][t % 6][1] = tab.rgrad[t / 6][t % 6][1];
  }
  __syncthreads();
  const int64_t e = int64_t(blockIdx.x) * kBlock + threadIdx.x;
  if (e >= a.n_elems) return;
  ElemGeo<T> g;
  int64_t frac;
  load_element(a, e, g, frac);
  T acc[21];
#pragma unroll
  for (int k = 0; k < 21; ++k) acc[k] = T(0);
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const T dxq = a.hw[q] * g.det;
    T vg[6][2];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      vg[i][0] = s_rg[q][i][0] * g.inv[0][0] + s_rg[q][i][1] * g.inv[1][0];
      vg[i][1] = s_rg[q][i][0] * g.inv[0][1] + s_rg[q][i][1] * g.inv[1][1];
    }
    int k = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
      for (int j = i; j < 6; ++j, ++k) {
        const T term = a.alpha * dot<T, 2>(vg[i], vg[j]) + a.beta * (s_phi[q][i] * s_phi[q][j]);
        acc[k] = acc[k] + term * dxq;
      }
    }
  }
  if (!a.slots) {  // local blocks, entry-major (36, n_elems)
    int k = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
      for (int j = i; j < 6; ++j, ++k) {
        a.vals[int64_t(6 * i + j) * a.n_elems + e] = acc[k];
        if (j != i) a.vals[int64_t(6 * j + i) * a.n_elems + e] = acc[k];
      }
    }
    return;
  }
  const int32_t *s = a.slots + 36 * e;
  int k = 0;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
#pragma unroll
    for (int j = i; j < 6; ++j, ++k) {
      atomic_add(a.vals + s[6 * i + j], acc[k]);
      if (j != i) atomic_add(a.vals + s[6 * j + i], acc[k]);
    }
  }
}

// ---------------------------------------------------------------------------------------
// load vector:  sum_q (f_q * phi_i(q)) * dx_q  -> out[conn_dof]
// ---------------------------------------------------------------------------------------
template <typename T, typename I, int Q, int N>
__global__ __launch_bounds__(kBlock) void k_load_atomic(const TriArgs<T, I> a,
                                                         const P2Tables<T> tab) {
  const int64_t e = int64_t(blockIdx.x) * kBlock + threadIdx.x;
  if (e >= a.n_elems) return;
  ElemGeo<T> g;
  int64_t frac;
  load_element(a, e, g, frac);
  T w[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    T dxq = a.hw[q] * g.det;
    if (a.frac_det) dxq = dxq * g.fdet;
    w[q] = dxq;
  }
  const T *f = a.fq + Q * e;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    T acc = T(0);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const T phi = (N == 3) ? a.lam[q][i] : tab.phi[q][i];
      acc = acc + (f[q] * phi) * w[q];
    }
    if (!a.conn_dof)
      a.out[int64_t(i) * a.n_elems + e] = acc;  // local vectors, entry-major (N, n_elems)
    else
      atomic_add(a.out + int64_t(a.conn_dof[N * e + i]), acc);
  }
}

// ---------------------------------------------------------------------------------------
// geometry cache
// ---------------------------------------------------------------------------------------
template <typename T, typename I, int Q, int P>
__global__ __launch_bounds__(kBlock) void k_geometry(const TriArgs<T, I> a,
                                                      const P2Tables<T> tab) {
  const int64_t e = int64_t(blockIdx.x) * kBlock + threadIdx.x;
  if (e >= a.n_elems) return;
  ElemGeo<T> g;
  int64_t frac;
  load_element(a, e, g, frac);
  if (a.o_inv) {
    T *o = a.o_inv + 4 * e;
    o[0] = g.inv[0][0];
    o[1] = g.inv[0][1];
    o[2] = g.inv[1][0];
    o[3] = g.inv[1][1];
  }
  if (a.o_dx) {
#pragma unroll
    for (int q = 0; q < Q; ++q) a.o_dx[Q * e + q] = a.hw[q] * g.det;
  }
  if (a.o_points) {
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      // bar^T @ X (basis.py:90-91): K = 3 accumulation in order
      a.o_points[2 * (Q * e + q)] =
          (a.lam[q][0] * g.x[0] + a.lam[q][1] * g.x[1]) + a.lam[q][2] * g.x[2];
      a.o_points[2 * (Q * e + q) + 1] =
          (a.lam[q][0] * g.y[0] + a.lam[q][1] * g.y[1]) + a.lam[q][2] * g.y[2];
    }
  }
  if (a.o_vgrad) {
    if constexpr (P == 1) {
      T grad[3][2];
      p1_gradients<T, I, 2>(a, g, frac, grad);
      T *o = a.o_vgrad + 6 * e;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        o[2 * i] = grad[i][0];
        o[2 * i + 1] = grad[i][1];
      }
    } else {
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        T *o = a.o_vgrad + 12 * (Q * e + q);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          o[2 * i] = tab.rgrad[q][i][0] * g.inv[0][0] + tab.rgrad[q][i][1] * g.inv[1][0];
          o[2 * i + 1] = tab.rgrad[q][i][0] * g.inv[0][1] + tab.rgrad[q][i][1] * g.inv[1][1];
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// generic quadrature-reduce + scatter
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void k_reduce_bilinear(const T *integrand, int64_t es,
                                                             int64_t qs, const T *dx,
                                                             int64_t n_entries, int nq, int nn,
                                                             const int32_t *slots, T *vals) {
  const int64_t idx = int64_t(blockIdx.x) * kBlock + threadIdx.x;
  if (idx >= n_entries) return;
  const int64_t e = idx / nn;
  const int k = int(idx - e * nn);
  const T *p = integrand + e * es + k;
  const T *w = dx + e * nq;
  T acc = T(0);
  for (int q = 0; q < nq; ++q) acc = acc + p[q * qs] * w[q];
  if (!slots)
    vals[int64_t(k) * (n_entries / nn) + e] = acc;  // local blocks, entry-major (nn, n_elems)
  else
    atomic_add(vals + slots[idx], acc);
}

template <typename T, typename I>
__global__ __launch_bounds__(kBlock) void k_reduce_linear(const T *integrand, int64_t es,
                                                           int64_t qs, const T *dx,
                                                           int64_t n_entries, int nq, int n,
                                                           const I *conn_dof, T *out) {
  const int64_t idx = int64_t(blockIdx.x) * kBlock + threadIdx.x;
  if (idx >= n_entries) return;
  const int64_t e = idx / n;
  const int k = int(idx - e * n);
  const T *p = integrand + e * es + k;
  const T *w = dx + e * nq;
  T acc = T(0);
  for (int q = 0; q < nq; ++q) acc = acc + p[q * qs] * w[q];
  if (!conn_dof)
    out[int64_t(k) * (n_entries / n) + e] = acc;  // local vectors, entry-major (n, n_elems)
  else
    atomic_add(out + int64_t(conn_dof[idx]), acc);
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_reduce_functional(const T *integrand, int64_t es,
                                                               int64_t qs, const T *dx,
                                                               int64_t n_elems, int nq, int n_inner,
                                                               T *out) {
  const int64_t e = int64_t(blockIdx.x) * kBlock + threadIdx.x;
  if (e >= n_elems) return;
  const T *w = dx + e * nq;
  T total = T(0);
  for (int k = 0; k < n_inner; ++k) {  // .sum(-3) then .sum(-2), abstract_basis.py:72
    const T *p = integrand + e * es + k;
    T acc = T(0);
    for (int q = 0; q < nq; ++q) acc = acc + p[q * qs] * w[q];
    total = total + acc;
  }
  out[e] = total;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_csr_to_dense(const int64_t *rowptr,
                                                          const int32_t *colind, const T *vals,
                                                          int64_t n, T *dense) {
  const int64_t r = int64_t(blockIdx.x) * kBlock + threadIdx.x;
  if (r >= n) return;
  for (int64_t k = rowptr[r]; k < rowptr[r + 1]; ++k) dense[r * n + colind[k]] = vals[k];
}

// ---------------------------------------------------------------------------------------
// launch helpers
// ---------------------------------------------------------------------------------------
static int check_common(int real_bytes, int idx_bytes, int64_t n_elems) {
  if (real_bytes != 4 && real_bytes != 8)
    return fail(TFEM_ERR_INVALID_ARGUMENT, "real_bytes must be 4 or 8, got %d", real_bytes);
  if (idx_bytes != 4 && idx_bytes != 8)
    return fail(TFEM_ERR_INVALID_ARGUMENT, "idx_bytes must be 4 or 8, got %d", idx_bytes);
  if (n_elems < 0) return fail(TFEM_ERR_INVALID_ARGUMENT, "n_elems < 0");
  return TFEM_OK;
}

template <typename T>
static void fill_p2(P2Tables<T> &p, const TriTables &t) {
  for (int q = 0; q < kMaxQuad; ++q)
    for (int i = 0; i < 6; ++i) {
      p.phi[q][i] = T(t.phi2[q][i]);
      p.rgrad[q][i][0] = T(t.rgrad2[q][i][0]);
      p.rgrad[q][i][1] = T(t.rgrad2[q][i][1]);
    }
}

struct FracSpec {
  const void *pinv;
  const void *det;
  int n_fractures;
  int64_t coords_per_fracture;
};

template <typename T, typename I>
static int setup_args(TriArgs<T, I> &a, const void *coords, const void *conn_geo, int64_t n_elems,
                      int64_t n_verts, int quad_order, const FracSpec &fr, TriTables &tables) {
  if (!build_tri_tables(quad_order, int(sizeof(T)), &tables))
    return fail(TFEM_ERR_UNSUPPORTED, "Integration order not implemented");
  if (n_elems > 0 && (!coords || !conn_geo))
    return fail(TFEM_ERR_INVALID_ARGUMENT, "coords / conn is NULL");
  std::memset(&a, 0, sizeof(a));
  a.coords = static_cast<const T *>(coords);
  a.conn_geo = static_cast<const I *>(conn_geo);
  a.n_elems = n_elems;
  a.elems_per_frac = n_elems > 0 ? n_elems : 1;
  a.coords_per_frac = n_verts;
  if (fr.n_fractures > 0 && (fr.pinv || fr.det)) {
    if (n_elems % fr.n_fractures != 0)
      return fail(TFEM_ERR_INVALID_ARGUMENT, "n_elems %lld not divisible by n_fractures %d",
                  (long long)n_elems, fr.n_fractures);
    a.elems_per_frac = n_elems > 0 ? n_elems / fr.n_fractures : 1;
    a.coords_per_frac = fr.coords_per_fracture;
    a.frac_pinv = static_cast<const T *>(fr.pinv);
    a.frac_det = static_cast<const T *>(fr.det);
  }
  fill_common(a, tables);
  return TFEM_OK;
}

#define TFEM_DISPATCH_Q(NQ, ...)                                         \
  switch (NQ) {                                                          \
    case 1: { constexpr int Q = 1; __VA_ARGS__; } break;                 \
    case 3: { constexpr int Q = 3; __VA_ARGS__; } break;                 \
    case 4: { constexpr int Q = 4; __VA_ARGS__; } break;                 \
    case 6: { constexpr int Q = 6; __VA_ARGS__; } break;                 \
    default: return fail(TFEM_ERR_UNSUPPORTED, "Integration order not implemented"); \
  }

template <typename T, typename I>
static int run_bilinear(const void *coords, const void *conn_geo, int64_t n_elems,
                        int64_t n_verts, int poly_order, int quad_order, double alpha,
                        double beta, const int32_t *slots, void *vals, int64_t nnz,
                        const FracSpec &fr, hipStream_t stream) {
  TriArgs<T, I> a;
  TriTables tables;
  if (int st = setup_args(a, coords, conn_geo, n_elems, n_verts, quad_order, fr, tables))
    return st;
  if (nnz > 0 && !vals) return fail(TFEM_ERR_INVALID_ARGUMENT, "vals is NULL");
  const int nn = poly_order == 2 ? 36 : 9;
  if (!slots && nnz != nn * n_elems)
    return fail(TFEM_ERR_INVALID_ARGUMENT, "local-block mode: vals must hold %d * n_elems entries", nn);
  a.alpha = T(alpha);
  a.beta = T(beta);
  a.slots = slots;
  a.vals = static_cast<T *>(vals);
  if (slots && nnz > 0) TFEM_HIP_CHECK(hipMemsetAsync(vals, 0, size_t(nnz) * sizeof(T), stream));
  if (n_elems == 0) return TFEM_OK;
  const dim3 grid(blocks_for(n_elems)), block(kBlock);
  if (poly_order == 1) {
    if (a.frac_pinv) {
      TFEM_DISPATCH_Q(tables.nq, hipLaunchKernelGGL((k_p1_bilinear_atomic<T, I, Q, 3>), grid,
                                                     block, 0, stream, a));
    } else {
      TFEM_DISPATCH_Q(tables.nq, hipLaunchKernelGGL((k_p1_bilinear_atomic<T, I, Q, 2>), grid,
                                                     block, 0, stream, a));
    }
  } else if (poly_order == 2) {
    if (a.frac_pinv) return fail(TFEM_ERR_UNSUPPORTED, "P2 on fractures not implemented");
    P2Tables<T> p2;
    fill_p2(p2, tables);
    TFEM_DISPATCH_Q(tables.nq, hipLaunchKernelGGL((k_p2_bilinear_atomic<T, I, Q>), grid, block,
                                                   0, stream, a, p2));
  } else {
    return fail(TFEM_ERR_UNSUPPORTED, "Polynomial order not implemented");
  }
  TFEM_HIP_CHECK(hipGetLastError());
  return TFEM_OK;
}

template <typename T, typename I>
static int run_load(const void *coords, const void *conn_geo, const void *conn_dof,
                    int64_t n_elems, int64_t n_verts, int poly_order, int quad_order,
                    const void *fq, void *out, int64_t n_dofs, const FracSpec &fr,
                    hipStream_t stream) {
  TriArgs<T, I> a;
  TriTables tables;
  if (int st = setup_args(a, coords, conn_geo, n_elems, n_verts, quad_order, fr, tables))
    return st;
  if (n_dofs > 0 && !out) return fail(TFEM_ERR_INVALID_ARGUMENT, "out is NULL");
  if (n_elems > 0 && !fq) return fail(TFEM_ERR_INVALID_ARGUMENT, "fq is NULL");
  const int n_local = poly_order == 2 ? 6 : 3;
  if (!conn_dof && n_dofs != n_local * n_elems)
    return fail(TFEM_ERR_INVALID_ARGUMENT, "local-vector mode: out must hold %d * n_elems entries", n_local);
  a.conn_dof = static_cast<const I *>(conn_dof);
  a.fq = static_cast<const T *>(fq);
  a.out = static_cast<T *>(out);
  if (conn_dof && n_dofs > 0) TFEM_HIP_CHECK(hipMemsetAsync(out, 0, size_t(n_dofs) * sizeof(T), stream));
  if (n_elems == 0) return TFEM_OK;
  P2Tables<T> p2;
  fill_p2(p2, tables);
  const dim3 grid(blocks_for(n_elems)), block(kBlock);
  if (poly_order == 1) {
    TFEM_DISPATCH_Q(tables.nq, hipLaunchKernelGGL((k_load_atomic<T, I, Q, 3>), grid, block, 0,
                                                   stream, a, p2));
  } else if (poly_order == 2) {
    TFEM_DISPATCH_Q(tables.nq, hipLaunchKernelGGL((k_load_atomic<T, I, Q, 6>), grid, block, 0,
                                                   stream, a, p2));
  } else {
    return fail(TFEM_ERR_UNSUPPORTED, "Polynomial order not implemented");
  }
  TFEM_HIP_CHECK(hipGetLastError());
  return TFEM_OK;
}

template <typename T, typename I>
static int run_geometry(const void *coords, const void *conn, int64_t n_elems, int64_t n_verts,
                        int poly_order, int quad_order, void *v_grad, void *dx, void *points,
                        void *inv_jac, hipStream_t stream) {
  TriArgs<T, I> a;
  TriTables tables;
  FracSpec none{nullptr, nullptr, 0, 0};
  if (int st = setup_args(a, coords, conn, n_elems, n_verts, quad_order, none, tables)) return st;
  if (poly_order != 1 && poly_order != 2)
    return fail(TFEM_ERR_UNSUPPORTED, "Polynomial order not implemented");
  if (n_elems == 0) return TFEM_OK;
  a.o_vgrad = static_cast<T *>(v_grad);
  a.o_dx = static_cast<T *>(dx);
  a.o_points = static_cast<T *>(points);
  a.o_inv = static_cast<T *>(inv_jac);
  P2Tables<T> p2;
  fill_p2(p2, tables);
  const dim3 grid(blocks_for(n_elems)), block(kBlock);
  if (poly_order == 1) {
    TFEM_DISPATCH_Q(tables.nq, hipLaunchKernelGGL((k_geometry<T, I, Q, 1>), grid, block, 0,
                                                   stream, a, p2));
  } else {
    TFEM_DISPATCH_Q(tables.nq, hipLaunchKernelGGL((k_geometry<T, I, Q, 2>), grid, block, 0,
                                                   stream, a, p2));
  }
  TFEM_HIP_CHECK(hipGetLastError());
  return TFEM_OK;
}

template <typename T, typename I>
static int run_reduce_linear(const void *integrand, int64_t es, int64_t qs, const void *dx,
                             int64_t n_elems, int n_quad, int n_local, const void *conn_dof,
                             void *out, hipStream_t s) {
  const int64_t entries = n_elems * n_local;
  hipLaunchKernelGGL((k_reduce_linear<T, I>), dim3(blocks_for(entries)), dim3(kBlock), 0, s,
                     static_cast<const T *>(integrand), es, qs, static_cast<const T *>(dx),
                     entries, n_quad, n_local, static_cast<const I *>(conn_dof),
                     static_cast<T *>(out));
  TFEM_HIP_CHECK(hipGetLastError());
  return TFEM_OK;
}

}  // namespace tfem

// ---------------------------------------------------------------------------------------
// C entry points
// ---------------------------------------------------------------------------------------
using namespace tfem;

#define TFEM_DISPATCH_TYPES(real_bytes, idx_bytes, FN, ...)              \
  (real_bytes == 8                                                       \
       ? (idx_bytes == 4 ? FN<double, int32_t>(__VA_ARGS__) : FN<double, int64_t>(__VA_ARGS__)) \
       : (idx_bytes == 4 ? FN<float, int32_t>(__VA_ARGS__) : FN<float, int64_t>(__VA_ARGS__)))

// CSR values from entry-major local blocks through the gather map (tfem_csr_gather_map): one
// lane per CSR entry sums its contributions in ascending element order -- the order of the
// reference's sequential index_put_(accumulate=True) -- no atomics, bitwise reproducible.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_csr_gather(const T *local, const int64_t *gptr,
                                                        const int32_t *gsrc, int64_t nnz, T *vals) {
  const int64_t p = int64_t(blockIdx.x) * kBlock + threadIdx.x;
  if (p >= nnz) return;
  T acc = T(0);
  for (int64_t t = gptr[p]; t < gptr[p + 1]; ++t) acc = acc + local[gsrc[t]];
  vals[p] = acc;
}

// y = A x for the CSR operator (the consumer of the assembled values: Krylov solves on
// operators far beyond the reference's dense torch.linalg.solve, abstract_basis.py:177-195).
// Eight lanes per row (P1 rows hold ~7 entries, P2 rows 6-22): coalesced reads of vals/colind,
// x through L2, one 8-byte store per row.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_csr_spmv(const int64_t *rowptr, const int32_t *colind,
                                                      const T *vals, const T *x, T *y, int64_t n_rows) {
  const int64_t gid = int64_t(blockIdx.x) * kBlock + threadIdx.x;
  const int64_t row = gid >> 3;
  const int sub = int(gid & 7);
  T acc = T(0);
  if (row < n_rows) {
    const int64_t end = rowptr[row + 1];
    for (int64_t p = rowptr[row] + sub; p < end; p += 8) acc = acc + vals[p] * x[colind[p]];
  }
  acc = acc + __shfl_xor(acc, 4, 64);
  acc = acc + __shfl_xor(acc, 2, 64);
  acc = acc + __shfl_xor(acc, 1, 64);
  if (row < n_rows && sub == 0) y[row] = acc;
}

// (x, y) pairs as one 16-byte (fp64) / 8-byte (fp32) access: vertex coordinates, edge points,
// gradients.  The arrays start at allocation boundaries and hold pairs only.
template <typename T>
struct alignas(2 * sizeof(T)) Pair {
  T x, y;
};
template <typename T>
__device__ __forceinline__ Pair<T> load_pair(const T *base, int64_t index) {
  return *reinterpret_cast<const Pair<T> *>(base + 2 * index);
}

// P1 field on the two sides of every interior edge (Basis.interpolate(InteriorEdgesBasis, u),
// reference basis.py:98-177 with the tensor argument): one lane per (edge, side).  The lane
// rebuilds the side's affine map from the three vertices (element_tri.py:132-145), pulls the
// edge's quadrature points back with (x - x0) J^-T (abstract_element.py:18-26), evaluates the
// barycentric shape functions there and contracts with the three nodal values; the gradient
// is constant along the edge.  No intermediate (N_e, 2, Q, 3, .) tensors reach HBM.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_edge_interpolate_p1(
    const T *coords, const int32_t *conn, const int64_t *edge_cells, const T *points, const T *u,
    int64_t n_sides, int n_points, T *value, T *grad) {
  const int64_t side = int64_t(blockIdx.x) * kBlock + threadIdx.x;
  if (side >= n_sides) return;
  const int64_t cell = edge_cells[side];
  const int32_t v0 = conn[3 * cell], v1 = conn[3 * cell + 1], v2 = conn[3 * cell + 2];
  const Pair<T> p0 = load_pair(coords, v0), p1 = load_pair(coords, v1), p2 = load_pair(coords, v2);
  const T x0 = p0.x, y0 = p0.y;
  const T a = p1.x - x0, c = p1.y - y0;
  const T b = p2.x - x0, d = p2.y - y0;
  const T inv_det = T(1) / (a * d - b * c);
  const T i00 = inv_det * d, i01 = inv_det * (-b), i10 = inv_det * (-c), i11 = inv_det * a;
  const T u0 = u[v0], u1 = u[v1], u2 = u[v2];
  // rows of barycentric_grad @ J^-1: (-i0 - i1, i0, i1)
  *reinterpret_cast<Pair<T> *>(grad + 2 * side) =
      Pair<T>{u0 * (-i00 - i10) + u1 * i00 + u2 * i10, u0 * (-i01 - i11) + u1 * i01 + u2 * i11};
  const T *pts = points + (side >> 1) * int64_t(2 * n_points);
  for (int q = 0; q < n_points; ++q) {
    const Pair<T> pq = load_pair(pts, q);
    const T dx = pq.x - x0, dy = pq.y - y0;
    const T xi = dx * i00 + dy * i01, eta = dx * i10 + dy * i11;
    value[side * n_points + q] = u0 * (T(1) - xi - eta) + u1 * xi + u2 * eta;
  }
}

// Adjoint of k_edge_interpolate_p1 with respect to the nodal values (training loops that
// differentiate jump terms): per (edge, side) the three weights of the value at every point and
// of the constant gradient, times the incoming cotangents, added to grad_u with hardware
// floating-point atomics (grad_u zeroed by the caller; summation order is not fixed).
template <typename T>
__global__ __launch_bounds__(kBlock) void k_edge_interpolate_p1_backward(
    const T *coords, const int32_t *conn, const int64_t *edge_cells, const T *points,
    const T *g_value, const T *g_grad, int64_t n_sides, int n_points, T *grad_u) {
  const int64_t side = int64_t(blockIdx.x) * kBlock + threadIdx.x;
  if (side >= n_sides) return;
  const int64_t cell = edge_cells[side];
  const int32_t v0 = conn[3 * cell], v1 = conn[3 * cell + 1], v2 = conn[3 * cell + 2];
  const Pair<T> p0 = load_pair(coords, v0), p1 = load_pair(coords, v1), p2 = load_pair(coords, v2);
  const T x0 = p0.x, y0 = p0.y;
  const T a = p1.x - x0, c = p1.y - y0;
  const T b = p2.x - x0, d = p2.y - y0;
  const T inv_det = T(1) / (a * d - b * c);
  const T i00 = inv_det * d, i01 = inv_det * (-b), i10 = inv_det * (-c), i11 = inv_det * a;
  const Pair<T> gg = load_pair(g_grad, side);
  const T gx = gg.x, gy = gg.y;
  T w0 = gx * (-i00 - i10) + gy * (-i01 - i11);
  T w1 = gx * i00 + gy * i01;
  T w2 = gx * i10 + gy * i11;
  const T *pts = points + (side >> 1) * int64_t(2 * n_points);
  for (int q = 0; q < n_points; ++q) {
    const Pair<T> pq = load_pair(pts, q);
    const T dx = pq.x - x0, dy = pq.y - y0;
    const T xi = dx * i00 + dy * i01, eta = dx * i10 + dy * i11;
    const T gv = g_value[side * n_points + q];
    w0 = w0 + gv * (T(1) - xi - eta);
    w1 = w1 + gv * xi;
    w2 = w2 + gv * eta;
  }
  atomicAdd(grad_u + v0, w0);
  atomicAdd(grad_u + v1, w1);
  atomicAdd(grad_u + v2, w2);
}

// Interface exchange of the element-range sharding (parallel.py): entries of the local CSR
// values / local vector that belong to DoFs shared with another rank are copied into the
// packed buffer the ranks all-reduce (pack) and back (unpack).  One launch each.
template <typename T>
__global__ void k_interface_pack(const T *vals, const T *f, const int64_t *k_idx, const int64_t *k_pos,
                                 int64_t nk, const int64_t *f_idx, const int64_t *f_pos, int64_t nf,
                                 T *buf) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < nk) {
    if (vals) buf[k_pos[i]] = vals[k_idx[i]];
  } else if (i < nk + nf) {
    if (f) buf[f_pos[i - nk]] = f[f_idx[i - nk]];
  }
}

// The same pack in ONE launch (no memset in front): one thread per buffer position, src[p] >= 0: an
// entry of vals, src[p] <= -2: entry -src[p] - 2 of f, -1: a position other ranks own (zero).
template <typename T>
__global__ void k_interface_pack_dense(const T *vals, const T *f, const int64_t *src, int64_t nbuf, T *buf) {
  const int64_t p = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (p >= nbuf) return;
  const int64_t s = src[p];
  buf[p] = s >= 0 ? vals[s] : (s <= -2 ? f[-s - 2] : T(0));
}

template <typename T>
__global__ void k_interface_unpack(T *vals, T *f, const int64_t *k_idx, const int64_t *k_pos,
                                   int64_t nk, const int64_t *f_idx, const int64_t *f_pos, int64_t nf,
                                   const T *buf) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < nk) {
    if (vals) vals[k_idx[i]] = buf[k_pos[i]];
  } else if (i < nk + nf) {
    if (f) f[f_idx[i - nk]] = buf[f_pos[i - nk]];
  }
}

extern "C" {

int tfem_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

int tfem_tri_geometry(const void *coords, int real_bytes, const void *conn, int idx_bytes,
                      int64_t n_elems, int64_t n_verts, int poly_order, int quad_order,
                      void *v_grad, void *dx, void *points, void *inv_jac, void *stream) {
  if (int st = check_common(real_bytes, idx_bytes, n_elems)) return st;
  return TFEM_DISPATCH_TYPES(real_bytes, idx_bytes, run_geometry, coords, conn, n_elems, n_verts,
                             poly_order, quad_order, v_grad, dx, points, inv_jac,
                             static_cast<hipStream_t>(stream));
}

int tfem_tri_bilinear_csr(const void *coords, int real_bytes, const void *conn_geo, int idx_bytes,
                          int64_t n_elems, int64_t n_verts, int poly_order, int quad_order,
                          double alpha, double beta, const int32_t *slots, void *vals,
                          int64_t nnz, const void *frac_pinv, const void *frac_det,
                          int n_fractures, int64_t coords_per_fracture, void *stream) {
  if (int st = check_common(real_bytes, idx_bytes, n_elems)) return st;
  if (nnz < 0) return fail(TFEM_ERR_INVALID_ARGUMENT, "nnz < 0");
  if ((frac_pinv == nullptr) != (frac_det == nullptr))
    return fail(TFEM_ERR_INVALID_ARGUMENT, "frac_pinv and frac_det must be given together");
  FracSpec fr{frac_pinv, frac_det, n_fractures, coords_per_fracture};
  return TFEM_DISPATCH_TYPES(real_bytes, idx_bytes, run_bilinear, coords, conn_geo, n_elems,
                             n_verts, poly_order, quad_order, alpha, beta, slots, vals, nnz, fr,
                             static_cast<hipStream_t>(stream));
}

int tfem_tri_load_vector(const void *coords, int real_bytes, const void *conn_geo,
                         const void *conn_dof, int idx_bytes, int64_t n_elems, int64_t n_verts,
                         int poly_order, int quad_order, const void *fq, void *out,
                         int64_t n_dofs, const void *frac_det, int n_fractures,
                         int64_t coords_per_fracture, void *stream) {
  if (int st = check_common(real_bytes, idx_bytes, n_elems)) return st;
  if (n_dofs < 0) return fail(TFEM_ERR_INVALID_ARGUMENT, "n_dofs < 0");
  FracSpec fr{nullptr, frac_det, n_fractures, coords_per_fracture};
  return TFEM_DISPATCH_TYPES(real_bytes, idx_bytes, run_load, coords, conn_geo, conn_dof, n_elems,
                             n_verts, poly_order, quad_order, fq, out, n_dofs, fr,
                             static_cast<hipStream_t>(stream));
}

int tfem_reduce_scatter_bilinear(const void *integrand, int real_bytes, int64_t es, int64_t qs,
                                 const void *dx, int64_t n_elems, int n_quad, int n_local,
                                 const int32_t *slots, void *vals, int64_t nnz, void *stream) {
  if (int st = check_common(real_bytes, 4, n_elems)) return st;
  if (n_quad < 1 || n_local < 1 || nnz < 0)
    return fail(TFEM_ERR_INVALID_ARGUMENT, "bad n_quad / n_local / nnz");
  if (n_elems > 0 && (!integrand || !dx)) return fail(TFEM_ERR_INVALID_ARGUMENT, "NULL input");
  if (nnz > 0 && !vals) return fail(TFEM_ERR_INVALID_ARGUMENT, "vals is NULL");
  if (!slots && nnz != int64_t(n_local) * n_local * n_elems)
    return fail(TFEM_ERR_INVALID_ARGUMENT, "local-block mode: vals must hold n_local^2 * n_elems entries");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (slots && nnz > 0) TFEM_HIP_CHECK(hipMemsetAsync(vals, 0, size_t(nnz) * size_t(real_bytes), s));
  if (n_elems == 0) return TFEM_OK;
  const int nn = n_local * n_local;
  const int64_t entries = n_elems * nn;
  if (real_bytes == 8)
    hipLaunchKernelGGL(k_reduce_bilinear<double>, dim3(blocks_for(entries)), dim3(kBlock), 0, s,
                       static_cast<const double *>(integrand), es, qs,
                       static_cast<const double *>(dx), entries, n_quad, nn, slots,
                       static_cast<double *>(vals));
  else
    hipLaunchKernelGGL(k_reduce_bilinear<float>, dim3(blocks_for(entries)), dim3(kBlock), 0, s,
                       static_cast<const float *>(integrand), es, qs,
                       static_cast<const float *>(dx), entries, n_quad, nn, slots,
                       static_cast<float *>(vals));
  TFEM_HIP_CHECK(hipGetLastError());
  return TFEM_OK;
}

int tfem_reduce_scatter_linear(const void *integrand, int real_bytes, int64_t es, int64_t qs,
                               const void *dx, int64_t n_elems, int n_quad, int n_local,
                               const void *conn_dof, int idx_bytes, void *out, int64_t n_dofs,
                               void *stream) {
  if (int st = check_common(real_bytes, idx_bytes, n_elems)) return st;
  if (n_quad < 1 || n_local < 1 || n_dofs < 0)
    return fail(TFEM_ERR_INVALID_ARGUMENT, "bad n_quad / n_local / n_dofs");
  if (n_elems > 0 && (!integrand || !dx)) return fail(TFEM_ERR_INVALID_ARGUMENT, "NULL input");
  if (n_dofs > 0 && !out) return fail(TFEM_ERR_INVALID_ARGUMENT, "out is NULL");
  if (!conn_dof && n_dofs != int64_t(n_local) * n_elems)
    return fail(TFEM_ERR_INVALID_ARGUMENT, "local-vector mode: out must hold n_local * n_elems entries");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (conn_dof && n_dofs > 0) TFEM_HIP_CHECK(hipMemsetAsync(out, 0, size_t(n_dofs) * size_t(real_bytes), s));
  if (n_elems == 0) return TFEM_OK;
  return TFEM_DISPATCH_TYPES(real_bytes, idx_bytes, run_reduce_linear, integrand, es, qs, dx,
                             n_elems, n_quad, n_local, conn_dof, out, s);
}

int tfem_reduce_functional(const void *integrand, int real_bytes, int64_t es, int64_t qs,
                           const void *dx, int64_t n_elems, int n_quad, int n_inner, void *out,
                           void *stream) {
  if (int st = check_common(real_bytes, 4, n_elems)) return st;
  if (n_quad < 1 || n_inner < 1) return fail(TFEM_ERR_INVALID_ARGUMENT, "bad n_quad / n_inner");
  if (n_elems == 0) return TFEM_OK;
  if (!integrand || !dx || !out) return fail(TFEM_ERR_INVALID_ARGUMENT, "NULL pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (real_bytes == 8)
    hipLaunchKernelGGL(k_reduce_functional<double>, dim3(blocks_for(n_elems)), dim3(kBlock), 0, s,
                       static_cast<const double *>(integrand), es, qs,
                       static_cast<const double *>(dx), n_elems, n_quad, n_inner,
                       static_cast<double *>(out));
  else
    hipLaunchKernelGGL(k_reduce_functional<float>, dim3(blocks_for(n_elems)), dim3(kBlock), 0, s,
                       static_cast<const float *>(integrand), es, qs,
                       static_cast<const float *>(dx), n_elems, n_quad, n_inner,
                       static_cast<float *>(out));
  TFEM_HIP_CHECK(hipGetLastError());
  return TFEM_OK;
}

int tfem_csr_to_dense(const int64_t *rowptr, const int32_t *colind, const void *vals,
                      int real_bytes, int64_t n_dofs, void *dense, void *stream) {
  if (int st = check_common(real_bytes, 4, n_dofs)) return st;
  if (n_dofs == 0) return TFEM_OK;
  if (!rowptr || !dense) return fail(TFEM_ERR_INVALID_ARGUMENT, "NULL pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  TFEM_HIP_CHECK(
      hipMemsetAsync(dense, 0, size_t(n_dofs) * size_t(n_dofs) * size_t(real_bytes), s));
  if (real_bytes == 8)
    hipLaunchKernelGGL(k_csr_to_dense<double>, dim3(blocks_for(n_dofs)), dim3(kBlock), 0, s,
                       rowptr, colind, static_cast<const double *>(vals), n_dofs,
                       static_cast<double *>(dense));
  else
    hipLaunchKernelGGL(k_csr_to_dense<float>, dim3(blocks_for(n_dofs)), dim3(kBlock), 0, s,
                       rowptr, colind, static_cast<const float *>(vals), n_dofs,
                       static_cast<float *>(dense));
  TFEM_HIP_CHECK(hipGetLastError());
  return TFEM_OK;
}

int tfem_csr_gather(const void *local, int real_bytes, const int64_t *gptr, const int32_t *gsrc,
                    int64_t nnz, void *vals, void *stream) {
  if (real_bytes != 4 && real_bytes != 8) return fail(TFEM_ERR_INVALID_ARGUMENT, "real_bytes must be 4 or 8");
  if (nnz < 0 || (nnz > 0 && (!local || !gptr || !gsrc || !vals)))
    return fail(TFEM_ERR_INVALID_ARGUMENT, "bad arguments");
  if (nnz == 0) return TFEM_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (real_bytes == 8)
    hipLaunchKernelGGL(k_csr_gather<double>, dim3(blocks_for(nnz)), dim3(kBlock), 0, s,
                       static_cast<const double *>(local), gptr, gsrc, nnz, static_cast<double *>(vals));
  else
    hipLaunchKernelGGL(k_csr_gather<float>, dim3(blocks_for(nnz)), dim3(kBlock), 0, s,
                       static_cast<const float *>(local), gptr, gsrc, nnz, static_cast<float *>(vals));
  TFEM_HIP_CHECK(hipGetLastError());
  return TFEM_OK;
}

int tfem_csr_spmv(const int64_t *rowptr, const int32_t *colind, const void *vals, int real_bytes,
                  int64_t n_rows, const void *x, void *y, void *stream) {
  if (real_bytes != 4 && real_bytes != 8) return fail(TFEM_ERR_INVALID_ARGUMENT, "real_bytes must be 4 or 8");
  if (n_rows < 0 || (n_rows > 0 && (!rowptr || !x || !y)))
    return fail(TFEM_ERR_INVALID_ARGUMENT, "bad arguments");
  if (n_rows == 0) return TFEM_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (real_bytes == 8)
    hipLaunchKernelGGL(k_csr_spmv<double>, dim3(blocks_for(8 * n_rows)), dim3(kBlock), 0, s, rowptr, colind,
                       static_cast<const double *>(vals), static_cast<const double *>(x),
                       static_cast<double *>(y), n_rows);
  else
    hipLaunchKernelGGL(k_csr_spmv<float>, dim3(blocks_for(8 * n_rows)), dim3(kBlock), 0, s, rowptr, colind,
                       static_cast<const float *>(vals), static_cast<const float *>(x),
                       static_cast<float *>(y), n_rows);
  TFEM_HIP_CHECK(hipGetLastError());
  return TFEM_OK;
}

int tfem_edge_interpolate_p1(const void *coords, int real_bytes, const int32_t *conn,
                             const int64_t *edge_cells, const void *points, int64_t n_edges,
                             int n_points, const void *u, void *value, void *grad, void *stream) {
  if (real_bytes != 4 && real_bytes != 8) return fail(TFEM_ERR_INVALID_ARGUMENT, "real_bytes must be 4 or 8");
  if (n_edges < 0 || n_points < 0) return fail(TFEM_ERR_INVALID_ARGUMENT, "negative size");
  if (n_edges == 0) return TFEM_OK;
  if (!coords || !conn || !edge_cells || !u || !value || !grad || (n_points > 0 && !points))
    return fail(TFEM_ERR_INVALID_ARGUMENT, "NULL pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t n_sides = 2 * n_edges;
  if (real_bytes == 8)
    hipLaunchKernelGGL(k_edge_interpolate_p1<double>, dim3(blocks_for(n_sides)), dim3(kBlock), 0, s,
                       static_cast<const double *>(coords), conn, edge_cells,
                       static_cast<const double *>(points), static_cast<const double *>(u), n_sides,
                       n_points, static_cast<double *>(value), static_cast<double *>(grad));
  else
    hipLaunchKernelGGL(k_edge_interpolate_p1<float>, dim3(blocks_for(n_sides)), dim3(kBlock), 0, s,
                       static_cast<const float *>(coords), conn, edge_cells,
                       static_cast<const float *>(points), static_cast<const float *>(u), n_sides,
                       n_points, static_cast<float *>(value), static_cast<float *>(grad));
  TFEM_HIP_CHECK(hipGetLastError());
  return TFEM_OK;
}

int tfem_edge_interpolate_p1_backward(const void *coords, int real_bytes, const int32_t *conn,
                                      const int64_t *edge_cells, const void *points, int64_t n_edges,
                                      int n_points, const void *g_value, const void *g_grad,
                                      void *grad_u, int64_t n_verts, void *stream) {
  if (real_bytes != 4 && real_bytes != 8) return fail(TFEM_ERR_INVALID_ARGUMENT, "real_bytes must be 4 or 8");
  if (n_edges < 0 || n_points < 0 || n_verts < 0) return fail(TFEM_ERR_INVALID_ARGUMENT, "negative size");
  if (n_verts == 0) return TFEM_OK;
  if (!grad_u) return fail(TFEM_ERR_INVALID_ARGUMENT, "NULL pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  TFEM_HIP_CHECK(hipMemsetAsync(grad_u, 0, size_t(n_verts) * size_t(real_bytes), s));
  if (n_edges == 0) return TFEM_OK;
  if (!coords || !conn || !edge_cells || !g_value || !g_grad || (n_points > 0 && !points))
    return fail(TFEM_ERR_INVALID_ARGUMENT, "NULL pointer");
  const int64_t n_sides = 2 * n_edges;
  if (real_bytes == 8)
    hipLaunchKernelGGL(k_edge_interpolate_p1_backward<double>, dim3(blocks_for(n_sides)), dim3(kBlock), 0, s,
                       static_cast<const double *>(coords), conn, edge_cells,
                       static_cast<const double *>(points), static_cast<const double *>(g_value),
                       static_cast<const double *>(g_grad), n_sides, n_points, static_cast<double *>(grad_u));
  else
    hipLaunchKernelGGL(k_edge_interpolate_p1_backward<float>, dim3(blocks_for(n_sides)), dim3(kBlock), 0, s,
                       static_cast<const float *>(coords), conn, edge_cells,
                       static_cast<const float *>(points), static_cast<const float *>(g_value),
                       static_cast<const float *>(g_grad), n_sides, n_points, static_cast<float *>(grad_u));
  TFEM_HIP_CHECK(hipGetLastError());
  return TFEM_OK;
}

int tfem_interface_pack(const void *vals, const void *f, int real_bytes, const int64_t *k_idx,
                        const int64_t *k_pos, int64_t nk, const int64_t *f_idx, const int64_t *f_pos,
                        int64_t nf, void *buf, int64_t nbuf, void *stream) {
  if (real_bytes != 4 && real_bytes != 8) return fail(TFEM_ERR_INVALID_ARGUMENT, "real_bytes must be 4 or 8");
  if (nk < 0 || nf < 0 || nbuf < 0 || (nbuf > 0 && !buf) || (nk > 0 && (!k_idx || !k_pos)) ||
      (nf > 0 && (!f_idx || !f_pos)))
    return fail(TFEM_ERR_INVALID_ARGUMENT, "bad arguments");
  if (nbuf == 0) return TFEM_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // positions this rank does not share stay zero: the neutral element of the all-reduce
  TFEM_HIP_CHECK(hipMemsetAsync(buf, 0, size_t(nbuf) * size_t(real_bytes), s));
  if (nk + nf == 0) return TFEM_OK;
  if (real_bytes == 8)
    hipLaunchKernelGGL(k_interface_pack<double>, dim3(blocks_for(nk + nf)), dim3(kBlock), 0, s,
                       static_cast<const double *>(vals), static_cast<const double *>(f), k_idx, k_pos,
                       nk, f_idx, f_pos, nf, static_cast<double *>(buf));
  else
    hipLaunchKernelGGL(k_interface_pack<float>, dim3(blocks_for(nk + nf)), dim3(kBlock), 0, s,
                       static_cast<const float *>(vals), static_cast<const float *>(f), k_idx, k_pos,
                       nk, f_idx, f_pos, nf, static_cast<float *>(buf));
  TFEM_HIP_CHECK(hipGetLastError());
  return TFEM_OK;
}

int tfem_interface_pack_dense(const void *vals, const void *f, int real_bytes, const int64_t *src,
                              int64_t nbuf, void *buf, void *stream) {
  if (real_bytes != 4 && real_bytes != 8) return fail(TFEM_ERR_INVALID_ARGUMENT, "real_bytes must be 4 or 8");
  if (nbuf < 0) return fail(TFEM_ERR_INVALID_ARGUMENT, "negative size");
  if (nbuf == 0) return TFEM_OK;
  if (!vals || !f || !src || !buf) return fail(TFEM_ERR_INVALID_ARGUMENT, "NULL pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (real_bytes == 8)
    hipLaunchKernelGGL(k_interface_pack_dense<double>, dim3(blocks_for(nbuf)), dim3(kBlock), 0, s,
                       static_cast<const double *>(vals), static_cast<const double *>(f), src, nbuf,
                       static_cast<double *>(buf));
  else
    hipLaunchKernelGGL(k_interface_pack_dense<float>, dim3(blocks_for(nbuf)), dim3(kBlock), 0, s,
                       static_cast<const float *>(vals), static_cast<const float *>(f), src, nbuf,
                       static_cast<float *>(buf));
  TFEM_HIP_CHECK(hipGetLastError());
  return TFEM_OK;
}

int tfem_interface_unpack(void *vals, void *f, int real_bytes, const int64_t *k_idx,
                          const int64_t *k_pos, int64_t nk, const int64_t *f_idx,
                          const int64_t *f_pos, int64_t nf, const void *buf, void *stream) {
  if (real_bytes != 4 && real_bytes != 8) return fail(TFEM_ERR_INVALID_ARGUMENT, "real_bytes must be 4 or 8");
  if (nk < 0 || nf < 0 || (nk + nf > 0 && !buf) || (nk > 0 && (!k_idx || !k_pos)) ||
      (nf > 0 && (!f_idx || !f_pos)))
    return fail(TFEM_ERR_INVALID_ARGUMENT, "bad arguments");
  if (nk + nf == 0) return TFEM_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (real_bytes == 8)
    hipLaunchKernelGGL(k_interface_unpack<double>, dim3(blocks_for(nk + nf)), dim3(kBlock), 0, s,
                       static_cast<double *>(vals), static_cast<double *>(f), k_idx, k_pos, nk, f_idx,
                       f_pos, nf, static_cast<const double *>(buf));
  else
    hipLaunchKernelGGL(k_interface_unpack<float>, dim3(blocks_for(nk + nf)), dim3(kBlock), 0, s,
                       static_cast<float *>(vals), static_cast<float *>(f), k_idx, k_pos, nk, f_idx,
                       f_pos, nf, static_cast<const float *>(buf));
  TFEM_HIP_CHECK(hipGetLastError());
  return TFEM_OK;
}

}  // extern "C"
