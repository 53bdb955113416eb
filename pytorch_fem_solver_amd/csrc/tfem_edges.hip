// Interior-edge path, second part:
//  * tfem_edge_interpolate_p1_fracture: FractureBasis.interpolate(InteriorEdgesFractureBasis, u)
//    (fracture_basis.py:225-272): a P1 DoF vector on both sides of every interior edge of every
//    fracture, at the edge's 3-D quadrature points -- one lane per (fracture, edge, side).
//  * tfem_edge_interpolate_p1_backward_rows: the adjoint of tfem_edge_interpolate_p1 in the
//    nodal values WITHOUT atomics: one lane per vertex walks the (edge, side) pairs around it
//    (incidence table built once per edge basis), fixed summation order, bitwise reproducible.
#include <hip/hip_runtime.h>

#include "tfem_common.hpp"

namespace tfem {
namespace {

constexpr int kEdgeBlock = 256;

static inline unsigned edge_blocks(int64_t work) { return unsigned((work + kEdgeBlock - 1) / kEdgeBlock); }

// One lane per (fracture, edge, side).  Operation order of the reference:
//   inv_jac (2x3) = inv(J) (2x2, element_tri.py:132-145) @ J_F^+ (2x3)      fracture_basis.py:24-26
//   local   (2)   = (p - origin) @ inv_jac^T, sum over the 3 components in order   abstract_element.py:18-26
//   bar           = (1 - xi - eta, xi, eta)                                   element_tri.py:23-26
//   v_grad (3x3)  = barycentric_grad (3x2) @ inv_jac (2x3)                    element_tri.py:41
//   value = sum_i u_i bar_i,  grad = sum_i u_i v_grad_i                       fracture_basis.py:266-272
// u is indexed with the cells' PER-FRACTURE vertex ids, as the reference does
// (fracture_basis.py:229-231 gathers mesh["cells", "vertices"]).
template <typename T>
__global__ __launch_bounds__(kEdgeBlock) void k_edge_interpolate_p1_fracture(
    const T *coords2d, const T *coords3d, const int32_t *conn, const int64_t *edge_cells, const T *points,
    const T *pinv, const T *u, int64_t n_fractures, int64_t n_verts, int64_t n_cells, int64_t n_edges,
    int n_points, int64_t n_u, T *value, T *grad) {
  const int64_t side = int64_t(blockIdx.x) * kEdgeBlock + threadIdx.x;
  const int64_t per_fracture = 2 * n_edges;
  if (side >= n_fractures * per_fracture) return;
  const int64_t f = side / per_fracture;
  const int64_t cell = edge_cells[side];  // (F, n_edges, 2) flat
  const int32_t *c = conn + 3 * (f * n_cells + cell);
  const int64_t v0 = c[0], v1 = c[1], v2 = c[2];
  const T *x2 = coords2d + 2 * f * n_verts;
  const T x0 = x2[2 * v0], y0 = x2[2 * v0 + 1];
  const T a = x2[2 * v1] - x0, cc = x2[2 * v1 + 1] - y0;
  const T b = x2[2 * v2] - x0, d = x2[2 * v2 + 1] - y0;
  const T r = T(1) / (a * d - b * cc);
  const T i2[2][2] = {{r * d, r * (-b)}, {r * (-cc), r * a}};
  const T *pf = pinv + 6 * f;  // (2, 3) row-major
  T inv[2][3];
#pragma unroll
  for (int row = 0; row < 2; ++row)
#pragma unroll
    for (int k = 0; k < 3; ++k) inv[row][k] = i2[row][0] * pf[k] + i2[row][1] * pf[3 + k];
  const T *o = coords3d + 3 * (f * n_verts + v0);  // the cell's first vertex in 3-D
  // the reference indexes the (global) vector with the per-fracture ids; ids beyond the vector
  // cannot occur (every fracture has at most as many vertices as the merged numbering)
  const T u0 = u[v0 < n_u ? v0 : 0], u1 = u[v1 < n_u ? v1 : 0], u2 = u[v2 < n_u ? v2 : 0];
#pragma unroll
  for (int k = 0; k < 3; ++k)
    grad[3 * side + k] = (u0 * ((-inv[0][k]) + (-inv[1][k])) + u1 * inv[0][k]) + u2 * inv[1][k];
  const T *pts = points + (side >> 1) * int64_t(3 * n_points);
  for (int q = 0; q < n_points; ++q) {
    const T d0 = pts[3 * q] - o[0], d1 = pts[3 * q + 1] - o[1], d2 = pts[3 * q + 2] - o[2];
    const T xi = (d0 * inv[0][0] + d1 * inv[0][1]) + d2 * inv[0][2];
    const T eta = (d0 * inv[1][0] + d1 * inv[1][1]) + d2 * inv[1][2];
    value[side * n_points + q] = (u0 * ((T(1) - xi) - eta) + u1 * xi) + u2 * eta;
  }
}

// Adjoint of k_edge_interpolate_p1 in row form: lane = vertex; (inc_ptr, inc_side) list the
// (edge, side) pairs whose cell has the vertex, inc_side = 4 * side + local index of the vertex
// in that cell, ascending.
template <typename T>
__global__ __launch_bounds__(kEdgeBlock) void k_edge_backward_rows(
    const T *coords, const int32_t *conn, const int64_t *edge_cells, const T *points, const T *g_value,
    const T *g_grad, const int64_t *inc_ptr, const int64_t *inc_side, int64_t n_verts, int n_points,
    T *grad_u) {
  const int64_t v = int64_t(blockIdx.x) * kEdgeBlock + threadIdx.x;
  if (v >= n_verts) return;
  T acc = T(0);
  for (int64_t p = inc_ptr[v]; p < inc_ptr[v + 1]; ++p) {
    const int64_t side = inc_side[p] >> 2;
    const int loc = int(inc_side[p] & 3);
    const int64_t cell = edge_cells[side];
    const int32_t v0 = conn[3 * cell], v1 = conn[3 * cell + 1], v2 = conn[3 * cell + 2];
    const T x0 = coords[2 * v0], y0 = coords[2 * v0 + 1];
    const T a = coords[2 * v1] - x0, c = coords[2 * v1 + 1] - y0;
    const T b = coords[2 * v2] - x0, d = coords[2 * v2 + 1] - y0;
    const T inv_det = T(1) / (a * d - b * c);
    const T i00 = inv_det * d, i01 = inv_det * (-b), i10 = inv_det * (-c), i11 = inv_det * a;
    const T gx = g_grad[2 * side], gy = g_grad[2 * side + 1];
    // weight of u_loc in the constant gradient: row `loc` of barycentric_grad @ J^-1
    T w = loc == 0 ? gx * (-i00 - i10) + gy * (-i01 - i11)
                   : (loc == 1 ? gx * i00 + gy * i01 : gx * i10 + gy * i11);
    const T *pts = points + (side >> 1) * int64_t(2 * n_points);
    for (int q = 0; q < n_points; ++q) {
      const T dx = pts[2 * q] - x0, dy = pts[2 * q + 1] - y0;
      const T xi = dx * i00 + dy * i01, eta = dx * i10 + dy * i11;
      const T shape = loc == 0 ? T(1) - xi - eta : (loc == 1 ? xi : eta);
      w = w + g_value[side * n_points + q] * shape;
    }
    acc = acc + w;
  }
  grad_u[v] = acc;
}

}  // namespace
}  // namespace tfem

extern "C" {

int tfem_edge_interpolate_p1_fracture(const void *coords2d, const void *coords3d, int real_bytes,
                                      const int32_t *conn, const int64_t *edge_cells, const void *points,
                                      const void *pinv, int64_t n_fractures, int64_t n_verts,
                                      int64_t n_cells, int64_t n_edges, int n_points, const void *u,
                                      int64_t n_u, void *value, void *grad, void *stream) {
  using namespace tfem;
  if (real_bytes != 4 && real_bytes != 8) return fail(TFEM_ERR_INVALID_ARGUMENT, "real_bytes must be 4 or 8");
  if (n_fractures < 0 || n_verts < 0 || n_cells < 0 || n_edges < 0 || n_points < 0 || n_u < 1)
    return fail(TFEM_ERR_INVALID_ARGUMENT, "bad sizes");
  const int64_t n_sides = 2 * n_edges * n_fractures;
  if (n_sides == 0) return TFEM_OK;
  if (!coords2d || !coords3d || !conn || !edge_cells || !pinv || !u || !value || !grad || (n_points > 0 && !points))
    return fail(TFEM_ERR_INVALID_ARGUMENT, "NULL pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (real_bytes == 8)
    hipLaunchKernelGGL(k_edge_interpolate_p1_fracture<double>, dim3(edge_blocks(n_sides)), dim3(kEdgeBlock), 0, s,
                       static_cast<const double *>(coords2d), static_cast<const double *>(coords3d), conn,
                       edge_cells, static_cast<const double *>(points), static_cast<const double *>(pinv),
                       static_cast<const double *>(u), n_fractures, n_verts, n_cells, n_edges, n_points, n_u,
                       static_cast<double *>(value), static_cast<double *>(grad));
  else
    hipLaunchKernelGGL(k_edge_interpolate_p1_fracture<float>, dim3(edge_blocks(n_sides)), dim3(kEdgeBlock), 0, s,
                       static_cast<const float *>(coords2d), static_cast<const float *>(coords3d), conn,
                       edge_cells, static_cast<const float *>(points), static_cast<const float *>(pinv),
                       static_cast<const float *>(u), n_fractures, n_verts, n_cells, n_edges, n_points, n_u,
                       static_cast<float *>(value), static_cast<float *>(grad));
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(TFEM_ERR_HIP, "fracture edge interpolation launch: %s", hipGetErrorString(e));
  return TFEM_OK;
}

int tfem_edge_interpolate_p1_backward_rows(const void *coords, int real_bytes, const int32_t *conn,
                                           const int64_t *edge_cells, const void *points,
                                           int64_t n_edges, int n_points, const void *g_value,
                                           const void *g_grad, const int64_t *inc_ptr,
                                           const int64_t *inc_side, void *grad_u, int64_t n_verts,
                                           void *stream) {
  using namespace tfem;
  if (real_bytes != 4 && real_bytes != 8) return fail(TFEM_ERR_INVALID_ARGUMENT, "real_bytes must be 4 or 8");
  if (n_edges < 0 || n_points < 0 || n_verts < 0) return fail(TFEM_ERR_INVALID_ARGUMENT, "negative size");
  if (n_verts == 0) return TFEM_OK;
  if (!grad_u || !inc_ptr || (n_edges > 0 && (!coords || !conn || !edge_cells || !g_value || !g_grad || !inc_side ||
                                             (n_points > 0 && !points))))
    return fail(TFEM_ERR_INVALID_ARGUMENT, "NULL pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (real_bytes == 8)
    hipLaunchKernelGGL(k_edge_backward_rows<double>, dim3(edge_blocks(n_verts)), dim3(kEdgeBlock), 0, s,
                       static_cast<const double *>(coords), conn, edge_cells, static_cast<const double *>(points),
                       static_cast<const double *>(g_value), static_cast<const double *>(g_grad), inc_ptr,
                       inc_side, n_verts, n_points, static_cast<double *>(grad_u));
  else
    hipLaunchKernelGGL(k_edge_backward_rows<float>, dim3(edge_blocks(n_verts)), dim3(kEdgeBlock), 0, s,
                       static_cast<const float *>(coords), conn, edge_cells, static_cast<const float *>(points),
                       static_cast<const float *>(g_value), static_cast<const float *>(g_grad), inc_ptr,
                       inc_side, n_verts, n_points, static_cast<float *>(grad_u));
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(TFEM_ERR_HIP, "edge backward launch: %s", hipGetErrorString(e));
  return TFEM_OK;
}

}  // extern "C"
