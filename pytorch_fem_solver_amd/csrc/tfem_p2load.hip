// P2 load vector in ROW form over the P2 row plan (tfem_p2rows_host.cpp) -- the load-vector
// counterpart of tfem_p2rows.hip (abstract_basis.py:95-112 for ElementTri(2, .),
// element_tri.py:43-70):
//     f_r = sum over the triangles T of row r   det_T * sum_q fq[T][q] phi_loc(q) w_q / 2
// One lane per DoF on the tiles of the stiffness launch: a vertex row walks its fan, an edge row
// takes its one or two triangles.  det_T is the cross product the stiffness rows form from the
// tile's coordinates in LDS (signed by the stored orientation, element_tri.py:139); the element
// and the local index of the row's DoF in it come from the plan's codes (element * 4 + index,
// 8 dwords per vertex row, 2 per edge row); the Q source values of the element come from HBM
// (every element is read by its six rows: once from HBM, five times from the L2 the rows of a
// tile share).  No atomics, no element vectors in HBM, one store per DoF.  The order of the sum
// is the fan's, not the reference's element order (parity at 1e-12, tests/test_hip_parity.py).
// Vertices with 8 .. 15 neighbours (long rows of the plan): sixteen lanes per row,
// k_p2_load_long_rows, as in the stiffness launch.
#include <hip/hip_runtime.h>

#include <cstring>

#include "tfem_common.hpp"
#include "tfem_rowkit.hpp"

#pragma clang fp contract(fast)

namespace tfem {

constexpr int kP2LoadBlock = 256;

template <typename T>
struct P2LoadArgs {
  const T *coords;
  const unsigned char *plan;
  const T *fq;
  T *out;
  unsigned coords_bytes, plan_bytes, fq_bytes, out_bytes;
  unsigned off_desc, off_rows, off_gid, off_codes;
  int n_tiles;
  int lds_vert;
  T tab[3][6];  // [index of the DoF among the three of its kind][q] = phi(q) w_q / 2
};

// Q source values of one element; an offset past the end of the buffer reads zeros
template <typename T, int NQ>
__device__ __forceinline__ void p2_load_fq(ring_rsrc_t r, unsigned byte, T (&v)[NQ]) {
  if constexpr (sizeof(T) == 8) {
#pragma unroll
    for (int q = 0; q + 1 < NQ; q += 2) {
      const ru32x4 x = __builtin_amdgcn_raw_buffer_load_b128(r, byte + unsigned(q) * 8u, 0, 0);
      v[q] = __builtin_bit_cast(double, ru32x2{x.x, x.y});
      v[q + 1] = __builtin_bit_cast(double, ru32x2{x.z, x.w});
    }
    if constexpr (NQ & 1) {  // two dword loads: raw_buffer_load_b64 is miscompiled by this hipcc (tfem_tiles.hip)
      const unsigned o = byte + unsigned(NQ - 1) * 8u;
      const unsigned lo = __builtin_amdgcn_raw_buffer_load_b32(r, o, 0, 0);
      const unsigned hi = __builtin_amdgcn_raw_buffer_load_b32(r, o + 4u, 0, 0);
      v[NQ - 1] = __builtin_bit_cast(double, ru32x2{lo, hi});
    }
  } else {
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      v[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte + unsigned(q) * 4u, 0, 0));
  }
}

constexpr unsigned kP2LoadPast = 0xFFFFFF00u;  // beyond every buffer the launch accepts

// det-free share of one triangle: sum_q fq[element][q] * tab[index][q]; 0 without a triangle.
// Two steps, so that the loads of all the slots of a row are in flight together with the
// coordinate gathers of the tile: p2_issue_fq before the barrier, p2_reduce_fq behind it.
template <typename T, int NQ>
__device__ __forceinline__ void p2_issue_fq(ring_rsrc_t r_fq, uint32_t code, bool on, T (&v)[NQ]) {
  p2_load_fq<T, NQ>(r_fq, on ? (code >> 2) * unsigned(NQ * sizeof(T)) : kP2LoadPast, v);
}
template <typename T, int NQ>
__device__ __forceinline__ T p2_reduce_fq(const T *tab, uint32_t code, bool on, const T (&v)[NQ]) {
  const T *t = tab + (on ? (code & 3u) : 0u) * 6u;
  T g = v[0] * t[0];
#pragma unroll
  for (int q = 1; q < NQ; ++q) g = g + v[q] * t[q];
  return on ? g : T(0);
}
template <typename T, int NQ>
__device__ __forceinline__ T p2_load_share(ring_rsrc_t r_fq, const T *tab, uint32_t code, bool on) {
  T v[NQ];
  p2_issue_fq<T, NQ>(r_fq, code, on, v);
  return p2_reduce_fq<T, NQ>(tab, code, on, v);
}

// KIND 0: vertex rows, KIND 1: edge rows.  One tile per 256-lane workgroup, as k_p2_rows.
template <typename T, int KIND, int NQ>
__global__ __launch_bounds__(kP2LoadBlock) void k_p2_load_rows(const P2LoadArgs<T> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char p2l_smem[];
  T *xy = reinterpret_cast<T *>(p2l_smem);  // [2 * lds_vert]
  T *tab = xy + 2 * a.lds_vert;             // [3][6]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int per = (a.n_tiles + 7) / 8;
  const int tile = int(blockIdx.x & 7) * per + int(blockIdx.x >> 3);
  if (tile >= a.n_tiles || int(blockIdx.x >> 3) >= per) return;
  ring_const_i32 d = (ring_const_i32)(uintptr_t)(a.plan + a.off_desc + 64u * unsigned(tile));
  const int vert_off = d[0], n_vert = d[1], row_off = d[2], n_own = d[7];
  const int row0 = d[3 + wave], row1 = d[4 + wave], dof0 = d[8 + wave];
  const ring_rsrc_t r_coords = ring_rsrc(a.coords, a.coords_bytes);
  const ring_rsrc_t r_plan = ring_rsrc(a.plan, a.plan_bytes);
  const ring_rsrc_t r_fq = ring_rsrc(a.fq, a.fq_bytes);
  const ring_rsrc_t r_out = ring_rsrc(a.out, a.out_bytes);
  const int my_row = row0 + lane;
  const bool has_row = my_row < row1;
  constexpr unsigned kNone = 0x3FFFFFFu;
  const unsigned row = has_row ? unsigned(row_off + my_row) : kNone;
  if (tid < 18) tab[tid] = a.tab[tid / 6][tid % 6];
  T acc = T(0);
  bool writes = has_row;
  if constexpr (KIND == 0) {
    uint32_t w[4], c[7];
    {
      const ru32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r_plan, a.off_rows + row * 32u, 0, 0);
      const ru32x4 u = __builtin_amdgcn_raw_buffer_load_b128(r_plan, a.off_codes + row * 32u, 0, 0);
      const ru32x4 t = __builtin_amdgcn_raw_buffer_load_b128(r_plan, a.off_codes + row * 32u + 16u, 0, 0);
      w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
      c[0] = u.x; c[1] = u.y; c[2] = u.z; c[3] = u.w;
      c[4] = t.x; c[5] = t.y; c[6] = t.z;
    }
    // level 1 (above: record, codes): the lane's own vertex and the id of its halo vertex (at most
    // one per lane); level 2: the halo vertex and the source values of the row's triangles -- all in
    // flight together; the coordinates are parked while the source values still travel
    T ox = T(0), oy = T(0), hx = T(0), hy = T(0);
    const int hl = n_own + tid;
    const unsigned hg = __builtin_amdgcn_raw_buffer_load_b32(
        r_plan, hl < n_vert ? a.off_gid + unsigned(vert_off + hl) * 4u : kP2LoadPast, 0, 0);
    if (has_row) ring_load_xy<T>(r_coords, unsigned(dof0 + lane), ox, oy);
    const int k = int((w[2] >> 24) & 7u);
    auto id = [&](int i) { return (w[i / 3] >> (10 * (i % 3))) & 0x3FFu; };
    auto flag_of = [&](int i) { return (w[2] >> (10 + 2 * i)) & 3u; };
    if (hl < n_vert) ring_load_xy<T>(r_coords, hg, hx, hy);
    T fv[7][NQ];
#pragma unroll
    for (int i = 0; i < 7; ++i) p2_issue_fq<T, NQ>(r_fq, c[i], flag_of(i) != 0u, fv[i]);
    if (has_row) {
      xy[2 * my_row] = ox;
      xy[2 * my_row + 1] = oy;
    }
    if (hl < n_vert) {
      xy[2 * hl] = hx;
      xy[2 * hl + 1] = hy;
    }
    __syncthreads();
    T g[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) g[i] = p2_reduce_fq<T, NQ>(tab, c[i], flag_of(i) != 0u, fv[i]);
    T xv, yv, px, py;
    lds_xy(xy, unsigned(has_row ? my_row : 0), xv, yv);
    const uint32_t id0 = id(0);
    lds_xy(xy, id0, px, py);
    T ecx = px - xv, ecy = py - yv;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const uint32_t idn = (i + 1 < 7 && i + 1 != k) ? id(i + 1 < 7 ? i + 1 : 0) : id0;
      lds_xy(xy, idn, px, py);
      const T enx = px - xv, eny = py - yv;
      const T cross = ecx * eny - ecy * enx;
      acc = acc + flag_weight<T>(cross, flag_of(i)) * g[i];
      ecx = enx;
      ecy = eny;
    }
    // a long row (k = 0, marked in w3) is written by k_p2_load_long_rows
    writes = has_row && !(k == 0 && (w[3] >> 31) != 0u);
  } else {
    uint32_t w[2], c[2];
    {
      const ru32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r_plan, a.off_rows + row * 16u, 0, 0);
      w[0] = v.x; w[1] = v.y;
      c[0] = __builtin_amdgcn_raw_buffer_load_b32(r_plan, a.off_codes + row * 8u, 0, 0);
      c[1] = __builtin_amdgcn_raw_buffer_load_b32(r_plan, a.off_codes + row * 8u + 4u, 0, 0);
    }
    const bool has2 = has_row && ((w[1] >> 10) & 1u);
    const bool rev = (w[1] >> 11) & 1u;
    // level 1 (above: record, codes): the ids of up to four tile vertices per lane; level 2: their
    // coordinates and the source values of the row's triangles
    unsigned eg[4];
    T ex[4], ey[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int l = tid + j * kP2LoadBlock;
      eg[j] = __builtin_amdgcn_raw_buffer_load_b32(
          r_plan, l < n_vert ? a.off_gid + unsigned(vert_off + l) * 4u : kP2LoadPast, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      ex[j] = ey[j] = T(0);
      if (tid + j * kP2LoadBlock < n_vert) ring_load_xy<T>(r_coords, eg[j], ex[j], ey[j]);
    }
    T fv1[NQ], fv2[NQ];
    p2_issue_fq<T, NQ>(r_fq, c[0], has_row, fv1);
    p2_issue_fq<T, NQ>(r_fq, c[1], has2, fv2);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int l = tid + j * kP2LoadBlock;
      if (l < n_vert) {
        xy[2 * l] = ex[j];
        xy[2 * l + 1] = ey[j];
      }
    }
    __syncthreads();
    const T g1 = p2_reduce_fq<T, NQ>(tab, c[0], has_row, fv1);
    const T g2 = p2_reduce_fq<T, NQ>(tab, c[1], has2, fv2);
    T ax, ay, bx, by, cx, cy, dx, dy;
    lds_xy(xy, w[0] & 0x3FFu, ax, ay);
    lds_xy(xy, (w[0] >> 10) & 0x3FFu, bx, by);
    lds_xy(xy, (w[0] >> 20) & 0x3FFu, cx, cy);
    lds_xy(xy, w[1] & 0x3FFu, dx, dy);
    {
      const T e1x = bx - ax, e1y = by - ay, e2x = cx - ax, e2y = cy - ay;
      acc = (e1x * e2y - e1y * e2x) * g1;
    }
    {
      // frame (a2, b2, d) = (b, a, d) when rev, (a, b, d) otherwise
      const T ox = rev ? bx : ax, oy = rev ? by : ay;
      const T tx = rev ? ax : bx, ty = rev ? ay : by;
      const T e1x = tx - ox, e1y = ty - oy, e2x = dx - ox, e2y = dy - oy;
      acc = acc + (e1x * e2y - e1y * e2x) * g2;
    }
  }
  // a wave's rows are consecutive DoFs: one contiguous store per wave
  const unsigned byte = writes ? unsigned(dof0 + lane) * unsigned(sizeof(T)) : kP2LoadPast;
  if constexpr (sizeof(T) == 8) {
    const ru32x2 b = __builtin_bit_cast(ru32x2, acc);
    __builtin_amdgcn_raw_buffer_store_b32(b.x, r_out, byte, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b32(b.y, r_out, byte + 4u, 0, 0);
  } else {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc), r_out, byte, 0, 0);
  }
}

// Vertex rows with 8 .. 15 neighbours: sixteen lanes per row, lane i = slot i of the fan, global
// ids from the long-row record (tfem_p2rows_host.cpp), the slot's code from the long codes.
template <typename T, int NQ>
__global__ __launch_bounds__(kP2LoadBlock) void k_p2_load_long_rows(const P2LoadArgs<T> a, unsigned off_long,
                                                                    unsigned off_long_codes, int n_long) {
  __shared__ T tab[18];
  if (threadIdx.x < 18) tab[threadIdx.x] = a.tab[threadIdx.x / 6][threadIdx.x % 6];
  __syncthreads();
  const int gtid = int(blockIdx.x) * kP2LoadBlock + int(threadIdx.x);
  const int row = gtid >> 4, i = gtid & 15;
  const bool live = row < n_long;
  const uint32_t *rec = reinterpret_cast<const uint32_t *>(a.plan + off_long) + 32 * size_t(live ? row : 0);
  const uint32_t *codes = reinterpret_cast<const uint32_t *>(a.plan + off_long_codes) + 16 * size_t(live ? row : 0);
  const uint32_t v = rec[0];
  const int k = int(rec[2] & 0xFFu);
  const bool slot = live && i < k;
  const uint32_t flag = slot ? (rec[3] >> (2 * i)) & 3u : 0u;
  const int nxt = i + 1 == k ? 0 : i + 1;
  const uint32_t g0 = rec[4 + (slot ? i : 0)], g1 = rec[4 + (slot ? nxt : 0)];
  const T xv = a.coords[2 * size_t(v)], yv = a.coords[2 * size_t(v) + 1];
  const T ecx = a.coords[2 * size_t(g0)] - xv, ecy = a.coords[2 * size_t(g0) + 1] - yv;
  const T enx = a.coords[2 * size_t(g1)] - xv, eny = a.coords[2 * size_t(g1) + 1] - yv;
  const ring_rsrc_t r_fq = ring_rsrc(a.fq, a.fq_bytes);
  const T g = p2_load_share<T, NQ>(r_fq, tab, codes[i], flag != 0u);
  T sum = flag_weight<T>(ecx * eny - ecy * enx, flag) * g;
  sum = sum + __shfl_xor(sum, 8, 64);
  sum = sum + __shfl_xor(sum, 4, 64);
  sum = sum + __shfl_xor(sum, 2, 64);
  sum = sum + __shfl_xor(sum, 1, 64);
  if (live && i == 0) a.out[v] = sum;
}

template <typename T, int NQ>
static int launch_p2_load_nq(const P2LoadArgs<T> &base, const TriTables &tables, const int64_t *z,
                             hipStream_t stream) {
  for (int kind = 0; kind < 2; ++kind) {
    if (z[kind] == 0) continue;
    P2LoadArgs<T> a = base;
    a.off_desc = unsigned(z[10 + 3 * kind]);
    a.off_rows = unsigned(z[11 + 3 * kind]);
    a.off_gid = unsigned(z[12 + 3 * kind]);
    a.off_codes = unsigned(z[19 + kind]);
    a.n_tiles = int(z[kind]);
    a.lds_vert = (int(z[4 + kind]) + 1) & ~1;
    // shape functions 0 .. 2 (vertex DoFs) or 3 .. 5 (edge DoFs) at the quadrature points, times
    // w_q / 2, in T (the reference multiplies f v by dx = w_q / 2 det: basis.py:93-96)
    for (int m = 0; m < 3; ++m)
      for (int q = 0; q < NQ; ++q) a.tab[m][q] = T(tables.phi2[q][3 * kind + m]) * T(tables.hw[q]);
    const size_t lds = size_t(2 * a.lds_vert + 18) * sizeof(T);
    void *kernel = kind == 0 ? reinterpret_cast<void *>(k_p2_load_rows<T, 0, NQ>)
                             : reinterpret_cast<void *>(k_p2_load_rows<T, 1, NQ>);
    const int per = int((z[kind] + 7) / 8);
    const dim3 grid{unsigned(per * 8)}, block{unsigned(kP2LoadBlock)};
    void *params[] = {&a};
    hipError_t e = hipLaunchKernel(kernel, grid, block, params, lds, stream);
    if (e != hipSuccess) return fail(TFEM_ERR_HIP, "P2 load-vector launch: %s", hipGetErrorString(e));
    if (kind == 0 && z[18] > 0) {  // the vertex rows with 8 .. 15 neighbours
      const dim3 lgrid{unsigned((16 * z[18] + kP2LoadBlock - 1) / kP2LoadBlock)};
      hipLaunchKernelGGL((k_p2_load_long_rows<T, NQ>), lgrid, block, 0, stream, a, unsigned(z[17]), unsigned(z[21]),
                         int(z[18]));
      e = hipGetLastError();
      if (e != hipSuccess) return fail(TFEM_ERR_HIP, "P2 long-row load launch: %s", hipGetErrorString(e));
    }
  }
  return TFEM_OK;
}

template <typename T>
static int launch_p2_load(const void *coords, int quad_order, const unsigned char *plan, const int64_t *z,
                          const void *fq, int64_t n_elems, void *out, int64_t n_dofs, hipStream_t stream) {
  TriTables tables;
  if (!build_tri_tables(quad_order, int(sizeof(T)), &tables))
    return fail(TFEM_ERR_UNSUPPORTED, "Integration order not implemented");
  if (n_dofs != z[2] + z[3])
    return fail(TFEM_ERR_INVALID_ARGUMENT, "the plan is for %lld DoFs, not %lld", (long long)(z[2] + z[3]),
                (long long)n_dofs);
  if (z[0] + z[1] == 0) return TFEM_OK;
  if (!coords || !plan || !fq || !out) return fail(TFEM_ERR_INVALID_ARGUMENT, "NULL pointer");
  if (z[4] > 1024 || z[5] > 1024)
    return fail(TFEM_ERR_INVALID_ARGUMENT, "P2 row plan exceeds the kernel's capacities");
  const int64_t rb = int64_t(sizeof(T));
  const int64_t extents[4] = {z[2] * 2 * rb, z[16], n_elems * tables.nq * rb, n_dofs * rb};
  for (int64_t e : extents)
    if (e < 0 || e >= int64_t(kP2LoadPast))
      return fail(TFEM_ERR_INDEX_RANGE, "an array of %lld bytes does not fit the 32-bit offsets "
                  "of the P2 load-vector kernel", (long long)e);
  P2LoadArgs<T> a;
  std::memset(&a, 0, sizeof(a));
  a.coords = static_cast<const T *>(coords);
  a.plan = plan;
  a.fq = static_cast<const T *>(fq);
  a.out = static_cast<T *>(out);
  a.coords_bytes = unsigned(extents[0]);
  a.plan_bytes = unsigned(extents[1]);
  a.fq_bytes = unsigned(extents[2]);
  a.out_bytes = unsigned(extents[3]);
  switch (tables.nq) {
    case 1: return launch_p2_load_nq<T, 1>(a, tables, z, stream);
    case 3: return launch_p2_load_nq<T, 3>(a, tables, z, stream);
    case 4: return launch_p2_load_nq<T, 4>(a, tables, z, stream);
    case 6: return launch_p2_load_nq<T, 6>(a, tables, z, stream);
  }
  return fail(TFEM_ERR_UNSUPPORTED, "a quadrature rule of %d points", tables.nq);
}

}  // namespace tfem

extern "C" {

int tfem_p2_load_rows(const void *coords, int real_bytes, int quad_order, const void *plan_device,
                      const int64_t *plan_layout_host, const void *fq, int64_t n_elems, void *out,
                      int64_t n_dofs, void *stream) {
  using namespace tfem;
  if (real_bytes != 4 && real_bytes != 8)
    return fail(TFEM_ERR_INVALID_ARGUMENT, "real_bytes must be 4 or 8");
  if (!plan_layout_host) return fail(TFEM_ERR_INVALID_ARGUMENT, "plan_layout_host is NULL");
  if (n_elems < 0 || n_dofs < 0) return fail(TFEM_ERR_INVALID_ARGUMENT, "negative size");
  const auto *plan = static_cast<const unsigned char *>(plan_device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  return real_bytes == 8
             ? launch_p2_load<double>(coords, quad_order, plan, plan_layout_host, fq, n_elems, out, n_dofs, s)
             : launch_p2_load<float>(coords, quad_order, plan, plan_layout_host, fq, n_elems, out, n_dofs, s);
}

}  // extern "C"
