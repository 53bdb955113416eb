// P1 bilinear assembly over a tile plan (tfem_tiles_host.cpp): the headline kernel.
//
// One 256-lane workgroup per tile:
//   1. every lane prefetches its <= kElemPerLane 12-byte element records (coalesced), the
//      tile's vertex coordinates are gathered into LDS (16 B per vertex, each read once
//      per tile) and the tile's CSR-row accumulators in LDS are zeroed;
//   2. per element: coordinates from LDS, Jacobian / signed det / inverse / gradients and
//      the 3x3 block in the reference's operation order, then ds_add_f64 of the entries
//      whose ROW this tile owns (LDS atomics only; nothing global);
//   3. the owned rows are streamed to the CSR value array, 8 lanes per row, each value
//      written exactly once with a plain store (rows are sorted by global id, so lanes
//      of a wave write runs of consecutive rows = contiguous bytes).
// HBM traffic per element ~ 13 B records + ~12 B coordinates/ids + ~3 B row info + 28 B
// values, against 48 B algorithmic.
#include <hip/hip_runtime.h>

#include <cstring>

#include "tfem_common.hpp"

#pragma clang fp contract(off)

namespace tfem {

constexpr int kTileBlock = 256;
constexpr int kElemPerLane = 5;  // tile element capacity = 1280

template <typename T>
struct TileArgs {
  const T *coords;
  const int32_t *desc;
  const uint32_t *records;
  const int32_t *vert_gid;
  const int32_t *row_gstart;
  const uint16_t *row_loff;
  T *vals;
  int n_tiles;
  int lds_acc;   // accumulator entries reserved in LDS
  int lds_vert;  // vertex slots reserved in LDS
  T alpha, beta;
  T hw[kMaxQuad];
  T lam[kMaxQuad][3];
};

// Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share one); give each XCD a
// contiguous range of the Z-ordered tiles so neighbouring tiles share its L2.
__device__ __forceinline__ int tile_of_block(int b, int n_tiles) {
  const int per = (n_tiles + 7) / 8;
  const int t = (b & 7) * per + (b >> 3);
  return t;
}

template <typename T, int Q>
__global__ __launch_bounds__(kTileBlock) void k_p1_bilinear_tiles(const TileArgs<T> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T *acc = reinterpret_cast<T *>(smem_raw);
  T *xy = acc + a.lds_acc;                                         // 2 per vertex
  int *loff = reinterpret_cast<int *>(xy + 2 * a.lds_vert);        // n_own + 1

  const int tile = tile_of_block(blockIdx.x, a.n_tiles);
  if (tile >= a.n_tiles) return;  // whole workgroup, before any barrier
  const int32_t *d = a.desc + 8 * tile;
  const int elem_off = d[0], n_elem = d[1], vert_off = d[2], n_vert = d[3], n_own = d[4];
  const int row_off = d[5], acc_size = d[6], loff_off = d[7];
  const int tid = threadIdx.x;

  // ---- 1a. prefetch this lane's element records ---------------------------------------
  uint32_t rec[kElemPerLane][3];
#pragma unroll
  for (int k = 0; k < kElemPerLane; ++k) {
    const int idx = tid + k * kTileBlock;
    if (idx < n_elem) {
      const uint32_t *r = a.records + 3 * size_t(elem_off + idx);
      rec[k][0] = r[0];
      rec[k][1] = r[1];
      rec[k][2] = r[2];
    }
  }
  // ---- 1b. gather coordinates, clear accumulators, stage row offsets --------------------
  for (int l = tid; l < n_vert; l += kTileBlock) {
    const int64_t g = a.vert_gid[vert_off + l];
    xy[2 * l] = a.coords[2 * g];
    xy[2 * l + 1] = a.coords[2 * g + 1];
  }
  for (int s = tid; s < acc_size; s += kTileBlock) acc[s] = T(0);
  for (int l = tid; l <= n_own; l += kTileBlock) loff[l] = a.row_loff[loff_off + l];
  __syncthreads();

  // ---- 2. element blocks -> LDS accumulators ------------------------------------------
#pragma unroll
  for (int k = 0; k < kElemPerLane; ++k) {
    const int idx = tid + k * kTileBlock;
    if (idx >= n_elem) break;
    int lid[3];
    T x[3], y[3];
#pragma unroll
    for (int v = 0; v < 3; ++v) {
      lid[v] = int(rec[k][v] & 0xFFFu);
      x[v] = xy[2 * lid[v]];
      y[v] = xy[2 * lid[v] + 1];
    }
    // reference operation order: basis.py:87-88, element_tri.py:132-145, :41
    const T ja = x[1] - x[0], jb = x[2] - x[0];
    const T jc = y[1] - y[0], jd = y[2] - y[0];
    const T det = ja * jd - jb * jc;
    const T r = T(1) / det;
    const T i00 = r * jd, i01 = r * (-jb), i10 = r * (-jc), i11 = r * ja;
    T g[3][2];
    g[0][0] = (-i00) + (-i10);
    g[0][1] = (-i01) + (-i11);
    g[1][0] = i00;
    g[1][1] = i01;
    g[2][0] = i10;
    g[2][1] = i11;
    T dx[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) dx[q] = a.hw[q] * det;
    T loc[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = i; j < 3; ++j) {
        const T s = a.alpha * (g[i][0] * g[j][0] + g[i][1] * g[j][1]);
        T sum = T(0);
#pragma unroll
        for (int q = 0; q < Q; ++q) sum = sum + (s + a.beta * (a.lam[q][i] * a.lam[q][j])) * dx[q];
        loc[i][j] = sum;
        loc[j][i] = sum;
      }
    }
    // local[i][j] -> A[row v_j][col v_i] (basis.py:73-76); only rows owned by this tile
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (lid[j] < n_own) {
        const int base = loff[lid[j]];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const int pos = int((rec[k][j] >> (12 + 4 * i)) & 0xFu);
          atomicAdd(&acc[base + pos], loc[i][j]);  // ds_add_f64
        }
      }
    }
  }
  __syncthreads();

  // ---- 3. stream the owned rows out: 8 lanes per row ------------------------------------
  const int lane8 = tid & 7;
  for (int row = tid >> 3; row < n_own; row += kTileBlock / 8) {
    const int start = loff[row];
    const int len = loff[row + 1] - start;
    T *out = a.vals + a.row_gstart[row_off + row];
    for (int c = lane8; c < len; c += 8) out[c] = acc[start + c];
  }
}

template <typename T>
static int launch_tiles(const void *coords, int quad_order, double alpha, double beta,
                        const int32_t *desc, int64_t n_tiles, const uint32_t *records,
                        const int32_t *vert_gid, const int32_t *row_gstart,
                        const uint16_t *row_loff, int max_n_vert, int max_n_own, int max_acc,
                        void *vals, hipStream_t stream) {
  TriTables tables;
  if (!build_tri_tables(quad_order, int(sizeof(T)), &tables))
    return fail(TFEM_ERR_UNSUPPORTED, "Integration order not implemented");
  if (n_tiles == 0) return TFEM_OK;
  if (!coords || !desc || !records || !vert_gid || !row_gstart || !row_loff || !vals)
    return fail(TFEM_ERR_INVALID_ARGUMENT, "NULL pointer");
  TileArgs<T> a;
  std::memset(&a, 0, sizeof(a));
  a.coords = static_cast<const T *>(coords);
  a.desc = desc;
  a.records = records;
  a.vert_gid = vert_gid;
  a.row_gstart = row_gstart;
  a.row_loff = row_loff;
  a.vals = static_cast<T *>(vals);
  a.n_tiles = int(n_tiles);
  a.lds_acc = (max_acc + 1) & ~1;
  a.lds_vert = max_n_vert;
  a.alpha = T(alpha);
  a.beta = T(beta);
  for (int q = 0; q < kMaxQuad; ++q) {
    a.hw[q] = T(tables.hw[q]);
    for (int i = 0; i < 3; ++i) a.lam[q][i] = T(tables.lam[q][i]);
  }
  const size_t lds = size_t(a.lds_acc) * sizeof(T) + size_t(2 * a.lds_vert) * sizeof(T) +
                     size_t(max_n_own + 1) * sizeof(int);
  if (lds > 160 * 1024) return fail(TFEM_ERR_INVALID_ARGUMENT, "tile needs %zu B of LDS", lds);
  const int per = int((n_tiles + 7) / 8);
  const dim3 grid(unsigned(per * 8)), block(kTileBlock);
#define TFEM_TILE_LAUNCH(QQ)                                                                    \
  {                                                                                             \
    auto kernel = k_p1_bilinear_tiles<T, QQ>;                                                   \
    if (lds > 64 * 1024) {                                                                      \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),                \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)); \
      if (e != hipSuccess)                                                                      \
        return fail(TFEM_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));             \
    }                                                                                           \
    hipLaunchKernelGGL(kernel, grid, block, lds, stream, a);                                    \
  }
  switch (tables.nq) {
    case 1: TFEM_TILE_LAUNCH(1) break;
    case 3: TFEM_TILE_LAUNCH(3) break;
    case 4: TFEM_TILE_LAUNCH(4) break;
    case 6: TFEM_TILE_LAUNCH(6) break;
    default: return fail(TFEM_ERR_UNSUPPORTED, "Integration order not implemented");
  }
#undef TFEM_TILE_LAUNCH
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(TFEM_ERR_HIP, "tile kernel launch: %s", hipGetErrorString(e));
  return TFEM_OK;
}

}  // namespace tfem

extern "C" {

int tfem_tile_elem_capacity(void) { return tfem::kTileBlock * tfem::kElemPerLane; }

int tfem_p1_bilinear_tiles(const void *coords, int real_bytes, int quad_order, double alpha,
                           double beta, const int32_t *desc, int64_t n_tiles,
                           const uint32_t *records, const int32_t *vert_gid,
                           const int32_t *row_gstart, const uint16_t *row_loff, int max_n_elem,
                           int max_n_vert, int max_n_own, int max_acc, void *vals, void *stream) {
  using namespace tfem;
  if (real_bytes != 4 && real_bytes != 8)
    return fail(TFEM_ERR_INVALID_ARGUMENT, "real_bytes must be 4 or 8");
  if (n_tiles < 0 || max_n_elem > kTileBlock * kElemPerLane || max_n_vert > 4096 ||
      max_acc > 65535 || max_n_own > max_n_vert)
    return fail(TFEM_ERR_INVALID_ARGUMENT, "tile plan exceeds the kernel's capacities");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (real_bytes == 8)
    return launch_tiles<double>(coords, quad_order, alpha, beta, desc, n_tiles, records, vert_gid,
                                row_gstart, row_loff, max_n_vert, max_n_own, max_acc, vals, s);
  return launch_tiles<float>(coords, quad_order, alpha, beta, desc, n_tiles, records, vert_gid,
                             row_gstart, row_loff, max_n_vert, max_n_own, max_acc, vals, s);
}

}  // extern "C"
