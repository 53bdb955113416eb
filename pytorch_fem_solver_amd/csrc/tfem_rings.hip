// P1 alpha * stiffness + beta * mass over a ring plan (tfem_rings_host.cpp): owner-computes
// ROW form of the element loop.  One lane owns one CSR row (= vertex v).  Its record lists
// the neighbours n_0 .. n_{k-1} of v in fan order as tile-local ids; the triangle of slot i is
// (v, n_i, n_next) with next = i + 1 (or 0 behind slot k - 1), flagged with the orientation it
// has in the connectivity.  With e_i = x(n_i) - x(v), d = e_next - e_i and the signed
// determinant det = +-(e_i x e_next) (element_tri.py:139) the P1 entries of row v are
// (basis.py:87-88, element_tri.py:41,132-145, abstract_basis.py:83; gradients are constant on
// the element, so sum_q w_q/2 = W is folded in)
//     K[v][v]      += (W / det) d.d
//     K[v][n_i]    -= (W / det) d.e_next
//     K[v][n_next] += (W / det) d.e_i
// and the mass part adds det * M_ii resp. det * M_ij (M = sum_q (w_q/2) l_i l_j).  Every
// triangle is evaluated by its three rows (three times the arithmetic of the element form)
// in exchange for: no atomics, no accumulators shared between lanes, one barrier per tile.
// The row's entries stay in registers; they are permuted into CSR order through a per-wave
// LDS stage (plain stores) and leave as lane-contiguous global stores.
//
// HBM traffic per element (N_v = N_T / 2): row records 8 B + row offsets 2 B + vertex ids
// ~2.7 B + coordinates ~10.5 B (halo re-reads included) + values 28 B ~ 51 B, against 48 B
// algorithmic.
#include <hip/hip_runtime.h>

#include <cstring>

#include "tfem_common.hpp"

// Reassociation is harmless here (see above: the row form is not the reference's operation
// order anyway; parity is asserted at 1e-12 against the oracle).
#pragma clang fp contract(fast)

namespace tfem {

constexpr int kRingBlock = 256;             // lanes per workgroup = owned rows per tile
constexpr int kRingWaves = kRingBlock / 64;
constexpr int kRingVertCap = 1024;          // 10-bit local ids

typedef unsigned int ru32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int ru32x4 __attribute__((ext_vector_type(4)));
using ring_rsrc_t = __amdgpu_buffer_rsrc_t;

__device__ __forceinline__ ring_rsrc_t ring_rsrc(const void *p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, int(bytes), 0x00020000);
}

template <typename T>
struct RingArgs {
  const T *coords;
  const unsigned char *plan;
  T *vals;
  unsigned coords_bytes, plan_bytes, vals_bytes;
  unsigned off_desc, off_rows, off_rowstart, off_gid;
  int n_tiles;
  int lds_vert;   // vertex slots reserved in LDS
  T stiff_w;      // alpha * sum_q w_q / 2
  T mass_d, mass_o;  // beta * sum_q (w_q/2) l_i l_i, beta * sum_q (w_q/2) l_i l_j (i != j)
};

typedef const int32_t __attribute__((address_space(4))) *ring_const_i32;

template <typename T>
__device__ __forceinline__ T fast_rcp(T x) {
  if constexpr (sizeof(T) == 8) {
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(r, e, r);
  } else {
    float r = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r, 1.0f);
    return __builtin_fmaf(r, e, r);
  }
}

// Field accessors of a row record (bit layout: tfem_rings_host.cpp).
template <int SLOTS>
struct RingRec {
  static constexpr int kWords = SLOTS == 7 ? 4 : 8;
  uint32_t w[kWords];
  __device__ __forceinline__ uint32_t id(int i) const { return (w[i / 3] >> (10 * (i % 3))) & 0x3FFu; }
  __device__ __forceinline__ int k() const {
    return SLOTS == 7 ? int((w[0] >> 30) | (((w[1] >> 30) & 1u) << 2))
                      : int((w[0] >> 30) | ((w[1] >> 30) << 2));
  }
  __device__ __forceinline__ int dpos() const {
    return SLOTS == 7 ? int((w[2] >> 24) & 7u) : int((w[2] >> 30) | ((w[3] >> 30) << 2));
  }
  __device__ __forceinline__ uint32_t flag(int i) const {
    return SLOTS == 7 ? (w[2] >> (10 + 2 * i)) & 3u : (w[SLOTS == 7 ? 0 : 5] >> (2 * i)) & 3u;
  }
  __device__ __forceinline__ int pos(int i) const {
    if (SLOTS == 7) return int((w[3] >> (3 * i)) & 7u);
    return i < 8 ? int((w[SLOTS == 7 ? 0 : 6] >> (4 * (i & 7))) & 15u)
                 : int((w[SLOTS == 7 ? 0 : 7] >> (4 * (i & 7))) & 15u);
  }
};

template <typename T>
__device__ __forceinline__ void lds_xy(const T *xy, uint32_t lid, T &x, T &y) {
  const T *p = xy + 2 * lid;  // one ds_read_b128 (double) / ds_read_b64 (float)
  x = p[0];
  y = p[1];
}

// The row of local vertex `lv`: entries of the neighbour slots in off[], the diagonal in diag.
template <typename T, int SLOTS, bool MASS>
__device__ __forceinline__ void ring_row(const RingArgs<T> &a, const RingRec<SLOTS> &rec,
                                         uint32_t lv, const T *xy, T (&off)[SLOTS], T &diag) {
  const int k = rec.k();
  T xv, yv;
  lds_xy(xy, lv, xv, yv);
  T px, py;
  lds_xy(xy, rec.id(0), px, py);
  const T e0x = px - xv, e0y = py - yv;
  T ecx = e0x, ecy = e0y;
  diag = T(0);
#pragma unroll
  for (int i = 0; i < SLOTS; ++i) off[i] = T(0);
#pragma unroll
  for (int i = 0; i < SLOTS; ++i) {
    const bool wrap = i + 1 == k;  // the triangle of slot k - 1 closes the fan on slot 0
    T rx = e0x, ry = e0y;          // e of slot i + 1 as stored (slot SLOTS does not exist)
    if (i + 1 < SLOTS) {
      lds_xy(xy, rec.id(i + 1), px, py);
      rx = px - xv;
      ry = py - yv;
    }
    const T enx = wrap ? e0x : rx, eny = wrap ? e0y : ry;
    const uint32_t flag = rec.flag(i);
    const bool has = i < k && flag != 0u;
    const T dx = enx - ecx, dy = eny - ecy;
    const T cross = ecx * eny - ecy * enx;
    const T sdet = flag == 1u ? cross : -cross;  // signed determinant of the stored element
    const T cs = has ? a.stiff_w * fast_rcp<T>(sdet) : T(0);
    const T dd = dx * dx + dy * dy;
    const T dn = dx * enx + dy * eny;
    const T dc = dx * ecx + dy * ecy;
    diag = diag + cs * dd;
    off[i] = off[i] - cs * dn;
    T nxt = cs * dc;
    if (MASS) {
      const T m = has ? sdet : T(0);
      diag = diag + a.mass_d * m;
      off[i] = off[i] + a.mass_o * m;
      nxt = nxt + a.mass_o * m;
    }
    if (i + 1 < SLOTS) off[i + 1] = off[i + 1] + (wrap ? T(0) : nxt);
    off[0] = off[0] + (wrap ? nxt : T(0));
    ecx = rx;
    ecy = ry;
  }
}

// LDS stage pitch per row (in entries): SLOTS + 2 is odd in units of 8 bytes, which spreads
// the 16 lanes of a ds_write_b64 group over all banks.
template <int SLOTS>
constexpr int ring_pitch() { return SLOTS + 2; }

// Row entries -> CSR order in the wave's stage -> global memory, lanes along the CSR array.
// `stage` = this wave's 64 * pitch entries; `rowstart`, `len` = this lane's row.
template <typename T, int SLOTS>
__device__ __forceinline__ void ring_flush(const RingRec<SLOTS> &rec, const T (&off)[SLOTS], T diag,
                                           T *stage, int rowstart, int len, ring_rsrc_t r_vals) {
  constexpr int kPitch = ring_pitch<SLOTS>();
  constexpr int kStride = SLOTS + 1;  // entries per row slot when reading back: 8 or 16
  const int lane = threadIdx.x & 63;
  const int k = rec.k();
  T *mine = stage + lane * kPitch;
#pragma unroll
  for (int i = 0; i < SLOTS; ++i)
    if (i < k) mine[rec.pos(i)] = off[i];
  if (k > 0) mine[rec.dpos()] = diag;
  __builtin_amdgcn_wave_barrier();  // same wave: LDS executes its operations in order
  constexpr int kRowsPerStep = 64 / kStride;
  const int sub = lane / kStride, p = lane % kStride;
#pragma unroll
  for (int u = 0; u < kStride; ++u) {
    const int r = u * kRowsPerStep + sub;  // row of this wave
    const T v = stage[r * kPitch + p];
    const int rs = __shfl(rowstart, r, 64);
    const int ln = __shfl(len, r, 64);
    if (p < ln) {
      const unsigned byte = unsigned(rs + p) * unsigned(sizeof(T));
      if constexpr (sizeof(T) == 8)
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(ru32x2, v), r_vals, byte, 0, 0);
      else
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r_vals, byte, 0, 0);
    }
  }
  __builtin_amdgcn_wave_barrier();
}

template <typename T>
__device__ __forceinline__ void ring_load_xy(ring_rsrc_t r, unsigned gid, T &x, T &y) {
  if constexpr (sizeof(T) == 8) {
    const ru32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, gid * 16u, 0, 0);
    x = __builtin_bit_cast(double, ru32x2{v.x, v.y});
    y = __builtin_bit_cast(double, ru32x2{v.z, v.w});
  } else {  // two dword loads: raw_buffer_load_b64 is miscompiled by this hipcc (tfem_tiles.hip)
    x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, gid * 8u, 0, 0));
    y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, gid * 8u + 4u, 0, 0));
  }
}

template <int SLOTS>
__device__ __forceinline__ void ring_load_rec(ring_rsrc_t r, unsigned byte, RingRec<SLOTS> &rec) {
  const ru32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte, 0, 0);
  rec.w[0] = v.x;
  rec.w[1] = v.y;
  rec.w[2] = v.z;
  rec.w[3] = v.w;
  if constexpr (SLOTS == 15) {
    const ru32x4 u = __builtin_amdgcn_raw_buffer_load_b128(r, byte + 16u, 0, 0);
    rec.w[4] = u.x;
    rec.w[5] = u.y;
    rec.w[6] = u.z;
    rec.w[7] = u.w;
  }
}

// ---------------------------------------------------------------------------------------
// One tile per workgroup.  Workgroup b works on tile (b & 7) * per + (b >> 3): consecutive
// workgroups go to different XCDs, so every XCD (own L2) walks one contiguous piece of the
// Z-order curve and neighbouring tiles share their halo coordinates in that L2.
// ---------------------------------------------------------------------------------------
template <typename T, int SLOTS, bool MASS>
__global__ __launch_bounds__(kRingBlock) void k_p1_rings(const RingArgs<T> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ring_smem[];
  T *xy = reinterpret_cast<T *>(ring_smem);                      // [2 * lds_vert]
  T *stage = xy + 2 * a.lds_vert;                                // [waves][64 * pitch]
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int per = (a.n_tiles + 7) / 8;
  const int tile = int(blockIdx.x & 7) * per + int(blockIdx.x >> 3);
  if (tile >= a.n_tiles || int(blockIdx.x >> 3) >= per) return;
  ring_const_i32 d = (ring_const_i32)(uintptr_t)(a.plan + a.off_desc + 16u * unsigned(tile));
  const int vert_off = d[0], n_vert = d[1], n_own = d[2], row_off = d[3];
  const ring_rsrc_t r_coords = ring_rsrc(a.coords, a.coords_bytes);
  const ring_rsrc_t r_plan = ring_rsrc(a.plan, a.plan_bytes);
  const ring_rsrc_t r_vals = ring_rsrc(a.vals, a.vals_bytes);

  // lanes without a row read the zero record behind the plan (k = 0: nothing is stored)
  RingRec<SLOTS> rec;
  const unsigned row = tid < n_own ? unsigned(row_off + tid) : 0x3FFFFFFu;
  ring_load_rec<SLOTS>(r_plan, a.off_rows + row * unsigned(4 * RingRec<SLOTS>::kWords), rec);
  const int rowstart = int(__builtin_amdgcn_raw_buffer_load_b32(r_plan, a.off_rowstart + row * 4u, 0, 0));
  for (int l = tid; l < n_vert; l += kRingBlock) {
    const unsigned g = __builtin_amdgcn_raw_buffer_load_b32(r_plan, a.off_gid + unsigned(vert_off + l) * 4u, 0, 0);
    T x, y;
    ring_load_xy<T>(r_coords, g, x, y);
    xy[2 * l] = x;
    xy[2 * l + 1] = y;
  }
  __syncthreads();
  T off[SLOTS], diag;
  ring_row<T, SLOTS, MASS>(a, rec, unsigned(tid < n_own ? tid : 0), xy, off, diag);
  const int k = rec.k();
  ring_flush<T, SLOTS>(rec, off, diag, stage + wave * 64 * ring_pitch<SLOTS>(), rowstart,
                       k > 0 ? k + 1 : 0, r_vals);
}

struct RingLaunch {
  const void *coords;
  int quad_order;
  double alpha, beta;
  const unsigned char *plan;
  const int64_t *layout;
  int64_t n_verts, nnz;
  void *vals;
  hipStream_t stream;
};

template <typename T>
static int launch_rings(const RingLaunch &L) {
  TriTables tables;
  if (!build_tri_tables(L.quad_order, int(sizeof(T)), &tables))
    return fail(TFEM_ERR_UNSUPPORTED, "Integration order not implemented");
  const int64_t *z = L.layout;
  if (z[0] == 0) return TFEM_OK;
  if (!L.coords || !L.plan || !L.vals) return fail(TFEM_ERR_INVALID_ARGUMENT, "NULL pointer");
  if (z[0] < 0 || z[4] > kRingBlock || z[3] > kRingVertCap || z[4] > z[3] ||
      !((z[6] == 7 && z[7] == 4) || (z[6] == 15 && z[7] == 8)) || z[5] > z[6] + 1)
    return fail(TFEM_ERR_INVALID_ARGUMENT, "ring plan exceeds the kernel's capacities");
  RingArgs<T> a;
  std::memset(&a, 0, sizeof(a));
  a.coords = static_cast<const T *>(L.coords);
  a.plan = L.plan;
  a.vals = static_cast<T *>(L.vals);
  const int64_t rb = int64_t(sizeof(T));
  const int64_t extents[3] = {L.n_verts * 2 * rb, z[12], L.nnz * rb};
  for (int64_t e : extents)
    if (e < 0 || e >= (int64_t(1) << 32))
      return fail(TFEM_ERR_INDEX_RANGE, "an array of %lld bytes does not fit the 32-bit offsets "
                  "of the ring kernel", (long long)e);
  a.coords_bytes = unsigned(extents[0]);
  a.plan_bytes = unsigned(extents[1]);
  a.vals_bytes = unsigned(extents[2]);
  a.off_desc = unsigned(z[8]);
  a.off_rows = unsigned(z[9]);
  a.off_rowstart = unsigned(z[10]);
  a.off_gid = unsigned(z[11]);
  a.n_tiles = int(z[0]);
  a.lds_vert = (int(z[3]) + 1) & ~1;
  // W = sum_q w_q/2 and M_ij = sum_q (w_q/2) l_i l_j, formed in T in quadrature order.  The
  // rules of element_tri.py:77-130 are symmetric, so M has one diagonal and one off-diagonal
  // value (up to rounding: entries 00 and 01 are used).
  T w = T(0), md = T(0), mo = T(0);
  for (int q = 0; q < tables.nq; ++q) {
    w = w + T(tables.hw[q]);
    md = md + T(tables.hw[q]) * (T(tables.lam[q][0]) * T(tables.lam[q][0]));
    mo = mo + T(tables.hw[q]) * (T(tables.lam[q][0]) * T(tables.lam[q][1]));
  }
  a.stiff_w = T(L.alpha) * w;
  a.mass_d = T(L.beta) * md;
  a.mass_o = T(L.beta) * mo;
  const bool mass = L.beta != 0.0;
  const int slots = int(z[6]);
  const size_t lds = size_t(2 * a.lds_vert) * sizeof(T) +
                     size_t(kRingWaves * 64 * (slots + 2)) * sizeof(T);
  void *kernel = nullptr;
  if (slots == 7)
    kernel = mass ? reinterpret_cast<void *>(k_p1_rings<T, 7, true>)
                  : reinterpret_cast<void *>(k_p1_rings<T, 7, false>);
  else
    kernel = mass ? reinterpret_cast<void *>(k_p1_rings<T, 15, true>)
                  : reinterpret_cast<void *>(k_p1_rings<T, 15, false>);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
    if (e != hipSuccess) return fail(TFEM_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  const int per = int((z[0] + 7) / 8);
  const dim3 grid{unsigned(per * 8)}, block{unsigned(kRingBlock)};
  void *params[] = {&a};
  hipError_t e = hipLaunchKernel(kernel, grid, block, params, lds, L.stream);
  if (e != hipSuccess) return fail(TFEM_ERR_HIP, "ring kernel launch: %s", hipGetErrorString(e));
  return TFEM_OK;
}

}  // namespace tfem

extern "C" {

int tfem_ring_capacity(int what) {
  using namespace tfem;
  switch (what) {
    case 0: return kRingBlock;
    case 1: return kRingVertCap;
    default: return 0;
  }
}

int tfem_p1_assemble_rings(const void *coords, int real_bytes, int64_t n_verts, int quad_order,
                           double alpha, double beta, const void *plan_device,
                           const int64_t *plan_layout_host, void *vals, int64_t nnz,
                           void *stream) {
  using namespace tfem;
  if (real_bytes != 4 && real_bytes != 8)
    return fail(TFEM_ERR_INVALID_ARGUMENT, "real_bytes must be 4 or 8");
  if (!plan_layout_host) return fail(TFEM_ERR_INVALID_ARGUMENT, "plan_layout_host is NULL");
  RingLaunch L{coords, quad_order, alpha, beta, static_cast<const unsigned char *>(plan_device),
               plan_layout_host, n_verts, nnz, vals, static_cast<hipStream_t>(stream)};
  return real_bytes == 8 ? launch_rings<double>(L) : launch_rings<float>(L);
}

}  // extern "C"
