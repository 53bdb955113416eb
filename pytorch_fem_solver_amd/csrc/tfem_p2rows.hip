// P2 (quadratic) alpha * stiffness + beta * mass over a P2 row plan (tfem_p2rows_host.cpp):
// owner-computes ROW form, one lane per CSR row, no atomics -- the P2 counterpart of
// tfem_rings.hip.
//
// On an affine triangle with frame (p0, p1, p2), e1 = p1 - p0, e2 = p2 - p0, det = e1 x e2
// (signed, element_tri.py:139) the physical gradient of a P2 shape function is
// rgrad_a(q) @ J^-1 = rg_a0 grad(l_1) + rg_a1 grad(l_2) (element_tri.py:43-70), so the element
// block (abstract_basis.py:83) is a CONSTANT linear map of three numbers:
//     K_ab = A_ab G11 + B_ab G12 + D_ab G22 + (beta M_ab) det
//     G11 = |e2|^2 / det, G12 = -(e1.e2) / det, G22 = |e1|^2 / det
//     A_ab = sum_q w_q/2 rg_a0 rg_b0, B_ab = sum_q w_q/2 (rg_a0 rg_b1 + rg_a1 rg_b0),
//     D_ab = sum_q w_q/2 rg_a1 rg_b1, M_ab = sum_q w_q/2 phi_a phi_b
// with the tables formed once on the host in the real type from the reference's own shape
// function tables.  A row needs one row of that block per incident triangle: row 0 (the vertex
// at p0) for a vertex DoF, row 3 (the edge (p0, p1)) for an edge DoF -- the frame is rotated so
// that the row's DoF sits there; the shape functions and the quadrature rules are symmetric
// under that relabelling, so this is the stored element's block up to rounding (parity is
// asserted at 1e-12 against the oracle).
//
// One tile per 256-lane workgroup: coordinates of the tile's vertices -> LDS, one barrier, then
// every lane walks its row, stages the entries in CSR order in its wave's LDS stage and the
// wave streams them out with 16-byte stores (a wave's rows are consecutive DoFs: one
// contiguous piece of the CSR array whose offset is in the descriptor).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <mutex>

#include "tfem_common.hpp"
#include "tfem_rowkit.hpp"

#pragma clang fp contract(fast)

namespace tfem {

constexpr int kP2Block = 256;
constexpr int kP2Waves = kP2Block / 64;
constexpr int kP2VertexRowMax = 22;  // 1 + 3 * 7
constexpr int kP2EdgeRowMax = 9;

template <typename T>
struct P2RowArgs {
  const T *coords;
  const unsigned char *plan;
  T *vals;
  unsigned coords_bytes, plan_bytes, vals_bytes;
  unsigned off_desc, off_rows, off_gid;
  int n_tiles;
  int lds_vert;
  int xcd_ranges;  // 1: every XCD works on one contiguous range of the tile list, 0: tile = workgroup
  // row 0 (vertex kinds) or row 3 (edge kind) of the constant maps, alpha / beta folded in
  T ca[6], cb[6], cd[6], cm[6];
};

// entries r[0..6) of one row of the element block in the frame whose edge vectors are e1, e2
template <typename T, bool MASS>
__device__ __forceinline__ void p2_block_row(const T (&ca)[6], const T (&cb)[6], const T (&cd)[6], const T (&cm)[6],
                                             T q1, T q2, T p, T cross, uint32_t flag, T (&r)[6]);

template <typename T, bool MASS>
__device__ __forceinline__ void p2_block_row(const P2RowArgs<T> &a, T q1, T q2, T p, T cross,
                                             uint32_t flag, T (&r)[6]) {
  p2_block_row<T, MASS>(a.ca, a.cb, a.cd, a.cm, q1, q2, p, cross, flag, r);
}

template <typename T, bool MASS>
__device__ __forceinline__ void p2_block_row(const T (&ca)[6], const T (&cb)[6], const T (&cd)[6], const T (&cm)[6],
                                             T q1, T q2, T p, T cross, uint32_t flag, T (&r)[6]) {
  // flag: 0 no triangle (all zero), 1 frame = (e1, e2) as given, 2 frame = (e2, e1)
  const T c = flag_weight<T>(T(1), flag) * fast_rcp<T>(flag ? cross : T(1));  // 1 / det or 0
  const T g11 = c * (flag == 2u ? q1 : q2);
  const T g12 = -(c * p);
  const T g22 = c * (flag == 2u ? q2 : q1);
  const T det = flag_weight<T>(T(1), flag) * cross;
#pragma unroll
  for (int m = 0; m < 6; ++m) {
    T v = ca[m] * g11 + cb[m] * g12 + cd[m] * g22;
    if (MASS) v = v + cm[m] * det;
    r[m] = v;
  }
}

// The wave's stage -> global memory: the wave's rows are consecutive DoFs, so stage index +
// delta = CSR index.  Lane j of step u takes entries 128 u + 2 j and the next one.
template <typename T, int MAXLEN, int AUX = 0>
__device__ __forceinline__ void p2_store(const T *stage, int total, int delta, ring_rsrc_t r_vals) {
  const int lane = threadIdx.x & 63;
  constexpr int kSteps = (64 * MAXLEN + 127) / 128;
#pragma unroll
  for (int u = 0; u < kSteps; ++u) {
    if (128 * u < total) {  // wave-uniform
      const int s0 = 128 * u + 2 * lane;
      const T v0 = stage[s0], v1 = stage[s0 + 1];
      const unsigned byte = unsigned(s0 + delta) * unsigned(sizeof(T));
      if (128 * (u + 1) <= total || s0 + 1 < total) {
        if constexpr (sizeof(T) == 8) {
          const ru32x2 x = __builtin_bit_cast(ru32x2, v0), y = __builtin_bit_cast(ru32x2, v1);
          __builtin_amdgcn_raw_buffer_store_b128(ru32x4{x.x, x.y, y.x, y.y}, r_vals, byte, 0, AUX);
        } else {
          __builtin_amdgcn_raw_buffer_store_b64(
              ru32x2{__builtin_bit_cast(unsigned, v0), __builtin_bit_cast(unsigned, v1)}, r_vals, byte, 0, AUX);
        }
      } else if (s0 < total) {
        if constexpr (sizeof(T) == 8)
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(ru32x2, v0), r_vals, byte, 0, AUX);
        else
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v0), r_vals, byte, 0, AUX);
      }
    }
  }
}

// The same when some rows of the wave are LONG rows (written by k_p2_long_rows): their entries
// are not in the stage, so the CSR position of the rows behind them is further on.  `pre` =
// stage index of the lane's row, `csr` = CSR offset of the lane's row, `is_long` marks the
// holes: the stage is streamed out piece by piece between them.
// [lane_lo, lane_hi): the lanes whose rows are in the stage (the persistent kernel stages a wave's
// vertex rows in two halves); `pre` is relative to the first of them.
template <typename T>
__device__ __forceinline__ void p2_store_pieces(const T *stage, int total, int pre, int csr, bool is_long,
                                                ring_rsrc_t r_vals, int lane_lo = 0, int lane_hi = 64) {
  const int lane = threadIdx.x & 63;
  unsigned long long holes = __ballot(is_long && lane >= lane_lo && lane < lane_hi);
  int first_lane = lane_lo;  // first lane of the current piece
  for (;;) {
    const int stop_lane = holes ? __builtin_ctzll(holes) : lane_hi;  // the piece: lanes [first_lane, stop_lane)
    if (first_lane < lane_hi && first_lane < stop_lane) {
      const int b = __builtin_amdgcn_readlane(pre, first_lane);
      const int e = stop_lane < lane_hi ? __builtin_amdgcn_readlane(pre, stop_lane) : total;
      const int delta = __builtin_amdgcn_readlane(csr, first_lane) - b;
      for (int s0 = b + lane; s0 < e; s0 += 64) {
        const unsigned byte = unsigned(s0 + delta) * unsigned(sizeof(T));
        if constexpr (sizeof(T) == 8)
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(ru32x2, stage[s0]), r_vals, byte, 0, 0);
        else
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, stage[s0]), r_vals, byte, 0, 0);
      }
    }
    if (!holes) break;
    holes &= holes - 1;
    first_lane = stop_lane + 1;
  }
}

// KIND 0: vertex rows, KIND 1: edge rows.  PASSES: a wave's rows go through its stage in that many
// parts of 64 / PASSES rows (vertex rows: 11.3 KB of stage per wave in one part -- three workgroups
// per CU; 2.8 KB in four parts -- six).
template <typename T, int KIND, bool MASS, int PASSES = 1>
__global__ __launch_bounds__(kP2Block) void k_p2_rows(const P2RowArgs<T> a) {
  constexpr int kMaxLen = KIND == 0 ? kP2VertexRowMax : kP2EdgeRowMax;
  constexpr int kStageEntries = (64 / PASSES) * kMaxLen + 2;
  static_assert(PASSES == 1 || KIND == 0, "edge rows are staged in one part");
  extern __shared__ __attribute__((aligned(16))) unsigned char p2_smem[];
  T *xy = reinterpret_cast<T *>(p2_smem);  // [2 * lds_vert]
  T *stage = xy + 2 * a.lds_vert;          // [waves][kStageEntries]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int per = (a.n_tiles + 7) / 8;
  const int tile = a.xcd_ranges ? int(blockIdx.x & 7) * per + int(blockIdx.x >> 3) : int(blockIdx.x);
  if (tile >= a.n_tiles || (a.xcd_ranges && int(blockIdx.x >> 3) >= per)) return;
  ring_const_i32 d = (ring_const_i32)(uintptr_t)(a.plan + a.off_desc + 64u * unsigned(tile));
  const int vert_off = d[0], n_vert = d[1], row_off = d[2], n_own = d[7];
  const int row0 = d[3 + wave], row1 = d[4 + wave], rs0 = d[12 + wave];
  const ring_rsrc_t r_coords = ring_rsrc(a.coords, a.coords_bytes);
  const ring_rsrc_t r_plan = ring_rsrc(a.plan, a.plan_bytes);
  const ring_rsrc_t r_vals = ring_rsrc(a.vals, a.vals_bytes);
  const int my_row = row0 + lane;
  const bool has_row = my_row < row1;
  constexpr unsigned kNone = 0x3FFFFFFu;
  constexpr int kWords = KIND == 0 ? 8 : 4;
  uint32_t w[kWords];
  {
    const unsigned byte = a.off_rows + (has_row ? unsigned(row_off + my_row) : kNone) * unsigned(4 * kWords);
    const ru32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r_plan, byte, 0, 0);
    w[0] = v.x;
    w[1] = v.y;
    w[2] = v.z;
    w[3] = v.w;
    if constexpr (KIND == 0) {
      const ru32x4 u = __builtin_amdgcn_raw_buffer_load_b128(r_plan, byte + 16u, 0, 0);
      w[4] = u.x;
      w[5] = u.y;
      w[6] = u.z;
      w[7] = u.w;
    }
  }
  // coordinates of the tile's vertices -> LDS.  Vertex tiles: the lane's own vertex is
  // gid0 + lane (consecutive ids, nothing read), the halo comes from the id list; edge tiles:
  // every vertex from the list.
  if (KIND == 0) {
    if (has_row) {
      T x, y;
      ring_load_xy<T>(r_coords, unsigned(d[8 + wave] + lane), x, y);
      xy[2 * my_row] = x;
      xy[2 * my_row + 1] = y;
    }
    for (int l = n_own + tid; l < n_vert; l += kP2Block) {
      const unsigned g = __builtin_amdgcn_raw_buffer_load_b32(r_plan, a.off_gid + unsigned(vert_off + l) * 4u, 0, 0);
      T x, y;
      ring_load_xy<T>(r_coords, g, x, y);
      xy[2 * l] = x;
      xy[2 * l + 1] = y;
    }
  } else {
    for (int l = tid; l < n_vert; l += kP2Block) {
      const unsigned g = __builtin_amdgcn_raw_buffer_load_b32(r_plan, a.off_gid + unsigned(vert_off + l) * 4u, 0, 0);
      T x, y;
      ring_load_xy<T>(r_coords, g, x, y);
      xy[2 * l] = x;
      xy[2 * l + 1] = y;
    }
  }
  __syncthreads();

  T *my_stage = stage + wave * kStageEntries;
  constexpr int kSpare = (64 / PASSES) * kMaxLen;
  int len, pre;
  if constexpr (KIND == 0) {
    // ---- vertex row: the fan, as in tfem_rings.hip ----------------------------------------
    const int k = int((w[2] >> 24) & 7u);
    auto id = [&](int i) { return (w[i / 3] >> (10 * (i % 3))) & 0x3FFu; };
    auto flag_of = [&](int i) { return (w[2] >> (10 + 2 * i)) & 3u; };
    auto field = [&](int f) { return int((w[3 + f / 6] >> (5 * (f % 6))) & 31u); };
    T xv, yv, px, py;
    lds_xy(xy, unsigned(has_row ? my_row : 0), xv, yv);
    const uint32_t id0 = id(0);
    lds_xy(xy, id0, px, py);
    T ecx = px - xv, ecy = py - yv;
    T qc = ecx * ecx + ecy * ecy;
    T diag = T(0), vcol[8], ecol[8], ocol[7];
#pragma unroll
    for (int i = 0; i < 8; ++i) vcol[i] = ecol[i] = T(0);
    int n_tri = 0;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const uint32_t idn = (i + 1 < 7 && i + 1 != k) ? id(i + 1 < 7 ? i + 1 : 0) : id0;
      lds_xy(xy, idn, px, py);
      const T enx = px - xv, eny = py - yv;
      const T qn = enx * enx + eny * eny;
      const T p = ecx * enx + ecy * eny;
      const T cross = ecx * eny - ecy * enx;
      const uint32_t flag = flag_of(i);  // 0 for every slot i >= k
      n_tri += flag ? 1 : 0;
      T r[6];
      p2_block_row<T, MASS>(a, qc, qn, p, cross, flag, r);
      // r: v, p1, p2, edge (v,p1), edge (p1,p2), edge (p2,v); (p1, p2) = (n_i, n_next) for
      // flag 1 and (n_next, n_i) for flag 2
      const bool fwd = flag != 2u;
      diag = diag + r[0];
      vcol[i] = vcol[i] + (fwd ? r[1] : r[2]);
      vcol[i + 1] = vcol[i + 1] + (fwd ? r[2] : r[1]);
      ecol[i] = ecol[i] + (fwd ? r[3] : r[5]);
      ecol[i + 1] = ecol[i + 1] + (fwd ? r[5] : r[3]);
      ocol[i] = r[4];
      ecx = enx;
      ecy = eny;
      qc = qn;
    }
    // what the closing triangle left in slot k belongs to slot 0
    T wv = vcol[1], we = ecol[1];
#pragma unroll
    for (int j = 2; j <= 7; ++j) {
      wv = k == j ? vcol[j] : wv;
      we = k == j ? ecol[j] : we;
    }
    vcol[0] = vcol[0] + wv;
    ecol[0] = ecol[0] + we;
    len = k > 0 ? 1 + 2 * k + n_tri : 0;
    const int incl = wave_inclusive_scan(len);
    pre = incl - len;
    // long rows (8 .. 15 neighbours, written by k_p2_long_rows): k = 0 here, their length only
    // moves the CSR position of the rows behind them
    const bool is_long = has_row && k == 0 && (w[3] >> 31) != 0u;
    const int true_len = is_long ? int(w[3] & 0x7FFFFFFFu) : len;
    const bool any_long = __ballot(is_long) != 0ull;
    if constexpr (PASSES == 1) {
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        my_stage[i < k ? pre + field(i) : kSpare] = vcol[i];
        my_stage[i < k ? pre + field(7 + i) : kSpare] = ecol[i];
        my_stage[(i < k && flag_of(i)) ? pre + field(14 + i) : kSpare] = ocol[i];
      }
      my_stage[k > 0 ? pre + int(w[2] >> 27) : kSpare] = diag;
      const int total = __builtin_amdgcn_readlane(incl, 63);
      __builtin_amdgcn_wave_barrier();
      if (!any_long) {
        p2_store<T, kMaxLen>(my_stage, total, rs0, r_vals);
      } else {
        const int csr = rs0 + wave_inclusive_scan(true_len) - true_len;
        p2_store_pieces<T>(my_stage, total, pre, csr, is_long, r_vals);
      }
    } else {
      constexpr int kPart = 64 / PASSES;  // rows per part
      const int csr = any_long ? rs0 + wave_inclusive_scan(true_len) - true_len : 0;
      int base = 0;  // entries in front of the part's first row
#pragma unroll
      for (int pass = 0; pass < PASSES; ++pass) {
        const bool mine = (lane / kPart) == pass;
        const int upto = __builtin_amdgcn_readlane(incl, kPart * (pass + 1) - 1);
        const int count = upto - base;
        const int at = pre - base;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
          my_stage[(mine && i < k) ? at + field(i) : kSpare] = vcol[i];
          my_stage[(mine && i < k) ? at + field(7 + i) : kSpare] = ecol[i];
          my_stage[(mine && i < k && flag_of(i)) ? at + field(14 + i) : kSpare] = ocol[i];
        }
        my_stage[(mine && k > 0) ? at + int(w[2] >> 27) : kSpare] = diag;
        __builtin_amdgcn_wave_barrier();
        if (!any_long)
          p2_store<T, (kPart * kMaxLen + 63) / 64>(my_stage, count, rs0 + base, r_vals);
        else
          p2_store_pieces<T>(my_stage, count, at, csr, is_long, r_vals, kPart * pass, kPart * pass + kPart);
        __builtin_amdgcn_wave_barrier();
        base = upto;
      }
    }
  } else {
    // ---- edge row: one or two triangles, each in its own stored frame ------------------------
    const bool has2 = (w[1] >> 10) & 1u;
    const bool rev = (w[1] >> 11) & 1u;
    T ax, ay, bx, by, cx, cy, dx, dy;
    lds_xy(xy, w[0] & 0x3FFu, ax, ay);
    lds_xy(xy, (w[0] >> 10) & 0x3FFu, bx, by);
    lds_xy(xy, (w[0] >> 20) & 0x3FFu, cx, cy);
    lds_xy(xy, w[1] & 0x3FFu, dx, dy);
    T r[6], s[6];
    {
      const T e1x = bx - ax, e1y = by - ay, e2x = cx - ax, e2y = cy - ay;
      p2_block_row<T, MASS>(a, e1x * e1x + e1y * e1y, e2x * e2x + e2y * e2y, e1x * e2x + e1y * e2y,
                            e1x * e2y - e1y * e2x, has_row ? 1u : 0u, r);
    }
    {
      // frame (a2, b2, d) = (b, a, d) when rev, (a, b, d) otherwise
      const T ox = rev ? bx : ax, oy = rev ? by : ay;
      const T tx = rev ? ax : bx, ty = rev ? ay : by;
      const T e1x = tx - ox, e1y = ty - oy, e2x = dx - ox, e2y = dy - oy;
      p2_block_row<T, MASS>(a, e1x * e1x + e1y * e1y, e2x * e2x + e2y * e2y, e1x * e2x + e1y * e2y,
                            e1x * e2y - e1y * e2x, (has_row && has2) ? 1u : 0u, s);
    }
    len = has_row ? (has2 ? 9 : 6) : 0;
    const int incl = wave_inclusive_scan(len);
    pre = incl - len;
    auto pos = [&](int f) { return int(((f < 8 ? w[2] >> (4 * f) : w[3]) & 15u)); };
    const bool on = has_row;
    my_stage[on ? pre + pos(0) : kSpare] = r[0] + (rev ? s[1] : s[0]);
    my_stage[on ? pre + pos(1) : kSpare] = r[1] + (rev ? s[0] : s[1]);
    my_stage[on ? pre + pos(2) : kSpare] = r[2];
    my_stage[on ? pre + pos(3) : kSpare] = r[3] + s[3];
    my_stage[on ? pre + pos(4) : kSpare] = r[4];
    my_stage[on ? pre + pos(5) : kSpare] = r[5];
    my_stage[(on && has2) ? pre + pos(6) : kSpare] = s[2];
    my_stage[(on && has2) ? pre + pos(7) : kSpare] = s[4];
    my_stage[(on && has2) ? pre + pos(8) : kSpare] = s[5];
    const int total = __builtin_amdgcn_readlane(incl, 63);
    __builtin_amdgcn_wave_barrier();
    p2_store<T, kMaxLen>(my_stage, total, rs0, r_vals);
  }
}

// ---------------------------------------------------------------------------------------
// ONE persistent launch for the vertex rows and the edge rows (TFEM_P2_PERSIST=1; the two launches
// of k_p2_rows above are the default: see launch_p2_rows for the measurement).  Workgroups
// stay resident and walk the unified tile list (vertex tiles, then edge tiles), dealt to the XCDs
// in blocks of four tiles so that every XCD gets its share of both kinds.  Software pipeline as in
// k_p1_rings (tfem_rings_kernel.hpp): iteration k issues the loads of tile k+1 (row record; the
// coordinates of up to four tile-local vertices per lane by ids that arrived an iteration earlier)
// and the vertex ids of tile k+2, evaluates the rows of tile k from xy[k & 1] in LDS, streams them
// out through the wave's stage, waits ONCE for its loads (issued a whole row phase earlier), parks
// the coordinates of tile k+1 in xy[(k+1) & 1] and passes one LDS barrier.  A wave's 64 vertex
// rows (up to 22 entries each) go through the stage in four quarters of 16 rows, its edge rows in
// two halves: 2.8 KB per wave instead of 11.3, so four workgroups fit a CU.
// ---------------------------------------------------------------------------------------
template <typename T>
struct P2AllArgs {
  const T *coords;
  const unsigned char *plan;
  T *vals;
  unsigned coords_bytes, plan_bytes, vals_bytes;
  unsigned off_desc[2], off_rows[2], off_gid[2];  // [0] vertex tiles, [1] edge tiles
  int n_tiles[2];
  int lds_vert;  // local vertices of the largest tile of either kind, even
  int nt_stores;  // developer switch (TFEM_P2_NT=1): non-temporal value stores
  T ca[2][6], cb[2][6], cd[2][6], cm[2][6];
};

constexpr int kP2StageHalf = 16 * kP2VertexRowMax + 2;  // a QUARTER of a wave's vertex rows; >= 32 * kP2EdgeRowMax + 2
constexpr int kP2LocalPerLane = 4;                      // <= 1024 local vertices per tile

struct P2Desc {
  int kind, vert_off, n_vert, row_off, n_own, row0, row1, rs0;
};

template <typename T>
__device__ __forceinline__ P2Desc p2_desc(const P2AllArgs<T> &a, int tile, int wave) {
  const int kind = tile >= a.n_tiles[0] ? 1 : 0;
  ring_const_i32 d = (ring_const_i32)(uintptr_t)(a.plan + a.off_desc[kind] + 64u * unsigned(tile - (kind ? a.n_tiles[0] : 0)));
  return P2Desc{kind, d[0], d[1], d[2], d[7], d[3 + wave], d[4 + wave], d[12 + wave]};
}

template <typename T, bool MASS>
__global__ __launch_bounds__(kP2Block, 4) void k_p2_rows_all(const P2AllArgs<T> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char p2_smem[];
  T *xy = reinterpret_cast<T *>(p2_smem);  // [2][2 * lds_vert]
  T *stage = xy + 4 * a.lds_vert;          // [waves][kP2StageHalf]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  T *my_stage = stage + wave * kP2StageHalf;
  const int n_all = a.n_tiles[0] + a.n_tiles[1];
  const int per = (n_all + 31) / 32 * 4;  // tiles per XCD, whole blocks of four
  const int xcd = blockIdx.x & 7, j0 = blockIdx.x >> 3, stride = gridDim.x >> 3;
  auto tile_at = [&](int k) {
    const int j = j0 + k * stride;
    const int t = ((j >> 2) * 8 + xcd) * 4 + (j & 3);
    return __builtin_amdgcn_readfirstlane((j < per && t < n_all) ? t : -1);
  };
  const ring_rsrc_t r_coords = ring_rsrc(a.coords, a.coords_bytes);
  const ring_rsrc_t r_plan = ring_rsrc(a.plan, a.plan_bytes);
  const ring_rsrc_t r_vals = ring_rsrc(a.vals, a.vals_bytes);
  constexpr unsigned kNone = 0x3FFFFFFu;
  const int n_local = (a.lds_vert + kP2Block - 1) / kP2Block;  // coordinate loads per lane (uniform)

  uint32_t w[8], w_ld[8];
  unsigned gid[kP2LocalPerLane], gid_ld[kP2LocalPerLane];
  T c_ld[kP2LocalPerLane][2];
#pragma unroll
  for (int i = 0; i < 8; ++i) w[i] = w_ld[i] = 0u;
#pragma unroll
  for (int j = 0; j < kP2LocalPerLane; ++j) {
    gid[j] = gid_ld[j] = 0u;
    c_ld[j][0] = c_ld[j][1] = T(0);
  }

  auto load_ids = [&](const P2Desc &d, unsigned (&g)[kP2LocalPerLane]) {
#pragma unroll
    for (int j = 0; j < kP2LocalPerLane; ++j) {
      if (j >= n_local) break;
      const int l = tid + j * kP2Block;
      g[j] = __builtin_amdgcn_raw_buffer_load_b32(
          r_plan, a.off_gid[d.kind] + (l < d.n_vert ? unsigned(d.vert_off + l) : kNone) * 4u, 0, 0);
    }
  };
  auto load_tile = [&](const P2Desc &d, const unsigned (&g)[kP2LocalPerLane]) {
    const int r = d.row0 + lane;
    const unsigned row = r < d.row1 ? unsigned(d.row_off + r) : kNone;
    if (d.kind == 0) {
      const unsigned byte = a.off_rows[0] + row * 32u;
      const ru32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r_plan, byte, 0, 0);
      const ru32x4 u = __builtin_amdgcn_raw_buffer_load_b128(r_plan, byte + 16u, 0, 0);
      w_ld[0] = v.x; w_ld[1] = v.y; w_ld[2] = v.z; w_ld[3] = v.w;
      w_ld[4] = u.x; w_ld[5] = u.y; w_ld[6] = u.z; w_ld[7] = u.w;
    } else {
      const ru32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r_plan, a.off_rows[1] + row * 16u, 0, 0);
      w_ld[0] = v.x; w_ld[1] = v.y; w_ld[2] = v.z; w_ld[3] = v.w;
    }
#pragma unroll
    for (int j = 0; j < kP2LocalPerLane; ++j) {
      if (j >= n_local) break;
      ring_load_xy<T>(r_coords, g[j], c_ld[j][0], c_ld[j][1]);
    }
  };
  auto park = [&](const P2Desc &d, T *dst) {
#pragma unroll
    for (int j = 0; j < kP2LocalPerLane; ++j) {
      if (j >= n_local) break;
      const int l = tid + j * kP2Block;
      if (l < d.n_vert) {
        dst[2 * l] = c_ld[j][0];
        dst[2 * l + 1] = c_ld[j][1];
      }
    }
  };

  int t_c = tile_at(0);
  if (t_c < 0) return;  // whole workgroup, before any barrier
  int t_n = tile_at(1), t_nn = tile_at(2);
  P2Desc dc = p2_desc(a, t_c, wave);
  P2Desc dn = p2_desc(a, t_n >= 0 ? t_n : t_c, wave);
  P2Desc dnn = p2_desc(a, t_nn >= 0 ? t_nn : t_c, wave);
  load_ids(dc, gid);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  load_tile(dc, gid);
  if (t_n >= 0) load_ids(dn, gid_ld);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  park(dc, xy);
#pragma unroll
  for (int i = 0; i < 8; ++i) w[i] = w_ld[i];
#pragma unroll
  for (int j = 0; j < kP2LocalPerLane; ++j) gid[j] = gid_ld[j];
  __syncthreads();

  int cur = 0;
  for (int k = 0;; ++k) {
    // ---- A: loads of tile k+1, vertex ids of tile k+2
    if (t_n >= 0) {
      load_tile(dn, gid);
      if (t_nn >= 0) load_ids(dnn, gid_ld);
    }
    // ---- C: the only wait -- the loads of A were issued a whole row phase ago, the stores of the
    // previous tile a whole iteration ago; D: coordinates of tile k+1 -> the other buffer.  Called
    // by the row code between its arithmetic and its stores.
    auto wait_and_park = [&]() {
      __builtin_amdgcn_s_waitcnt(0x0F70);
      if (t_n >= 0) park(dn, xy + (cur ^ 1) * 2 * a.lds_vert);
    };
    // ---- B: rows of tile k
    const T *xyc = xy + cur * 2 * a.lds_vert;
    const int my_row = dc.row0 + lane;
    const bool has_row = my_row < dc.row1;
    if (dc.kind == 0) {
      const int kk = int((w[2] >> 24) & 7u);
      auto id = [&](int i) { return (w[i / 3] >> (10 * (i % 3))) & 0x3FFu; };
      auto flag_of = [&](int i) { return (w[2] >> (10 + 2 * i)) & 3u; };
      auto field = [&](int f) { return int((w[3 + f / 6] >> (5 * (f % 6))) & 31u); };
      T xv, yv, px, py;
      lds_xy(xyc, unsigned(has_row ? my_row : 0), xv, yv);
      const uint32_t id0 = id(0);
      lds_xy(xyc, id0, px, py);
      T ecx = px - xv, ecy = py - yv;
      T qc = ecx * ecx + ecy * ecy;
      T diag = T(0), vcol[8], ecol[8], ocol[7];
#pragma unroll
      for (int i = 0; i < 8; ++i) vcol[i] = ecol[i] = T(0);
      int n_tri = 0;
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const uint32_t idn = (i + 1 < 7 && i + 1 != kk) ? id(i + 1 < 7 ? i + 1 : 0) : id0;
        lds_xy(xyc, idn, px, py);
        const T enx = px - xv, eny = py - yv;
        const T qn = enx * enx + eny * eny;
        const T p = ecx * enx + ecy * eny;
        const T cross = ecx * eny - ecy * enx;
        const uint32_t flag = flag_of(i);  // 0 for every slot i >= k
        n_tri += flag ? 1 : 0;
        T r[6];
        p2_block_row<T, MASS>(a.ca[0], a.cb[0], a.cd[0], a.cm[0], qc, qn, p, cross, flag, r);
        const bool fwd = flag != 2u;
        diag = diag + r[0];
        vcol[i] = vcol[i] + (fwd ? r[1] : r[2]);
        vcol[i + 1] = vcol[i + 1] + (fwd ? r[2] : r[1]);
        ecol[i] = ecol[i] + (fwd ? r[3] : r[5]);
        ecol[i + 1] = ecol[i + 1] + (fwd ? r[5] : r[3]);
        ocol[i] = r[4];
        ecx = enx;
        ecy = eny;
        qc = qn;
      }
      T wv = vcol[1], we = ecol[1];
#pragma unroll
      for (int j = 2; j <= 7; ++j) {
        wv = kk == j ? vcol[j] : wv;
        we = kk == j ? ecol[j] : we;
      }
      vcol[0] = vcol[0] + wv;
      ecol[0] = ecol[0] + we;
      const int len = kk > 0 ? 1 + 2 * kk + n_tri : 0;
      const int incl = wave_inclusive_scan(len);
      const int pre = incl - len;
      const bool is_long = has_row && kk == 0 && (w[3] >> 31) != 0u;
      const int true_len = is_long ? int(w[3] & 0x7FFFFFFFu) : len;
      const bool any_long = __ballot(is_long) != 0ull;
      const int csr = any_long ? dc.rs0 + wave_inclusive_scan(true_len) - true_len : 0;
      // entries in front of the rows of lanes 16, 32, 48 and behind the last one
      const int cut[5] = {0, __builtin_amdgcn_readlane(incl, 15), __builtin_amdgcn_readlane(incl, 31),
                          __builtin_amdgcn_readlane(incl, 47), __builtin_amdgcn_readlane(incl, 63)};
      constexpr int kSpare = kP2StageHalf - 2;
      wait_and_park();  // ---- C, D: in front of this tile's stores (they get a whole iteration)
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const bool mine = (lane >> 4) == pass;
        const int base = cut[pass], count = cut[pass + 1] - cut[pass];
        const int at = pre - base;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
          my_stage[(mine && i < kk) ? at + field(i) : kSpare] = vcol[i];
          my_stage[(mine && i < kk) ? at + field(7 + i) : kSpare] = ecol[i];
          my_stage[(mine && i < kk && flag_of(i)) ? at + field(14 + i) : kSpare] = ocol[i];
        }
        my_stage[(mine && kk > 0) ? at + int(w[2] >> 27) : kSpare] = diag;
        __builtin_amdgcn_wave_barrier();
        if (!any_long) {  // 16 x 22 = 352 <= 64 x 6 entries
          if (a.nt_stores)
            p2_store<T, 6, kStreamNT>(my_stage, count, dc.rs0 + base, r_vals);
          else
            p2_store<T, 6>(my_stage, count, dc.rs0 + base, r_vals);
        } else {
          p2_store_pieces<T>(my_stage, count, at, csr, is_long, r_vals, 16 * pass, 16 * pass + 16);
        }
        __builtin_amdgcn_wave_barrier();
      }
    } else {
      const bool has2 = (w[1] >> 10) & 1u;
      const bool rev = (w[1] >> 11) & 1u;
      T ax, ay, bx, by, cx, cy, dx, dy;
      lds_xy(xyc, w[0] & 0x3FFu, ax, ay);
      lds_xy(xyc, (w[0] >> 10) & 0x3FFu, bx, by);
      lds_xy(xyc, (w[0] >> 20) & 0x3FFu, cx, cy);
      lds_xy(xyc, w[1] & 0x3FFu, dx, dy);
      T r[6], s[6];
      {
        const T e1x = bx - ax, e1y = by - ay, e2x = cx - ax, e2y = cy - ay;
        p2_block_row<T, MASS>(a.ca[1], a.cb[1], a.cd[1], a.cm[1], e1x * e1x + e1y * e1y, e2x * e2x + e2y * e2y,
                              e1x * e2x + e1y * e2y, e1x * e2y - e1y * e2x, has_row ? 1u : 0u, r);
      }
      {
        const T ox = rev ? bx : ax, oy = rev ? by : ay;
        const T tx = rev ? ax : bx, ty = rev ? ay : by;
        const T e1x = tx - ox, e1y = ty - oy, e2x = dx - ox, e2y = dy - oy;
        p2_block_row<T, MASS>(a.ca[1], a.cb[1], a.cd[1], a.cm[1], e1x * e1x + e1y * e1y, e2x * e2x + e2y * e2y,
                              e1x * e2x + e1y * e2y, e1x * e2y - e1y * e2x, (has_row && has2) ? 1u : 0u, s);
      }
      const int len = has_row ? (has2 ? 9 : 6) : 0;
      const int incl = wave_inclusive_scan(len);
      const int pre = incl - len;
      constexpr int kSpare = kP2StageHalf - 2;
      const uint32_t w2 = w[2], w3 = w[3];
      auto pos = [&](int f) { return int(((f < 8 ? w2 >> (4 * f) : w3) & 15u)); };
      wait_and_park();  // ---- C, D
      const int half = __builtin_amdgcn_readlane(incl, 31), total = __builtin_amdgcn_readlane(incl, 63);
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const bool on = has_row && (lane >> 5) == pass;
        const int base = pass ? half : 0, count = pass ? total - half : half;
        const int at = pre - base;
        my_stage[on ? at + pos(0) : kSpare] = r[0] + (rev ? s[1] : s[0]);
        my_stage[on ? at + pos(1) : kSpare] = r[1] + (rev ? s[0] : s[1]);
        my_stage[on ? at + pos(2) : kSpare] = r[2];
        my_stage[on ? at + pos(3) : kSpare] = r[3] + s[3];
        my_stage[on ? at + pos(4) : kSpare] = r[4];
        my_stage[on ? at + pos(5) : kSpare] = r[5];
        my_stage[(on && has2) ? at + pos(6) : kSpare] = s[2];
        my_stage[(on && has2) ? at + pos(7) : kSpare] = s[4];
        my_stage[(on && has2) ? at + pos(8) : kSpare] = s[5];
        __builtin_amdgcn_wave_barrier();
        if (a.nt_stores)  // 32 x 9 = 288 <= 64 x 5 entries
          p2_store<T, 5, kStreamNT>(my_stage, count, dc.rs0 + base, r_vals);
        else
          p2_store<T, 5>(my_stage, count, dc.rs0 + base, r_vals);
        __builtin_amdgcn_wave_barrier();
      }
    }
    if (t_n < 0) break;
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = w_ld[i];
#pragma unroll
    for (int j = 0; j < kP2LocalPerLane; ++j) gid[j] = gid_ld[j];
    // ---- E: xy[(k+1) & 1] is complete, nobody reads xy[k & 1] any more
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    t_c = t_n;
    dc = dn;
    t_n = t_nn;
    dn = dnn;
    t_nn = tile_at(k + 3);
    if (t_nn >= 0) dnn = p2_desc(a, t_nn, wave);
    cur ^= 1;
  }
}

// Vertex rows with 8 .. 15 neighbours: SIXTEEN lanes per row, lane i = slot i of the fan (the
// rows are scattered over the mesh -- ~13 % of the vertices of a Delaunay mesh -- so everything
// goes by global ids).  A lane evaluates its slot's triangle, takes what the previous slot's
// triangle adds to its own column from the lane before it (the first slot from slot k - 1: the
// fan closes there, or that slot has no triangle), and writes its three entries; the diagonal is
// the sum over the sixteen lanes.  No loop, no serial chain of dependent loads.
template <typename T, bool MASS>
__global__ __launch_bounds__(kP2Block) void k_p2_long_rows(const P2RowArgs<T> a, unsigned off_long, int n_long) {
  const int gtid = int(blockIdx.x) * kP2Block + int(threadIdx.x);
  const int row = gtid >> 4, i = gtid & 15;
  const bool live = row < n_long;
  const uint32_t *rec = reinterpret_cast<const uint32_t *>(a.plan + off_long) + 32 * size_t(live ? row : 0);
  const uint32_t v = rec[0];
  const int k = int(rec[2] & 0xFFu);
  const int dpos = int(rec[2] >> 8);
  const bool slot = live && i < k;
  const uint32_t flag = slot ? (rec[3] >> (2 * i)) & 3u : 0u;
  const int nxt = i + 1 == k ? 0 : i + 1;
  const uint32_t g0 = rec[4 + (slot ? i : 0)], g1 = rec[4 + (slot ? nxt : 0)];
  const T xv = a.coords[2 * size_t(v)], yv = a.coords[2 * size_t(v) + 1];
  const T ecx = a.coords[2 * size_t(g0)] - xv, ecy = a.coords[2 * size_t(g0) + 1] - yv;
  const T enx = a.coords[2 * size_t(g1)] - xv, eny = a.coords[2 * size_t(g1) + 1] - yv;
  T r[6];
  p2_block_row<T, MASS>(a, ecx * ecx + ecy * ecy, enx * enx + eny * eny, ecx * enx + ecy * eny,
                        ecx * eny - ecy * enx, flag, r);
  // r: v, p1, p2, edge (v,p1), edge (p1,p2), edge (p2,v); (p1, p2) = (n_i, n_next) for flag 1 and
  // (n_next, n_i) for flag 2
  const bool fwd = flag != 2u;
  const T own_v = fwd ? r[1] : r[2], own_e = fwd ? r[3] : r[5];    // to this slot's columns
  const T next_v = fwd ? r[2] : r[1], next_e = fwd ? r[5] : r[3];  // to the next slot's columns
  const int lane = int(threadIdx.x) & 63;
  const int from = (lane & ~15) + (i == 0 ? (k > 0 ? k - 1 : 0) : i - 1);
  const T vcol = own_v + __shfl(next_v, from, 64);
  const T ecol = own_e + __shfl(next_e, from, 64);
  T diag = r[0];
  diag = diag + __shfl_xor(diag, 8, 64);
  diag = diag + __shfl_xor(diag, 4, 64);
  diag = diag + __shfl_xor(diag, 2, 64);
  diag = diag + __shfl_xor(diag, 1, 64);
  if (!slot) return;
  auto field = [&](int f) { return int((rec[19 + f / 5] >> (6 * (f % 5))) & 63u); };
  T *out = a.vals + rec[1];
  out[field(i)] = vcol;
  out[field(15 + i)] = ecol;
  if (flag) out[field(30 + i)] = r[4];
  if (i == 0) out[dpos] = diag;
}

template <typename T>
static int launch_p2_rows(const void *coords, int quad_order, double alpha, double beta,
                          const unsigned char *plan, const int64_t *z, int64_t nnz, void *vals,
                          hipStream_t stream) {
  TriTables tables;
  if (!build_tri_tables(quad_order, int(sizeof(T)), &tables))
    return fail(TFEM_ERR_UNSUPPORTED, "Integration order not implemented");
  if (z[0] + z[1] == 0) return TFEM_OK;
  if (!coords || !plan || !vals) return fail(TFEM_ERR_INVALID_ARGUMENT, "NULL pointer");
  if (z[4] > 1024 || z[5] > 1024 || z[6] > kP2Block)
    return fail(TFEM_ERR_INVALID_ARGUMENT, "P2 row plan exceeds the kernel's capacities");
  const int64_t rb = int64_t(sizeof(T));
  const int64_t extents[3] = {z[2] * 2 * rb, z[16], nnz * rb};
  for (int64_t e : extents)
    if (e < 0 || e >= (int64_t(1) << 32))
      return fail(TFEM_ERR_INDEX_RANGE, "an array of %lld bytes does not fit the 32-bit offsets "
                  "of the P2 row kernel", (long long)e);
  const bool mass = beta != 0.0;
  // row 0 (vertex DoF at p0) / row 3 (edge DoF (p0, p1)) of the constant maps, in T, sums in
  // quadrature order like the reference's (integrand * dx).sum(-3)
  auto fill_tables = [&](int kind, T (&oa)[6], T (&ob)[6], T (&od)[6], T (&om)[6]) {
    const int row = kind == 0 ? 0 : 3;
    for (int m = 0; m < 6; ++m) {
      T ca = T(0), cb = T(0), cd = T(0), cm = T(0);
      for (int q = 0; q < tables.nq; ++q) {
        const T hw = T(tables.hw[q]);
        const T r0 = T(tables.rgrad2[q][row][0]), r1 = T(tables.rgrad2[q][row][1]);
        const T m0 = T(tables.rgrad2[q][m][0]), m1 = T(tables.rgrad2[q][m][1]);
        ca = ca + hw * (r0 * m0);
        cb = cb + hw * (r0 * m1 + r1 * m0);
        cd = cd + hw * (r1 * m1);
        cm = cm + hw * (T(tables.phi2[q][row]) * T(tables.phi2[q][m]));
      }
      oa[m] = T(alpha) * ca;
      ob[m] = T(alpha) * cb;
      od[m] = T(alpha) * cd;
      om[m] = T(beta) * cm;
    }
  };
  // TFEM_P2_PERSIST=1: ONE persistent, software-pipelined launch for both kinds of row
  // (k_p2_rows_all).  Measured at S(707) = 999,698 elements (profiles/r03_p2_persistent.log): 55.5 us
  // against 50.7 us for the two launches below -- both move their 273 MB at 5.0-5.4 TB/s, the rate of
  // a mixed read / write stream on this part; what separates them from the 216 MB the roofline
  // counts are the row records and the coordinate gathers of the edge tiles, not the launch
  // structure.  The two launches stay the default.
  bool persist = false;
  if (const char *v = std::getenv("TFEM_P2_PERSIST")) persist = std::atoi(v) != 0;
  if (persist) {
    P2AllArgs<T> a;
    std::memset(&a, 0, sizeof(a));
    a.coords = static_cast<const T *>(coords);
    a.plan = plan;
    a.vals = static_cast<T *>(vals);
    a.coords_bytes = unsigned(extents[0]);
    a.plan_bytes = unsigned(extents[1]);
    a.vals_bytes = unsigned(extents[2]);
    for (int kind = 0; kind < 2; ++kind) {
      a.off_desc[kind] = unsigned(z[10 + 3 * kind]);
      a.off_rows[kind] = unsigned(z[11 + 3 * kind]);
      a.off_gid[kind] = unsigned(z[12 + 3 * kind]);
      a.n_tiles[kind] = int(z[kind]);
      fill_tables(kind, a.ca[kind], a.cb[kind], a.cd[kind], a.cm[kind]);
    }
    a.lds_vert = (int(std::max(z[4], z[5])) + 1) & ~1;
    if (const char *v = std::getenv("TFEM_P2_NT")) a.nt_stores = std::atoi(v);
    if (a.lds_vert > kP2LocalPerLane * kP2Block)
      return fail(TFEM_ERR_INVALID_ARGUMENT, "P2 row plan exceeds the kernel's capacities");
    const size_t lds = size_t(4 * a.lds_vert) * sizeof(T) + size_t(kP2Waves) * size_t(kP2StageHalf) * sizeof(T);
    void *kernel = mass ? reinterpret_cast<void *>(k_p2_rows_all<T, true>) : reinterpret_cast<void *>(k_p2_rows_all<T, false>);
    static std::mutex occ_mutex;
    static struct { void *kernel; size_t lds; int per_cu; } occ[8];
    static int occ_used = 0;
    int per_cu = 0;
    {
      std::lock_guard<std::mutex> guard(occ_mutex);
      for (int i = 0; i < occ_used; ++i)
        if (occ[i].kernel == kernel && occ[i].lds == lds) per_cu = occ[i].per_cu;
      if (per_cu == 0) {
        if (lds > 64 * 1024) {
          hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
          if (e != hipSuccess) return fail(TFEM_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        }
        hipError_t oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kP2Block, lds);
        if (oe != hipSuccess || per_cu < 1) per_cu = 1;
        if (occ_used < 8) {
          occ[occ_used].kernel = kernel;
          occ[occ_used].lds = lds;
          occ[occ_used++].per_cu = per_cu;
        }
      }
    }
    if (const char *v = std::getenv("TFEM_P2_PER_CU")) per_cu = std::max(1, std::min(per_cu, std::atoi(v)));
    int cus = 256, dev = 0;
    hipDeviceProp_t prop;
    static int cu_count = 0;
    if (cu_count == 0 && hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      cu_count = prop.multiProcessorCount;
    if (cu_count > 0) cus = cu_count;
    const int64_t n_all = z[0] + z[1];
    const int per = int((n_all + 31) / 32) * 4;
    const int blocks = std::min(per * 8, (cus * per_cu / 8) * 8);
    const dim3 grid{unsigned(blocks)}, block{unsigned(kP2Block)};
    void *params[] = {&a};
    hipError_t e = hipLaunchKernel(kernel, grid, block, params, lds, stream);
    if (e != hipSuccess) return fail(TFEM_ERR_HIP, "P2 row kernel launch: %s", hipGetErrorString(e));
    if (z[18] > 0) {  // the vertex rows with 8 .. 15 neighbours
      P2RowArgs<T> la;
      std::memset(&la, 0, sizeof(la));
      la.coords = a.coords;
      la.plan = plan;
      la.vals = a.vals;
      for (int m = 0; m < 6; ++m) {
        la.ca[m] = a.ca[0][m];
        la.cb[m] = a.cb[0][m];
        la.cd[m] = a.cd[0][m];
        la.cm[m] = a.cm[0][m];
      }
      const dim3 lgrid{unsigned((16 * z[18] + kP2Block - 1) / kP2Block)};
      if (mass)
        hipLaunchKernelGGL((k_p2_long_rows<T, true>), lgrid, block, 0, stream, la, unsigned(z[17]), int(z[18]));
      else
        hipLaunchKernelGGL((k_p2_long_rows<T, false>), lgrid, block, 0, stream, la, unsigned(z[17]), int(z[18]));
      e = hipGetLastError();
      if (e != hipSuccess) return fail(TFEM_ERR_HIP, "P2 long-row kernel launch: %s", hipGetErrorString(e));
    }
    return TFEM_OK;
  }
  for (int kind = 0; kind < 2; ++kind) {
    if (z[kind] == 0) continue;
    P2RowArgs<T> a;
    std::memset(&a, 0, sizeof(a));
    a.coords = static_cast<const T *>(coords);
    a.plan = plan;
    a.vals = static_cast<T *>(vals);
    a.coords_bytes = unsigned(extents[0]);
    a.plan_bytes = unsigned(extents[1]);
    a.vals_bytes = unsigned(extents[2]);
    a.off_desc = unsigned(z[10 + 3 * kind]);
    a.off_rows = unsigned(z[11 + 3 * kind]);
    a.off_gid = unsigned(z[12 + 3 * kind]);
    a.n_tiles = int(z[kind]);
    a.xcd_ranges = 1;
    if (const char *v = std::getenv("TFEM_P2_XCD")) a.xcd_ranges = std::strcmp(v, "interleave") != 0;
    a.lds_vert = (int(z[4 + kind]) + 1) & ~1;
    fill_tables(kind, a.ca, a.cb, a.cd, a.cm);
    const int max_len = kind == 0 ? kP2VertexRowMax : kP2EdgeRowMax;
    // vertex rows: the wave's stage in parts (developer switch TFEM_P2_PASSES=2|4) -- more workgroups
    // per CU (3 -> 5 -> 6) and SLOWER: S(707) 44.7-45.2 us in one part, 46.0-46.2 in two, 48.7-49.4 in
    // four (profiles/r03_p2_stage_passes.log): the launch is not short of waves, it is at the rate of
    // the bytes it moves
    int passes = 1;
    if (const char *v = std::getenv("TFEM_P2_PASSES")) {
      const int p = std::atoi(v);
      if (kind == 0 && (p == 1 || p == 2 || p == 4)) passes = p;
    }
    const size_t lds = size_t(2 * a.lds_vert) * sizeof(T) + size_t(kP2Waves) * size_t((64 / passes) * max_len + 2) * sizeof(T);
    void *kernel;
    if (kind == 1)
      kernel = mass ? reinterpret_cast<void *>(k_p2_rows<T, 1, true>) : reinterpret_cast<void *>(k_p2_rows<T, 1, false>);
    else if (passes == 4)
      kernel = mass ? reinterpret_cast<void *>(k_p2_rows<T, 0, true, 4>) : reinterpret_cast<void *>(k_p2_rows<T, 0, false, 4>);
    else if (passes == 2)
      kernel = mass ? reinterpret_cast<void *>(k_p2_rows<T, 0, true, 2>) : reinterpret_cast<void *>(k_p2_rows<T, 0, false, 2>);
    else
      kernel = mass ? reinterpret_cast<void *>(k_p2_rows<T, 0, true>) : reinterpret_cast<void *>(k_p2_rows<T, 0, false>);
    if (lds > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
      if (e != hipSuccess) return fail(TFEM_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    const int per = int((z[kind] + 7) / 8);
    const dim3 grid{unsigned(per * 8)}, block{unsigned(kP2Block)};
    void *params[] = {&a};
    hipError_t e = hipLaunchKernel(kernel, grid, block, params, lds, stream);
    if (e != hipSuccess) return fail(TFEM_ERR_HIP, "P2 row kernel launch: %s", hipGetErrorString(e));
    if (kind == 0 && z[18] > 0) {  // the vertex rows with 8 .. 15 neighbours
      const dim3 lgrid{unsigned((16 * z[18] + kP2Block - 1) / kP2Block)};  // sixteen lanes per row
      if (mass)
        hipLaunchKernelGGL((k_p2_long_rows<T, true>), lgrid, block, 0, stream, a, unsigned(z[17]), int(z[18]));
      else
        hipLaunchKernelGGL((k_p2_long_rows<T, false>), lgrid, block, 0, stream, a, unsigned(z[17]), int(z[18]));
      e = hipGetLastError();
      if (e != hipSuccess) return fail(TFEM_ERR_HIP, "P2 long-row kernel launch: %s", hipGetErrorString(e));
    }
  }
  return TFEM_OK;
}

}  // namespace tfem

extern "C" {

int tfem_p2_assemble_rows(const void *coords, int real_bytes, int quad_order, double alpha,
                          double beta, const void *plan_device, const int64_t *plan_layout_host,
                          void *vals, int64_t nnz, void *stream) {
  using namespace tfem;
  if (real_bytes != 4 && real_bytes != 8)
    return fail(TFEM_ERR_INVALID_ARGUMENT, "real_bytes must be 4 or 8");
  if (!plan_layout_host) return fail(TFEM_ERR_INVALID_ARGUMENT, "plan_layout_host is NULL");
  const auto *plan = static_cast<const unsigned char *>(plan_device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  return real_bytes == 8
             ? launch_p2_rows<double>(coords, quad_order, alpha, beta, plan, plan_layout_host, nnz, vals, s)
             : launch_p2_rows<float>(coords, quad_order, alpha, beta, plan, plan_layout_host, nnz, vals, s);
}

}  // extern "C"
