// The P1 row-form kernel k_p1_rings (template) and its launch structure: shared by
// tfem_rings.hip (matrix-only launches and launches that read pre-evaluated source values) and
// tfem_rings_src.hip (launches that evaluate a source program; built with other flags).
// Description of the kernel: tfem_rings.hip.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "tfem_common.hpp"
#include "tfem_rowkit.hpp"
#include "tfem_source.hpp"

// Reassociation is harmless here (see above: the row form is not the reference's operation
// order anyway; parity is asserted at 1e-12 against the oracle).
#pragma clang fp contract(fast)

// developer switch: 0 = the general sum over the rule's points in the source-program launches' g
#ifndef TFEM_SRC_QSYM
#define TFEM_SRC_QSYM 1
#endif

namespace tfem {

constexpr int kRingBlock = 256;             // lanes per workgroup = owned rows per tile
constexpr int kRingWaves = kRingBlock / 64;
constexpr int kRingVertCap = 1024;          // 10-bit local ids

template <typename T>
struct RingArgs {
  const T *coords;
  const unsigned char *plan;
  T *vals;
  const T *fq;  // (n_elems, Q) source values, load vector only
  T *fout;
  unsigned coords_bytes, plan_bytes, vals_bytes, fq_bytes, fout_bytes;
  unsigned off_desc, off_rows, off_rowstart, off_gid, off_elems, off_telems;
  int lds_elem;   // element slots reserved in LDS per buffer (load vector)
  int xcd_interleave;  // 0: every XCD walks its own contiguous eighth of the tile list
                       // G > 0: the tile list is dealt to the XCDs in blocks of G tiles (one
                       //        front of tiles across the chip)
  int n_tiles;    // tiles of this launch (off_desc points at the first one's descriptor)
  int lds_vert;   // vertex slots reserved in LDS
  T stiff_w;      // alpha * sum_q w_q / 2
  T mass_d, mass_o;  // beta * sum_q (w_q/2) l_i l_i, beta * sum_q (w_q/2) l_i l_j (i != j)
  T lamw[3][kMaxQuad];  // l_i(q) * w_q / 2 by local vertex i (load vector)
  T lam[3][kMaxQuad];   // l_i(q): the integration points of an element (source programs)
  T hw[kMaxQuad];       // w_q / 2 (source programs: g_i = sum_q (f_q w_q / 2) l_i(q) needs this and `lam`
                        // only -- a second table of 3 Q doubles does not fit the scalar registers and
                        // came back from its spill lanes with two v_readlane per use)
  // source programs, the 4-point rule (order 3: the centroid, weights (c0, c0, c0), and the three
  // points with weights (b, b, b) + d on vertex 1, 2, 0; element_tri.py:99-107): [0..2] c0, b, d;
  // [3..5] c0 w_0 / 2, b w_1 / 2, d w_1 / 2 of g_i = sum_q (f_q w_q / 2) l_i(q) (w_1 = w_2 = w_3):
  // g_i = c0 w_0/2 f_0 + b w_1/2 (f_1 + f_2 + f_3) + d w_1/2 f_{q(i)}
  T qsym[6];
  unsigned off_tverts;  // packed local vertex triples of the tiles' elements (source programs)
  // source programs: the launch walks positions [u_first, u_first + n_tiles) of the plan's chain
  // order (off_chain: position -> tile), a workgroup takes chain_len consecutive positions;
  // off_hin: per owned row the local id its vertex has in the previous tile of the block (uint16)
  unsigned off_chain, off_hin;
  int chain_len, u_first;
  // n_runs > 0 (launches over the whole plan): the chain order in n_runs runs (off_runs: first
  // position of every run, + the number of tiles), one per resident workgroup, a run = one block
  int n_runs;
  unsigned off_runs;
  int flags;      // 1024: plain instead of non-temporal value stores (every build: the launch's store
                  // policy).  Ablation build only (TFEM_RINGS_DEBUG): 1 no value stores, 2 no row arithmetic,
                  // 4 no coordinate gather, 8 no staging and stores, 16 no record loads, 32 no
                  // source-value loads, 64 no element-id loads, 128 no g staging, 256 stamps
  unsigned long long *stamps;  // ablation build, flag 256: 8 cycle sums per wave
  SrcProgram<T> src;  // SRC instantiations: f(x, y), evaluated at the integration points in the launch
};

// Ablation build only: shader-clock stamp (cdna_hip_programming.md section 7).
__device__ __forceinline__ unsigned long long ring_stamp() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

#ifndef TFEM_RING_BAND
#define TFEM_RING_BAND 11
#endif
// Developer ablation of the source-program launches (tools/ablate_src.py; results are wrong by
// design): 1 two instead of three elements per lane, 2 no barrier behind phase G, 4 no LDS adds,
// 8 sin / cos cost nothing (tfem_source.hpp), 16 no phase G at all
#ifndef TFEM_SRC_ABL
#define TFEM_SRC_ABL 0
#endif
constexpr int kRingBand = TFEM_RING_BAND;  // short slot loop of the 15-slot kernels (0: none)

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

// The row of local vertex `lv`: entries of the neighbour slots in off[0 .. k), the diagonal in
// diag (off[k ..] is scratch).  With q_i = |e_i|^2 and p = e_i.e_next the three entries of a
// triangle are c (p - q_next), c (p - q_i) and their negative sum, c = +-W / (e_i x e_next); the
// mass part adds det * M.  The triangle of slot i lands in off[i] and off[i + 1]; for the slot
// that closes the fan (i + 1 == k) the second one belongs to slot 0 and is moved there at the
// end (slots i >= k carry flag 0 and contribute nothing).  No branches: the reciprocal chains
// of the slots interleave.
// NIT < SLOTS: the caller knows that no row of the wave has more than NIT neighbours (the
// 15-slot records of unstructured meshes: 11 slots cover ~94 % of the waves of a Delaunay mesh).
template <typename T, int SLOTS, bool MASS, bool DETS, int NIT = SLOTS>
__device__ __forceinline__ void ring_row(const RingArgs<T> &a, const RingRec<SLOTS> &rec,
                                         uint32_t lv, const T *xy, T (&off)[SLOTS + 1], T &diag,
                                         T (&sdets)[SLOTS]) {
  const int k = rec.k();
  T xv, yv, px, py;
  lds_xy(xy, lv, xv, yv);
  const uint32_t id0 = rec.id(0);
  lds_xy(xy, id0, px, py);
  T ecx = px - xv, ecy = py - yv;
  T qc = ecx * ecx + ecy * ecy;
  T dsum = T(0);  // sum of the signed determinants around the vertex (mass part)
#pragma unroll
  for (int i = 0; i <= SLOTS; ++i) off[i] = T(0);
  if (DETS) {
#pragma unroll
    for (int i = NIT; i < SLOTS; ++i) sdets[i] = T(0);
  }
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    // neighbour behind slot i: slot i + 1, or slot 0 where the fan closes
    const uint32_t idn = (i + 1 < SLOTS && i + 1 != k) ? rec.id(i + 1 < SLOTS ? i + 1 : 0) : id0;
    lds_xy(xy, idn, px, py);
    const T enx = px - xv, eny = py - yv;
    const T qn = enx * enx + eny * eny;
    const T p = ecx * enx + ecy * eny;
    const T cross = ecx * eny - ecy * enx;  // +- the signed determinant (element_tri.py:139)
    const uint32_t flag = rec.flag(i);      // 0 for every slot i >= k
    const T cs = flag_weight<T>(a.stiff_w, flag) * fast_rcp<T>(flag ? cross : T(1));
    off[i] = off[i] + cs * (p - qn);
    off[i + 1] = off[i + 1] + cs * (p - qc);
    if (MASS || DETS) {
      const T sdet = flag_weight<T>(T(1), flag) * cross;  // signed determinant, 0 without triangle
      if (DETS) sdets[i] = sdet;
      if (MASS) {
        const T m = a.mass_o * sdet;
        off[i] = off[i] + m;
        off[i + 1] = off[i + 1] + m;
        dsum = dsum + sdet;
      }
    }
    ecx = enx;
    ecy = eny;
    qc = qn;
  }
  // the closing triangle's second entry sits in off[k]: it belongs to slot 0
  T wrapv = off[1];
#pragma unroll
  for (int j = 2; j <= SLOTS; ++j) wrapv = k == j ? off[j] : wrapv;
  // stiffness rows sum to zero (constants are in the kernel of the gradient): the diagonal is
  // minus the sum of the off-diagonal stiffness entries; the mass part is added on top
  T sum = off[0];
#pragma unroll
  for (int j = 1; j <= SLOTS; ++j) sum = sum + off[j];  // = sum_{j<k} off[j] + wrapv (once)
  off[0] = off[0] + wrapv;
  if (MASS) {
    // the sum above holds stiffness AND off-diagonal mass (M_ij det, twice per triangle): take
    // the mass out again before negating, then add the diagonal mass
    diag = a.mass_d * dsum - (sum - T(2) * a.mass_o * dsum);
  } else {
    diag = -sum;
  }
}

// ring_row for a REGULAR row: six neighbours, six triangles, all of them stored counter-clockwise
// around the vertex (record: k = 6, flags 1 1 1 1 1 1 0) -- the interior vertices of a structured
// mesh.  No flag arithmetic, no selection of the closing neighbour; the same operations in the
// same order as ring_row otherwise (bitwise the same entries).  7-slot records only.
template <typename T, bool MASS>
__device__ __forceinline__ void ring_row_regular(const RingArgs<T> &a, const RingRec<7> &rec, uint32_t lv,
                                                 const T *xy, T (&off)[8], T &diag) {
  T xv, yv, px, py;
  lds_xy(xy, lv, xv, yv);
  lds_xy(xy, rec.id(0), px, py);
  const T e0x = px - xv, e0y = py - yv;
  const T q0 = e0x * e0x + e0y * e0y;
  T ecx = e0x, ecy = e0y, qc = q0;
  T dsum = T(0), wrapv = T(0);
#pragma unroll
  for (int i = 0; i < 8; ++i) off[i] = T(0);
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    T enx = e0x, eny = e0y, qn = q0;
    if (i < 5) {
      lds_xy(xy, rec.id(i + 1), px, py);
      enx = px - xv;
      eny = py - yv;
      qn = enx * enx + eny * eny;
    }
    const T p = ecx * enx + ecy * eny;
    const T cross = ecx * eny - ecy * enx;
    const T cs = a.stiff_w * fast_rcp<T>(cross);
    T &behind = i < 5 ? off[i + 1] : wrapv;  // the neighbour behind the triangle (ring_row: off[6] for the last)
    off[i] = off[i] + cs * (p - qn);
    behind = behind + cs * (p - qc);
    if (MASS) {
      const T m = a.mass_o * cross;
      off[i] = off[i] + m;
      behind = behind + m;
      dsum = dsum + cross;
    }
    ecx = enx;
    ecy = eny;
    qc = qn;
  }
  T sum = off[0];
#pragma unroll
  for (int j = 1; j < 6; ++j) sum = sum + off[j];
  sum = (sum + wrapv) + T(0);  // off[6] = wrapv, off[7] = 0 in ring_row's sum
  off[0] = off[0] + wrapv;
  if (MASS)
    diag = a.mass_d * dsum - (sum - T(2) * a.mass_o * dsum);
  else
    diag = -sum;
}

// Per-wave LDS stage: the wave's CSR entries, compact and in CSR order (row r of the wave
// starts at the exclusive prefix sum of the row lengths).  Two spare entries behind the
// 64 * (SLOTS + 1) real ones absorb the slots a row does not have, so staging has no branches.
template <typename T, int SLOTS>
constexpr int ring_stage_entries() { return 64 * (SLOTS + 1) + 2; }

// Row entries -> the wave's stage (plain LDS stores).  `pre` = stage index of this lane's row;
// returns the number of entries the wave staged (uniform).
template <typename T, int SLOTS, int NIT = SLOTS>
__device__ __forceinline__ int ring_stage(const RingRec<SLOTS> &rec, const T (&off)[SLOTS + 1], T diag,
                                          T *stage, int &pre) {
  const int k = rec.k();
  const int len = k > 0 ? k + 1 : 0;
  const int incl = wave_inclusive_scan(len);
  pre = incl - len;
  constexpr int kSpare = 64 * (SLOTS + 1);
#pragma unroll
  for (int i = 0; i < NIT; ++i) stage[i < k ? pre + rec.pos(i) : kSpare] = off[i];
  stage[k > 0 ? pre + rec.dpos() : kSpare] = diag;
  return __builtin_amdgcn_readlane(incl, 63);
}

// ring_stage for a wave of regular rows (ring_row_regular): every row has seven entries, so row r
// of the wave starts at 7 r -- no scan, no tests (lanes without a row sit behind the last row and
// write behind `total`, which nothing reads).
template <typename T>
__device__ __forceinline__ int ring_stage_regular(const RingRec<7> &rec, const T (&off)[8], T diag, T *stage,
                                                  int &pre, int n_rows) {
  pre = 7 * int(threadIdx.x & 63);
#pragma unroll
  for (int i = 0; i < 6; ++i) stage[pre + rec.pos(i)] = off[i];
  stage[pre + rec.dpos()] = diag;
  return 7 * n_rows;
}

// The wave's stage -> global memory when the wave's rows form ONE run (a group of rows that is
// contiguous in the CSR value array): stage index + delta = CSR index for the whole wave.
// Lane j of step u takes entries 128 u + 2 j and the next one: 16-byte stores, 1 KiB
// contiguous per wave instruction; whole steps need no per-lane test.  Every LDS read is
// issued first (one LDS latency per tile).  Same wave as ring_stage: LDS executes a wave's
// operations in order.
template <typename T, int SLOTS, bool DBG = false>
__device__ __forceinline__ void ring_store_run1(const T *stage, int total, int delta, ring_rsrc_t r_vals,
                                                int flags = 0) {
  const int lane = threadIdx.x & 63;
  constexpr int kSteps = 64 * (SLOTS + 1) / 128;
  T va[kSteps][2];
#pragma unroll
  for (int u = 0; u < kSteps; ++u) {
    // 15-slot stage: rows hold ~7 of 16 entries, the steps behind `total` are skipped (uniform)
    if (SLOTS > 7 && u >= 2 && 128 * u >= total) {
      va[u][0] = va[u][1] = T(0);
      continue;
    }
    va[u][0] = stage[128 * u + 2 * lane];
    va[u][1] = stage[128 * u + 2 * lane + 1];
  }
#pragma unroll
  for (int u = 0; u < kSteps; ++u) {
    if (SLOTS > 7 && u >= 2 && 128 * u >= total) continue;
    const int s0 = 128 * u + 2 * lane;
    const unsigned byte = unsigned(s0 + delta) * unsigned(sizeof(T));
    const T v0 = va[u][0], v1 = va[u][1];
    if (DBG && (flags & 1)) {
      if (v0 == T(-1.2345e30) && v1 == v0) __builtin_amdgcn_raw_buffer_store_b32(0u, r_vals, byte, 0, 0);
    } else if (128 * (u + 1) <= total || s0 + 1 < total) {  // first test is wave-uniform
      if constexpr (sizeof(T) == 8) {
        const ru32x2 x = __builtin_bit_cast(ru32x2, v0), y = __builtin_bit_cast(ru32x2, v1);
        if (flags & 1024)  // plain (temporal) stores: the launch's store policy (RingArgs::flags)
          __builtin_amdgcn_raw_buffer_store_b128(ru32x4{x.x, x.y, y.x, y.y}, r_vals, byte, 0, 0);
        else
          __builtin_amdgcn_raw_buffer_store_b128(ru32x4{x.x, x.y, y.x, y.y}, r_vals, byte, 0, kStreamNT);
      } else {
        __builtin_amdgcn_raw_buffer_store_b64(
            ru32x2{__builtin_bit_cast(unsigned, v0), __builtin_bit_cast(unsigned, v1)}, r_vals, byte, 0, kStreamNT);
      }
    } else if (s0 < total) {
      if constexpr (sizeof(T) == 8)
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(ru32x2, v0), r_vals, byte, 0, kStreamNT);
      else
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v0), r_vals, byte, 0, kStreamNT);
    }
  }
}

// One run per wave EXCEPT for long rows (plans of unstructured meshes: vertices with 8 .. 15
// neighbours are written by k_p1_long_rows): their entries are not in the stage, the rows behind
// them start further on in the CSR array.  `pre` = stage index of the lane's row, `csr` = CSR
// offset of the lane's row; the stage is streamed out piece by piece between the long rows.
template <typename T>
__device__ __forceinline__ void ring_store_pieces(const T *stage, int total, int pre, int csr, bool is_long,
                                                  ring_rsrc_t r_vals) {
  const int lane = threadIdx.x & 63;
  unsigned long long holes = __ballot(is_long);
  int first_lane = 0;
  for (;;) {
    const int stop_lane = holes ? __builtin_ctzll(holes) : 64;
    if (first_lane < stop_lane) {
      const int b = __builtin_amdgcn_readlane(pre, first_lane);
      const int e = stop_lane < 64 ? __builtin_amdgcn_readlane(pre, stop_lane) : total;
      const int delta = __builtin_amdgcn_readlane(csr, first_lane) - b;
      for (int s0 = b + lane; s0 < e; s0 += 64) {
        const unsigned byte = unsigned(s0 + delta) * unsigned(sizeof(T));
        if constexpr (sizeof(T) == 8)
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(ru32x2, stage[s0]), r_vals, byte, 0, kStreamNT);
        else
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, stage[s0]), r_vals, byte, 0, kStreamNT);
      }
    }
    if (!holes) break;
    holes &= holes - 1;
    first_lane = stop_lane + 1;
  }
}

// General form: the rows of a wave form several runs (one per grid line of a Z-order tile;
// one per row for a numbering without locality).  Per run, as above.
template <typename T, int SLOTS, bool DBG = false>
__device__ __forceinline__ void ring_store(const T *stage, int total, int pre, int rowstart, int len,
                                           ring_rsrc_t r_vals, int flags = 0) {
  const int lane = threadIdx.x & 63;
  __builtin_amdgcn_wave_barrier();
  // A row starts a run when the rows since the previous non-empty row are not one contiguous
  // piece of the CSR array.  `breaks` = lanes whose row does not begin where the row of lane
  // r - 1 ends (wave_shr:1; rows without entries -- isolated vertices -- take part with their
  // offset: an empty row between two pieces must not glue them together).
  const int prev_end = __builtin_amdgcn_update_dpp(-1, rowstart + len, 0x138, 0xF, 0xF, false);
  const unsigned long long has_row = __ballot(len > 0);
  const unsigned long long breaks = __ballot(prev_end != rowstart);
  const unsigned long long below = (1ull << lane) - 1ull;           // lanes < this one
  const unsigned long long ne_below = has_row & below;
  // lanes in (previous non-empty lane, this lane]
  const unsigned long long since = ne_below ? ~((2ull << (63 - __builtin_clzll(ne_below))) - 1ull) : ~0ull;
  const bool start = len > 0 && ((breaks & since & (below | (1ull << lane))) != 0ull || ne_below == 0ull);
  unsigned long long starts = __ballot(start);
  if ((starts & (starts - 1)) == 0) {
    const int delta = starts ? __builtin_amdgcn_readlane(rowstart, __builtin_ctzll(starts)) : 0;
    ring_store_run1<T, SLOTS, DBG>(stage, total, delta, r_vals, flags);
    __builtin_amdgcn_wave_barrier();
    return;
  }
  while (starts) {
    const int r = __builtin_ctzll(starts);
    starts &= starts - 1;
    const int b = __builtin_amdgcn_readlane(pre, r);
    const int delta = __builtin_amdgcn_readlane(rowstart, r) - b;
    const int e = starts ? __builtin_amdgcn_readlane(pre, __builtin_ctzll(starts)) : total;
    for (int s0 = b + 2 * lane; s0 - 2 * lane < e; s0 += 128) {
      const T v0 = stage[s0], v1 = stage[s0 + 1];
      const unsigned byte = unsigned(s0 + delta) * unsigned(sizeof(T));
      if (DBG && (flags & 1)) {
        if (v0 == T(-1.2345e30) && v1 == v0) __builtin_amdgcn_raw_buffer_store_b32(0u, r_vals, byte, 0, 0);
      } else if (s0 + 1 < e) {
        if constexpr (sizeof(T) == 8) {
          const ru32x2 x = __builtin_bit_cast(ru32x2, v0), y = __builtin_bit_cast(ru32x2, v1);
          __builtin_amdgcn_raw_buffer_store_b128(ru32x4{x.x, x.y, y.x, y.y}, r_vals, byte, 0, kStreamNT);
        } else {
          __builtin_amdgcn_raw_buffer_store_b64(
              ru32x2{__builtin_bit_cast(unsigned, v0), __builtin_bit_cast(unsigned, v1)}, r_vals, byte, 0, kStreamNT);
        }
      } else if (s0 < e) {
        if constexpr (sizeof(T) == 8)
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(ru32x2, v0), r_vals, byte, 0, kStreamNT);
        else
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v0), r_vals, byte, 0, kStreamNT);
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
}

template <int SLOTS>
__device__ __forceinline__ void ring_load_rec(ring_rsrc_t r, unsigned byte, RingRec<SLOTS> &rec) {
  const ru32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte, 0, kStreamLoadNT);
  rec.w[0] = v.x;
  rec.w[1] = v.y;
  rec.w[2] = v.z;
  rec.w[3] = v.w;
  if constexpr (SLOTS == 15) {
    const ru32x4 u = __builtin_amdgcn_raw_buffer_load_b128(r, byte + 16u, 0, kStreamLoadNT);
    rec.w[4] = u.x;
    rec.w[5] = u.y;
    rec.w[6] = u.z;
    rec.w[7] = u.w;
  }
}

// What one WAVE needs of a tile descriptor (20 ints, tfem_rings_host.cpp): the wave owns the
// tile's rows [row0, row1), at most 64.  In a plan with consecutive-vertex tiles those rows
// are the consecutive vertices gid0, gid0 + 1, ... and their CSR entries start at rs0: neither
// the ids of the owned vertices nor the row offsets are read from memory.  Scalar loads with a
// wave-uniform index (the plan is immutable during the launch: constant address space).
struct RingDesc {
  int vert_off, n_vert, row_off, n_own, row0, row1, gid0, rs0, elem_off, n_elem, elem_mode, tvert_off;
  int n_tv;  // elements the tile evaluates itself (source programs)
};
constexpr int kRingElemRuns = 8;  // desc[18] = 1: the tile's elements are <= 8 runs of consecutive ids

template <bool CHUNK>
__device__ __forceinline__ RingDesc ring_desc(const unsigned char *plan, unsigned off_desc, int tile,
                                              int wave) {
  ring_const_i32 d = (ring_const_i32)(uintptr_t)(plan + off_desc + 80u * unsigned(tile));
  RingDesc r{d[0], d[1], d[2], d[7], d[3 + wave], d[4 + wave], 0, 0, d[16], d[17], d[18] & 0xFF, d[19], d[18] >> 8};
  if (CHUNK) {
    r.gid0 = d[8 + wave];
    r.rs0 = d[12 + wave];
  }
  return r;
}

// ---------------------------------------------------------------------------------------
// Persistent, pipelined kernel: workgroups stay resident and walk a strided list of tiles
// inside their XCD's piece of the curve (workgroup b: XCD b & 7, so every XCD -- own L2 --
// walks one contiguous piece and neighbouring tiles share their halo coordinates in that L2).
// Every lane owns one row; it also fetches the coordinates of its row's vertex and of one halo
// vertex of the tile.  Iteration k (tile k: record and row offset in registers, coordinates in
// xy[k & 1]):
//   A  issue the loads of tile k+1 (row record, row offset, coordinates by the vertex ids that
//      arrived during iteration k-1) and the vertex ids of tile k+2
//   B  rows of tile k: LDS reads, arithmetic, entries -> the wave's stage
//   C  s_waitcnt vmcnt(0): the loads of A were issued a whole row phase ago, the stores of
//      tile k-1 a whole iteration ago
//   D  coordinates of tile k+1 -> xy[(k+1) & 1]; stage -> global stores of tile k
//   E  LDS barrier (the only one): xy[(k+1) & 1] is complete, nobody reads xy[k & 1] any more
// Neither a load's latency nor a store's acknowledgement is waited for inside an iteration.
// DBG = true is the ablation build of tools/time_rings.py (flags in RingArgs); its results are
// wrong by design and the product path never uses it.
// ---------------------------------------------------------------------------------------
constexpr int kRingHaloCap = kRingBlock;  // halo vertices per tile: one per lane
constexpr int kRingElemPerLane = 3;       // elements staged per tile <= 3 * kRingBlock

__device__ __forceinline__ void ring_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

#ifndef TFEM_NT_FQ
#define TFEM_NT_FQ 0
#endif
constexpr int kFqLoadNT = TFEM_NT_FQ ? 2 : 0;
// Q source values of one element (load vector): 16-byte loads where the type allows.
template <typename T, int QL>
__device__ __forceinline__ void ring_load_fq(ring_rsrc_t r, unsigned byte, T (&v)[QL > 0 ? QL : 1]) {
#pragma unroll
  for (int q = 0; q + 1 < QL; q += 2) {
    if constexpr (sizeof(T) == 8) {
      const ru32x4 x = __builtin_amdgcn_raw_buffer_load_b128(r, byte + unsigned(q) * 8u, 0, kFqLoadNT);
      v[q] = __builtin_bit_cast(double, ru32x2{x.x, x.y});
      v[q + 1] = __builtin_bit_cast(double, ru32x2{x.z, x.w});
    } else {  // two dword loads: raw_buffer_load_b64 is miscompiled by this hipcc (tfem_tiles.hip)
      v[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte + unsigned(q) * 4u, 0, 0));
      v[q + 1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte + unsigned(q) * 4u + 4u, 0, 0));
    }
  }
  if (QL & 1) {
    const unsigned o = byte + unsigned(QL - 1) * unsigned(sizeof(T));
    if constexpr (sizeof(T) == 8) {
      const unsigned lo = __builtin_amdgcn_raw_buffer_load_b32(r, o, 0, 0);
      const unsigned hi = __builtin_amdgcn_raw_buffer_load_b32(r, o + 4u, 0, 0);
      v[QL > 0 ? QL - 1 : 0] = __builtin_bit_cast(double, ru32x2{lo, hi});
    } else {
      v[QL > 0 ? QL - 1 : 0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, o, 0, 0));
    }
  }
}

// QL = 0: matrix only.  QL = Q > 0: the launch also forms the load vector
//   f_v = sum over the triangles T of the fan  det_T * g[T][loc],
//   g[T][i] = sum_q fq[T][q] l_i(q) w_q / 2
// (abstract_basis.py:95-112 with basis.py:93-96).  The source values of a tile's elements
// (plan: tile_elems, ascending) are fetched ONCE per tile with coalesced 16-byte loads -- three
// elements per lane, prefetched like the coordinates -- reduced to the three numbers g[T][.]
// and staged in LDS; a row reads one of them per fan slot by the slot's
// 12-bit code (tile-local element | loc << 10).  (Gathering the Q values per row and slot
// instead is bound by the texture addresser: 14 scattered loads per row.)
// SRC (1: any program, one element per pass of the interpreter; 2: programs that hold at most two
// values, three elements per pass): the source values are not read from memory but computed in the launch from the program
// in RingArgs::src: per tile element the three tile-local vertex ids arrive with the plan
// (tile_tverts, 4 bytes instead of 8 Q), the integration points are formed from the
// coordinates the tile holds in LDS anyway, f is evaluated there (tfem_source.hpp) and reduced
// to g[T][.] as above.  That happens at the START of the tile's iteration (its coordinates are
// complete after the previous iteration's barrier), behind the issue of the next tile's loads.
template <typename T, int SLOTS, bool MASS, bool CHUNK, int QL, bool DBG, bool KMAT = true, int SRC = 0>
// The matrix-only 15-slot instantiations are asked for 3 waves per SIMD: left alone, hipcc's
// scheduler hoists every LDS read of the unrolled fan loop and ends at 250 VGPRs (2 waves); with
// the bound it needs 112-128 and nothing spills (the load-vector instantiations would spill).
__global__ __launch_bounds__(kRingBlock, (SRC && SLOTS == 7 && QL <= 4) ? 4 : (SLOTS > 7 && QL == 0) ? 3 : 1) void k_p1_rings(const RingArgs<T> a) {
  constexpr bool LOAD = QL > 0;
  constexpr bool FQ = LOAD && !SRC;  // source values streamed from memory
  static_assert(KMAT || LOAD, "nothing to assemble");
  static_assert(LOAD || !SRC, "a source program needs a load vector");
  constexpr int kEW = (12 * SLOTS + 31) / 32;  // dwords of packed 12-bit slot codes per row: 3 or 6
  extern __shared__ __attribute__((aligned(16))) unsigned char ring_smem[];
  T *xy = reinterpret_cast<T *>(ring_smem);                      // [2][2 * lds_vert]
  T *stage = xy + 4 * a.lds_vert;                                // [waves][stage entries]
  // [3 * lds_elem + 4]; the load-vector-only instantiation has no stage
  T *gtab = stage + (KMAT ? kRingWaves * ring_stage_entries<T, SLOTS>() : 0);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  T *my_stage = stage + wave * ring_stage_entries<T, SLOTS>();
  // tiles per XCD, a whole number of blocks when the tile list is dealt in blocks
  const int deal = a.xcd_interleave > 0 ? a.xcd_interleave : 1;
  const int per = (a.n_tiles + 8 * deal - 1) / (8 * deal) * deal;
  const int xcd = blockIdx.x & 7;
  const int j0 = blockIdx.x >> 3;
  const int stride = gridDim.x >> 3;
  // SRC: positions of the plan's chain order instead of tile numbers.  Blocks of chain_len
  // positions (aligned to multiples of chain_len in the plan's order, whatever range the launch
  // takes: the plan hands shares over inside such blocks) are dealt to the XCDs round-robin; a
  // workgroup takes the positions of a block one after the other.
  const int clen = SRC ? a.chain_len : 1;
  const int u_end = a.u_first + a.n_tiles;
  const int block0 = SRC ? a.u_first / clen : 0;
  // SRC, runs: the workgroup takes runs blockIdx, blockIdx + gridDim, ... of the plan's run list (with
  // gridDim = the number of runs: its own one) and walks their positions one after the other
  const bool balanced = SRC && a.n_runs > 0;
  int run = int(blockIdx.x) - int(gridDim.x), run_at = 0, run_end = 0;
  auto next_position = [&]() {  // -1: the workgroup's share is done
    while (run_at == run_end) {
      run += int(gridDim.x);
      if (run >= a.n_runs) return -1;
      ring_const_i32 first = (ring_const_i32)(uintptr_t)(a.plan + a.off_runs);
      run_at = first[run];
      run_end = first[run + 1];
    }
    return run_at++;
  };
  auto tile_at = [&](int k) {
    if constexpr (SRC != 0) {
      ring_const_i32 order = (ring_const_i32)(uintptr_t)(a.plan + a.off_chain);
      if (balanced) {  // called with k = 0, 1, 2, ...: the positions in the workgroup's order
        const int u = next_position();
        return u < 0 ? -1 : __builtin_amdgcn_readfirstlane(order[u]);
      }
      // the workgroup of the launch's first block starts inside it when the range does
      const int k2 = k + ((xcd == 0 && j0 == 0) ? a.u_first - block0 * clen : 0);
      const int kb = k2 / clen, ki = k2 - kb * clen;
      const int u = (block0 + (j0 + kb * stride) * 8 + xcd) * clen + ki;
      if (u >= u_end) return -1;
      return __builtin_amdgcn_readfirstlane(order[u]);
    } else {
      const int j = j0 + k * stride;
      const int g = a.xcd_interleave;
      const int t = g ? ((j / g) * 8 + xcd) * g + j % g : xcd * per + j;
      return __builtin_amdgcn_readfirstlane((j < per && t < a.n_tiles) ? t : -1);
    }
  };
  const ring_rsrc_t r_coords = ring_rsrc(a.coords, a.coords_bytes);
  const ring_rsrc_t r_plan = ring_rsrc(a.plan, a.plan_bytes);
  const ring_rsrc_t r_vals = ring_rsrc(a.vals, a.vals_bytes);
  const ring_rsrc_t r_fq = ring_rsrc(a.fq, a.fq_bytes);
  const ring_rsrc_t r_fout = ring_rsrc(a.fout, a.fout_bytes);
  constexpr unsigned kRecBytes = unsigned(4 * RingRec<SLOTS>::kWords);
  constexpr unsigned kNone = 0x3FFFFFFu;  // row / vertex index behind every array: loads give 0
  if (FQ && tid < 4)  // the spare entries slots without a triangle read (times a zero determinant)
    gtab[3 * a.lds_elem + tid] = T(0);
  // SRC: the same LDS region holds the load-vector accumulators: THREE buffers (tile number modulo
  // 3) of one sum per tile-LOCAL vertex, halo included, + one entry that stays zero.  A tile adds
  // every share of every element it evaluates to the sum of the element's vertex -- no test, no
  // table.  A row takes its own sum from its tile's buffer and, when the tile before it in the chain
  // block evaluated elements of its fan (an element shared by consecutive tiles of a block is
  // evaluated by the earlier one only), what that tile summed for its vertex from the PREVIOUS
  // tile's buffer (plan: hand_in, the vertex's local id there).  A buffer is zeroed two tiles
  // after it was filled: behind the barrier that ends the iteration of the last tile reading it.
  const int acc_stride = a.lds_vert + 2;
  T *accs = gtab;  // [3][acc_stride]
  if (SRC)
    for (int i = tid; i < 3 * acc_stride; i += kRingBlock) accs[i] = T(0);

  // SRC: the program, one operation per lane, for the whole launch
  SrcLanes<T> prog;
  if constexpr (SRC) prog = src_load_lanes<T>(src_in_kernarg<T>(__builtin_offsetof(RingArgs<T>, src)));
  RingRec<SLOTS> rec, rec_ld;
  uint32_t se[FQ ? kEW : 1], se_ld[FQ ? kEW : 1];  // slot codes of the row (load vector from fq)
  unsigned eid[FQ ? kRingElemPerLane : 1], eid_ld[FQ ? kRingElemPerLane : 1];  // element ids, like gid_*
  T fqe_ld[FQ ? kRingElemPerLane : 1][QL > 0 ? QL : 1];  // their source values
  unsigned tev[SRC ? kRingElemPerLane : 1], tev_ld[SRC ? kRingElemPerLane : 1];  // local vertex triples
  unsigned gid_row = 0;                                    // vertex of this lane's current row
  int rowstart = 0, rowstart_ld = 0;
  unsigned gid_own = 0, gid_halo = 0;        // vertex ids of the tile whose coordinates load next
  unsigned gid_own_ld = 0, gid_halo_ld = 0;  // ... and of the tile after it
  T own_ld[2], halo_ld[2];
  unsigned hin = 0xFFFFu, hin_ld = 0xFFFFu;  // hand-in of this lane's current row, of the next tile's

  // vertex ids of a tile: the lane's own row vertex and halo vertex number tid.  With
  // consecutive-vertex tiles the own id is arithmetic on the descriptor.
  auto load_ids = [&](const RingDesc &d, unsigned &g_own, unsigned &g_halo) {
    const int r = d.row0 + lane;
    if (CHUNK)
      g_own = unsigned(d.gid0 + lane);
    else
      g_own = __builtin_amdgcn_raw_buffer_load_b32(
          r_plan, a.off_gid + (r < d.row1 ? unsigned(d.vert_off + r) : kNone) * 4u, 0, 0);
    const int h = d.n_own + tid;
    g_halo = __builtin_amdgcn_raw_buffer_load_b32(
        r_plan, a.off_gid + (h < d.n_vert ? unsigned(d.vert_off + h) : kNone) * 4u, 0, 0);
  };
  auto load_eids = [&](const RingDesc &d, unsigned (&e)[FQ ? kRingElemPerLane : 1]) {
    if (!FQ || (DBG && (a.flags & 64))) return;  // ablation: no element-id loads
    if (d.elem_mode) return;                        // runs of consecutive ids: nothing to fetch
#pragma unroll
    for (int j = 0; j < kRingElemPerLane; ++j) {
      const int l = tid + j * kRingBlock;
      e[FQ ? j : 0] = __builtin_amdgcn_raw_buffer_load_b32(
          r_plan, a.off_telems + (l < d.n_elem ? unsigned(d.elem_off + l) : kNone) * 4u, 0, 0);
    }
  };
  // source values of the tile's elements by the ids that arrived an iteration earlier (lanes
  // past the tile's last element carry the id 0 of the zero-filled load: harmless)
  auto load_fq = [&](const RingDesc &d, const unsigned (&e)[FQ ? kRingElemPerLane : 1]) {
    if (!FQ || (DBG && (a.flags & 32))) return;  // ablation: no source-value loads
    if (d.elem_mode) {
      // position l in the tile's ascending element list -> id, from the first ids of the runs
      // and the list positions they end at (16 scalars of the plan)
      ring_const_i32 rg = (ring_const_i32)(uintptr_t)(a.plan + a.off_telems + 4u * unsigned(d.elem_off));
      int first[kRingElemRuns], upto[kRingElemRuns];
#pragma unroll
      for (int r = 0; r < kRingElemRuns; ++r) {
        first[r] = rg[r];
        upto[r] = rg[kRingElemRuns + r];
      }
#pragma unroll
      for (int j = 0; j < kRingElemPerLane; ++j) {
        const int l = tid + j * kRingBlock;
        int id = first[0] + l;
#pragma unroll
        for (int r = 1; r < kRingElemRuns; ++r) id = l >= upto[r - 1] ? first[r] + (l - upto[r - 1]) : id;
        ring_load_fq<T, QL>(r_fq, l < d.n_elem ? unsigned(id) * unsigned(QL * sizeof(T)) : 0xFFFFFFF0u,
                            fqe_ld[FQ ? j : 0]);
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < kRingElemPerLane; ++j) {
      const int l = tid + j * kRingBlock;
      ring_load_fq<T, QL>(r_fq, l < d.n_elem ? e[FQ ? j : 0] * unsigned(QL * sizeof(T)) : 0xFFFFFFF0u,
                          fqe_ld[FQ ? j : 0]);
    }
  };
  // g[T][i] = sum_q fq[T][q] l_i(q) w_q / 2 of the elements just loaded -> LDS
  auto park_g = [&](const RingDesc &d, T *dst) {
    if (!FQ || (DBG && (a.flags & 128))) return;  // ablation: no reduction / staging of g
#pragma unroll
    for (int j = 0; j < kRingElemPerLane; ++j) {
      const int l = tid + j * kRingBlock;
      if (l < d.n_elem) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          T g = T(0);
#pragma unroll
          for (int q = 0; q < QL; ++q) g = g + fqe_ld[FQ ? j : 0][q] * a.lamw[i][q];
          dst[3 * l + i] = g;
        }
      }
    }
  };
  // SRC: the packed local vertex ids of the tile's elements (zero-filled past the last one)
  // Which lanes take which elements of a tile's table ROTATES over the waves from tile to tile
  // (`rot` = number of the tile in the workgroup's sequence): wave w takes the elements
  // ((w + rot) & 3) * 64 + lane + 256 j.  A table of ~520 elements holds 8 in its third round, which
  // one wave sees; wave w of every workgroup sits on SIMD w, so without the rotation SIMD 0 would run
  // three rounds for every tile while the others run two.
  auto load_tverts = [&](const RingDesc &d, unsigned (&tv)[SRC ? kRingElemPerLane : 1], int rot) {
    if (!SRC) return;
    const int vtid = (((wave + rot) & (kRingWaves - 1)) << 6) + lane;
#pragma unroll
    for (int j = 0; j < kRingElemPerLane; ++j) {
      const int l = vtid + j * kRingBlock;
      tv[SRC ? j : 0] = __builtin_amdgcn_raw_buffer_load_b32(
          r_plan, a.off_tverts + (l < d.n_tv ? unsigned(d.tvert_off + l) : kNone) * 4u, 0, kStreamLoadNT);
    }
  };
  // SRC: det_T g[T][i], g[T][i] = sum_q f(x_q) l_i(q) w_q / 2, of the tile's elements from the
  // tile's coordinates in LDS (basis.py:90-91: x_q = bar(q)^T X), added to the accumulators of
  // the elements' vertices in LDS (element form: no slot codes, nothing per fan slot)
  // one share det_T g[T][i] -> the accumulator the tile's table names for local vertex `lid`
  auto add_share = [&](unsigned lid, T *acc, T v) {
    if (TFEM_SRC_ABL & 4) {
      if (v == T(-1.2345e300)) acc[lid] = v;
    } else {
      atomicAdd(acc + lid, v);
    }
  };
  // NE elements per lane through the wide interpreter, then their shares
  auto compute_g_wide = [&](auto ne_tag, const RingDesc &d, const unsigned (&tv)[SRC ? kRingElemPerLane : 1],
                            const T *xyc, T *acc, int vtid) {
    constexpr int kNE = decltype(ne_tag)::value;
    unsigned codes[kNE];
#pragma unroll
    for (int j = 0; j < kNE; ++j) codes[j] = tv[SRC ? j : 0];
    T fv[kNE * (QL > 0 ? QL : 1)];
    src_run_wide<T, (QL > 0 ? QL : 1), kNE>(prog, xyc, codes, a.lam, fv);
#pragma unroll
    for (int j = 0; j < kNE; ++j) {
      const int l = vtid + j * kRingBlock;
      if (l < d.n_tv) {
        const unsigned code = codes[j];
        T x0, y0, x1, y1, x2, y2;
        lds_xy(xyc, code & 0x3FFu, x0, y0);
        lds_xy(xyc, (code >> 10) & 0x3FFu, x1, y1);
        lds_xy(xyc, (code >> 20) & 0x3FFu, x2, y2);
        const T det = (x1 - x0) * (y2 - y0) - (x2 - x0) * (y1 - y0);
        if constexpr (QL == 4 && TFEM_SRC_QSYM) {
          // the order-3 rule's structure (RingArgs::qsym): 7 operations instead of 16
          const T *f4 = fv + j * 4;
          const T base = src_fma<T>(a.qsym[3], f4[0], a.qsym[4] * ((f4[1] + f4[2]) + f4[3]));
          add_share(code & 0x3FFu, acc, det * src_fma<T>(a.qsym[5], f4[3], base));
          add_share((code >> 10) & 0x3FFu, acc, det * src_fma<T>(a.qsym[5], f4[1], base));
          add_share((code >> 20) & 0x3FFu, acc, det * src_fma<T>(a.qsym[5], f4[2], base));
        } else {
          T fw[QL > 0 ? QL : 1];
#pragma unroll
          for (int q = 0; q < QL; ++q) fw[q] = fv[j * (QL > 0 ? QL : 1) + q] * a.hw[q];
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            T g = T(0);
#pragma unroll
            for (int q = 0; q < QL; ++q) g = g + fw[q] * a.lam[i][q];
            add_share((code >> (10 * i)) & 0x3FFu, acc, det * g);
          }
        }
      }
    }
  };
  // SRC: det_T g[T][i], g[T][i] = sum_q f(x_q) l_i(q) w_q / 2, of the elements the tile evaluates
  // from the tile's coordinates in LDS (basis.py:90-91: x_q = bar(q)^T X), added to the
  // accumulators of the elements' vertices in LDS (element form: no slot codes, nothing per fan
  // slot).  Lane `tid` takes the elements tid, tid + 256, tid + 512 of the tile's table; a wave
  // runs the interpreter on as many of them as ITS lanes have (wave-uniform): of a tile's ~520
  // elements the third round holds 8, which only wave 0 sees.
  auto compute_g = [&](const RingDesc &d, const unsigned (&tv)[SRC ? kRingElemPerLane : 1], const T *xyc,
                       T *acc, int rot) {
    if (TFEM_SRC_ABL & 16) return;
    const int vwave = (wave + rot) & (kRingWaves - 1);
    const int vtid = (vwave << 6) + lane;
    // rounds of this wave: elements vwave * 64 + 256 j + lane < n_tv
    const int rounds = (d.n_tv - vwave * 64 + kRingBlock - 1) / kRingBlock;
    if (rounds <= 0) return;  // wave-uniform
    if constexpr (SRC == 2) {
      if ((rounds >= 3 || (TFEM_SRC_ABL & 32)) && !(TFEM_SRC_ABL & 1))
        compute_g_wide(std::integral_constant<int, 3>{}, d, tv, xyc, acc, vtid);
      else
        compute_g_wide(std::integral_constant<int, 2>{}, d, tv, xyc, acc, vtid);
    } else if constexpr (SRC == 1) {
      // the loop is not unrolled (one copy of the interpreter): every pass takes entry 0 and
      // rotates the array -- a register array indexed by the loop counter would go to scratch
      unsigned codes[kRingElemPerLane];
#pragma unroll
      for (int j = 0; j < kRingElemPerLane; ++j) codes[j] = tv[SRC ? j : 0];
#pragma unroll 1
      for (int j = 0; j < kRingElemPerLane; ++j) {
        if (j >= rounds) break;  // wave-uniform
        const int l = vtid + j * kRingBlock;
        const unsigned code = codes[0];
#pragma unroll
        for (int r = 0; r + 1 < kRingElemPerLane; ++r) codes[r] = codes[r + 1];
        T x0, y0, x1, y1, x2, y2;
        lds_xy(xyc, code & 0x3FFu, x0, y0);
        lds_xy(xyc, (code >> 10) & 0x3FFu, x1, y1);
        lds_xy(xyc, (code >> 20) & 0x3FFu, x2, y2);
        T xq[QL > 0 ? QL : 1], yq[QL > 0 ? QL : 1], fv[QL > 0 ? QL : 1];
#pragma unroll
        for (int q = 0; q < QL; ++q) {
          xq[q] = (a.lam[0][q] * x0 + a.lam[1][q] * x1) + a.lam[2][q] * x2;
          yq[q] = (a.lam[0][q] * y0 + a.lam[1][q] * y1) + a.lam[2][q] * y2;
        }
        src_run<T, (QL > 0 ? QL : 1)>(prog, xq, yq, fv);
        if (l < d.n_tv) {
          // signed determinant in the element's own vertex order (element_tri.py:139)
          const T det = (x1 - x0) * (y2 - y0) - (x2 - x0) * (y1 - y0);
          T fw[QL > 0 ? QL : 1];
#pragma unroll
          for (int q = 0; q < QL; ++q) fw[q] = fv[q] * a.hw[q];
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            T g = T(0);
#pragma unroll
            for (int q = 0; q < QL; ++q) g = g + fw[q] * a.lam[i][q];
            add_share((code >> (10 * i)) & 0x3FFu, acc, det * g);
          }
        }
      }
    }
  };

  auto load_tile = [&](const RingDesc &d, unsigned g_own, unsigned g_halo) {
    const int r = d.row0 + lane;
    const unsigned row = r < d.row1 ? unsigned(d.row_off + r) : kNone;
    if (!(DBG && (a.flags & 16))) {
      ring_load_rec<SLOTS>(r_plan, a.off_rows + row * kRecBytes, rec_ld);
      if (!CHUNK)
        rowstart_ld = int(__builtin_amdgcn_raw_buffer_load_b32(r_plan, a.off_rowstart + row * 4u, 0, 0));
      if (FQ) {  // 12-bit slot codes: 3 (SLOTS 7) or 6 dwords per row
        const unsigned eb = a.off_elems + row * unsigned(4 * kEW);
#pragma unroll
        for (int i = 0; i < kEW; i += 3) {
          const ru32x3 v = __builtin_amdgcn_raw_buffer_load_b96(r_plan, eb + unsigned(4 * i), 0, kStreamLoadNT);
          se_ld[FQ ? i : 0] = v.x;
          se_ld[FQ ? i + 1 : 0] = v.y;
          se_ld[FQ ? i + 2 : 0] = v.z;
        }
      }
    }
    if (!(DBG && (a.flags & 4))) {
      ring_load_xy<T>(r_coords, g_own, own_ld[0], own_ld[1]);
      ring_load_xy<T>(r_coords, g_halo, halo_ld[0], halo_ld[1]);
    }
    if (SRC)  // local id of the row's vertex in the previous tile of the chain block (0xFFFF: none)
      hin_ld = __builtin_amdgcn_raw_buffer_load_b16(r_plan, a.off_hin + row * 2u, 0, 0);
  };
  auto park = [&](const RingDesc &d, T *dst) {
    const int r = d.row0 + lane;
    if (r < d.row1) {
      dst[2 * r] = own_ld[0];
      dst[2 * r + 1] = own_ld[1];
    }
    const int h = d.n_own + tid;
    if (h < d.n_vert) {
      dst[2 * h] = halo_ld[0];
      dst[2 * h + 1] = halo_ld[1];
    }
  };

  int t_c = tile_at(0);
  if (t_c < 0) return;  // whole workgroup, before any barrier
  int t_n = tile_at(1), t_nn = tile_at(2);
  RingDesc dc = ring_desc<CHUNK>(a.plan, a.off_desc, t_c, wave);
  RingDesc dn = ring_desc<CHUNK>(a.plan, a.off_desc, t_n >= 0 ? t_n : t_c, wave);
  RingDesc dnn = ring_desc<CHUNK>(a.plan, a.off_desc, t_nn >= 0 ? t_nn : t_c, wave);
  // prologue: tile 0 taken over, vertex ids of tile 1 in registers
  load_ids(dc, gid_own, gid_halo);
  load_eids(dc, eid);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  load_tile(dc, gid_own, gid_halo);
  load_fq(dc, eid);
  load_tverts(dc, tev_ld, 0);
  if (t_n >= 0) {
    load_ids(dn, gid_own_ld, gid_halo_ld);
    load_eids(dn, eid_ld);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  park(dc, xy);
  park_g(dc, gtab);
  rec = rec_ld;
  hin = hin_ld;
  rowstart = rowstart_ld;
  gid_row = gid_own;
#pragma unroll
  for (int i = 0; i < (FQ ? kEW : 1); ++i) se[i] = se_ld[i];
#pragma unroll
  for (int j = 0; j < (FQ ? kRingElemPerLane : 1); ++j) eid[j] = eid_ld[j];
#pragma unroll
  for (int j = 0; j < (SRC ? kRingElemPerLane : 1); ++j) tev[j] = tev_ld[j];
  gid_own = gid_own_ld;
  gid_halo = gid_halo_ld;
  __syncthreads();

#if defined(TFEM_SRC_STAGGER)
  if (SRC) {  // developer experiment: workgroups start out of phase
    const int ph = TFEM_SRC_STAGGER == 1 ? int(blockIdx.x >> 3) & 3 : TFEM_SRC_STAGGER == 2 ? int(blockIdx.x >> 8) & 3
                                                                                             : int(blockIdx.x >> 5) & 3;
    for (int i = 0; i < ph * TFEM_SRC_STAGGER_N; ++i) __builtin_amdgcn_s_sleep(127);
  }
#endif
  int cur = 0;
  int a_cur = 0, a_prev = 2, a_next = 1;  // accumulator buffers of this tile, the one before, the one after
#ifdef TFEM_SRC_TIMING
  const bool timing = a.stamps != nullptr;  // developer build: phase stamps of the source-program launch
#else
  const bool timing = DBG && (a.flags & 256);
#endif
  unsigned long long tsum[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  // the clock the chip holds: shader cycles (s_memtime) against the constant 100 MHz counter
  // (s_memrealtime) over the wave's whole tile loop (MI355X_MICROARCH.md, DVFS give-back item 6)
  unsigned long long clk0 = 0, real0 = 0;
  if (timing) {
    clk0 = ring_stamp();
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(real0)::"memory");
  }
  for (int k = 0;; ++k) {
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, t6 = 0, t7 = 0, tg = 0, tb = 0;
    if (timing) t0 = ring_stamp();
    if (SRC) {
      // The vector pipe serves a SIMD's OLDEST wave first: of four equally loaded workgroups on a CU
      // the first one placed finished its loop after 100 us, the last after 190
      // (profiles/r03_wave_loop_spread.log).  The workgroups take turns in issue priority, tile by tile.
#ifndef TFEM_SRC_PRIO_SHIFT
#define TFEM_SRC_PRIO_SHIFT 3
#endif
      switch (TFEM_SRC_PRIO_SHIFT < 0 ? 0 : (k + int(blockIdx.x >> (TFEM_SRC_PRIO_SHIFT < 0 ? 0 : TFEM_SRC_PRIO_SHIFT))) & 3) {
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
      }
    }
#ifndef TFEM_SRC_PRIO
#define TFEM_SRC_PRIO 0
#endif
    auto phase_a = [&]() {
      if (t_n >= 0) {
        load_tile(dn, gid_own, gid_halo);
        load_fq(dn, eid);
        load_tverts(dn, tev_ld, k + 1);
        if (t_nn >= 0) {
          load_ids(dnn, gid_own_ld, gid_halo_ld);
          load_eids(dnn, eid_ld);
        }
      }
    };
    if (SRC && (TFEM_SRC_PRIO & 2)) phase_a();
    if (SRC) {
      // ---- G ---- source values of tile k (its coordinates are complete).  Before A: the
      // registers of tile k+1's loads are not live while the program runs (3 workgroups per CU)
      if (TFEM_SRC_PRIO & 1) __builtin_amdgcn_s_setprio(0);
      compute_g(dc, tev, xy + cur * 2 * a.lds_vert, accs + a_cur * acc_stride, k);
      if (TFEM_SRC_PRIO & 1) __builtin_amdgcn_s_setprio(3);
      // (the barrier behind G stands in front of the rows' read of their sums: the next tile's
      // loads and the rows' own arithmetic need nothing of G)
    }
    if (timing) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      tg = ring_stamp();
    }
    // ---- A ----
    if (!(SRC && (TFEM_SRC_PRIO & 2))) phase_a();
    if (timing) t1 = ring_stamp();
    // ---- B ----
    T off[SLOTS + 1], diag, sdets[SLOTS];
    // wave-uniform: shorter slot loops when no row of this wave needs the long ones
    // (7-slot records, launches bound by vector issue: the interior vertices of a structured mesh
    // have six neighbours, the seventh slot of such a wave is empty)
    constexpr int kBandSlots = (SLOTS > kRingBand && kRingBand > 0) ? kRingBand : (SLOTS == 7 && SRC ? 6 : SLOTS);
    const bool banded = kBandSlots < SLOTS && __builtin_amdgcn_ballot_w64(rec.k() > kBandSlots) == 0;
    bool regular = false;
    if (!(DBG && (a.flags & 2))) {
      const int my_row = dc.row0 + lane;
      const uint32_t lv = unsigned(my_row < dc.row1 ? my_row : 0);
      if constexpr (SLOTS == 7 && SRC) {
        // wave-uniform: every row of the wave (lanes without a row aside) is a regular one
        const bool mine = ((rec.w[2] >> 10) & 0x3FFFu) == 0x555u && rec.k() == 6;
        regular = __builtin_amdgcn_ballot_w64(!(mine || my_row >= dc.row1)) == 0;
      }
      if (regular) {
        if constexpr (SLOTS == 7) ring_row_regular<T, MASS>(a, rec, lv, xy + cur * 2 * a.lds_vert, off, diag);
      } else if (banded)
        ring_row<T, SLOTS, MASS, FQ, kBandSlots>(a, rec, lv, xy + cur * 2 * a.lds_vert, off, diag, sdets);
      else
        ring_row<T, SLOTS, MASS, FQ>(a, rec, lv, xy + cur * 2 * a.lds_vert, off, diag, sdets);
    } else {
      diag = T(1);
#pragma unroll
      for (int i = 0; i <= SLOTS; ++i) off[i] = T(i);
    }
    if (timing) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      t2 = ring_stamp();
    }
    int total = 0, pre = 0;
    if (KMAT && !(DBG && (a.flags & 8))) {
      if (regular) {
        if constexpr (SLOTS == 7) total = ring_stage_regular<T>(rec, off, diag, my_stage, pre, dc.row1 - dc.row0);
      } else {
        total = banded ? ring_stage<T, SLOTS, kBandSlots>(rec, off, diag, my_stage, pre)
                       : ring_stage<T, SLOTS>(rec, off, diag, my_stage, pre);
      }
    }
    T facc = T(0);
    if (timing) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      tb = ring_stamp();
    }
    if (SRC) {  // the row's sum is complete behind the barrier: one LDS read
      if (!(TFEM_SRC_ABL & 2)) ring_lds_barrier();
      const int my_row = dc.row0 + lane;
      const unsigned from = (hin & 0xFFFFu) == 0xFFFFu ? unsigned(a.lds_vert) : (hin & 0xFFFFu);  // lds_vert: the zero
      facc = accs[a_cur * acc_stride + (my_row < dc.row1 ? my_row : a.lds_vert)] + accs[a_prev * acc_stride + from];
    }
    if (FQ) {
      const T *g = gtab;
#pragma unroll
      for (int i = 0; i < SLOTS; ++i) {
        // code: tile-local element | loc << 10; 0xFFF (no triangle) reads the spare entries
        // behind the table, its determinant is 0
        const int bit = 12 * i, w0 = bit / 32, sh = bit % 32;  // constants once the loop is unrolled
        const uint32_t lo = se[w0] >> sh;
        const uint32_t code = (sh > 20 ? lo | (se[w0 + 1 < kEW ? w0 + 1 : w0] << (32 - sh)) : lo) & 0xFFFu;
        const uint32_t at = code == 0xFFFu ? unsigned(3 * a.lds_elem) : 3u * (code & 0x3FFu) + (code >> 10);
        facc = facc + sdets[i] * g[at];
      }
    }
    const int kk = rec.k();
    const int len_c = kk > 0 ? kk + 1 : 0;
    if (timing) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      t3 = ring_stamp();
    }
    // ---- C ----
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0); the builtin, so that hipcc's own wait
                                         // insertion knows the loads have landed
    if (timing) t4 = ring_stamp();
    // ---- D ----
    if (SRC) {
      // the buffer the NEXT tile fills: the tile two back filled it, the rows of the tile before this
      // one were its last readers (previous iteration, in front of its closing barrier)
      for (int i = tid; i < acc_stride; i += kRingBlock) accs[a_next * acc_stride + i] = T(0);
    }
    if (t_n >= 0) {
      park(dn, xy + (cur ^ 1) * 2 * a.lds_vert);
      if (FQ) {
        // the element table is single-buffered (LDS for a third workgroup per CU): every wave
        // has read tile k's entries (the loop above) before anybody overwrites them
        ring_lds_barrier();
        park_g(dn, gtab);
      }
    }
    if (timing) t5 = ring_stamp();
    if (KMAT && !(DBG && (a.flags & 8))) {
      if (CHUNK) {  // one run per wave by construction, its CSR offset in the descriptor
        __builtin_amdgcn_wave_barrier();
        // long rows (k = 0, bit 31 of the last record word, the row's length below it) leave holes
        const bool is_long = SLOTS == 7 && dc.row0 + lane < dc.row1 && kk == 0 && (rec.w[3] >> 31) != 0u;
        if (SLOTS != 7 || __ballot(is_long) == 0ull) {
          ring_store_run1<T, SLOTS, DBG>(my_stage, total, dc.rs0, r_vals, a.flags);
        } else {
          const int true_len = is_long ? int(rec.w[3] & 0x7FFFFFFFu) : len_c;
          const int csr = dc.rs0 + wave_inclusive_scan(true_len) - true_len;
          ring_store_pieces<T>(my_stage, total, pre, csr, is_long, r_vals);
        }
        __builtin_amdgcn_wave_barrier();
      } else {
        ring_store<T, SLOTS, DBG>(my_stage, total, pre, rowstart, len_c, r_vals, a.flags);
      }
    }
    if (LOAD && dc.row0 + lane < dc.row1) {
      const unsigned byte = gid_row * unsigned(sizeof(T));
      if constexpr (sizeof(T) == 8)
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(ru32x2, facc), r_fout, byte, 0, kStreamNT);
      else
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, facc), r_fout, byte, 0, kStreamNT);
    }
    if (timing) {
      t6 = ring_stamp();
      tsum[8] += tg - t0;  // G
      tsum[9] += t3 - tb;  // barrier behind G + the read of the row's sum
      t0 = tg;
      t3 = tb;  // stage: without that barrier
      tsum[0] += t1 - t0;  // A load issue
      tsum[1] += t2 - t1;  // B rows
      tsum[2] += t3 - t2;  // stage
      tsum[3] += t4 - t3;  // vmcnt(0)
      tsum[4] += t5 - t4;  // park
      tsum[5] += t6 - t5;  // stores
      tsum[7] += 1;
    }
    if (t_n < 0) break;
    rec = rec_ld;
    hin = hin_ld;
    rowstart = rowstart_ld;
    gid_row = gid_own;
#pragma unroll
    for (int i = 0; i < (FQ ? kEW : 1); ++i) se[i] = se_ld[i];
#pragma unroll
    for (int j = 0; j < (FQ ? kRingElemPerLane : 1); ++j) eid[j] = eid_ld[j];
#pragma unroll
    for (int j = 0; j < (SRC ? kRingElemPerLane : 1); ++j) tev[j] = tev_ld[j];
    gid_own = gid_own_ld;
    gid_halo = gid_halo_ld;
    {  // accumulator buffers: tile k + 1 fills what was the clean one
      const int t = a_prev;
      a_prev = a_cur;
      a_cur = a_next;
      a_next = t;
    }
    // ---- E ----
    ring_lds_barrier();
    if (timing) {
      t7 = ring_stamp();
      tsum[6] += t7 - t6;  // barrier (and the register hand-over)
    }
    t_c = t_n;
    dc = dn;
    t_n = t_nn;
    dn = dnn;
    t_nn = tile_at(k + 3);
    if (t_nn >= 0) dnn = ring_desc<CHUNK>(a.plan, a.off_desc, t_nn, wave);
    cur ^= 1;
  }
  if (timing && a.stamps && lane == 0) {
    unsigned long long *o = a.stamps + 12 * (size_t(blockIdx.x) * kRingWaves + size_t(wave));
    for (int i = 0; i < 10; ++i) o[i] = tsum[i];
    unsigned long long real1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(real1)::"memory");
    o[10] = ring_stamp() - clk0;
    o[11] = real1 - real0;
  }
}

// SRC instantiations (tfem_rings_src.hip): nq > 0; kmat = false: the load vector alone.
// wide: the three-elements-per-pass interpreter (programs of depth <= 2)
template <typename T>
void *pick_ring_src_kernel(int slots, bool mass, bool chunk, int nq, bool kmat, bool wide);

}  // namespace tfem
