// P1 assembly over a tile plan (tfem_tiles_host.cpp): the headline kernel.
// One launch produces the CSR values of  alpha * stiffness + beta * mass  and/or the load
// vector  f_i = sum_e sum_q f(x_q) phi_i(x_q) dx_q.
//
// A tile owns a compact set of CSR rows (= vertices).  Per tile:
//   - 12-byte element records (three tile-local vertex ids, pre-scaled to LDS byte offsets,
//     + 9 four-bit column positions) are read coalesced; the tile's vertex coordinates are
//     gathered into LDS once;
//   - per element: coordinates from LDS, Jacobian, signed det, inverse, gradients, the 3x3
//     block and the 3 load entries, then ds_add_f64 into the accumulators of the rows this
//     tile owns (LDS atomics only; nothing global; rows of other tiles go to a trash area);
//   - the accumulators are streamed to the CSR value array lane-contiguously through the
//     tile's OUTPUT RUNS (groups of owned rows that are contiguous in the CSR array), each
//     value written exactly once with a plain, fully coalesced store; the load entries go
//     to f[global vertex id] the same way.
//
// k_p1_tiles_pipe: persistent 512-lane workgroups, two per CU (4 waves per SIMD), each
// walking a strided list of tiles inside its XCD's contiguous Z-order range; everything the
// next tiles need is in flight while the current tile is computed (see the kernel comment).
//
// Arithmetic: the P1 gradients are constant on an element, so the quadrature sum
// sum_q (alpha g_i.g_j + beta l_i(q) l_j(q)) w_q det/2 (abstract_basis.py:83) is evaluated
// as  alpha (g_i.g_j) (det W) + (beta M_ij) det  with W = sum_q w_q/2 and
// M_ij = sum_q (w_q/2) l_i(q) l_j(q); the load entry sum_q (f_q l_i(q)) (w_q/2 det)
// (abstract_basis.py:104) as (sum_q f_q (l_i(q) w_q/2)) det.  The constants are formed once
// on the host in the same precision -- the same numbers up to a few units of rounding
// (checked against the oracle at 1e-12; the strict operation-order versions are the atomic
// kernels in tfem_kernels.hip).
//
// HBM traffic per element ~ 13.6 B records + ~12.7 B coordinates/ids + ~1 B row/run info +
// 28 B values = ~55 B, against 48 B algorithmic; the load vector adds the 8 Q bytes of
// source values per (element, owning tile) pair and 4.5 B of element ids.
#include <hip/hip_runtime.h>

#include <cstring>

#include "tfem_common.hpp"

// This kernel trades the reference's exact operation order for fewer fp64 instructions
// (see "Arithmetic" above); let the compiler fuse multiply-adds here.
#pragma clang fp contract(fast)

namespace tfem {

// 512 lanes (8 waves) share one tile: with two workgroups per CU that is 4 waves per SIMD,
// which is what hides the dependent fp64 / LDS latency chains of the element phase.
#ifndef TFEM_TILE_BLOCK
#define TFEM_TILE_BLOCK 512
#endif
constexpr int kTileBlock = TFEM_TILE_BLOCK;
constexpr int kWaves = kTileBlock / 64;
constexpr int kElemPerLane = 2;  // tile element capacity   = 1024: two full rounds, all 8
                                 // waves equally loaded (the plan fills tiles up to it)
constexpr int kVertPerLane = 2;  // tile vertex capacity    = 1024
constexpr int kRowPerLane = 1;   // tile owned-row capacity = 512
constexpr int kAccPerLane = 8;   // tile accumulator capacity = 4096 entries
constexpr int kChunkLen = 128;   // accumulator entries per output chunk: two per lane
constexpr int kChunks = kTileBlock * kAccPerLane / kChunkLen;
constexpr int kDescStride = 12;  // ints per tile descriptor (tfem_tiles_host.cpp)
// Accumulator entries behind lds_acc that absorb the rows a tile does not own; halo vertex l
// uses entries (l & 15) .. (l & 15) + 15 so that neighbouring lanes do not pile up on one
// address.
constexpr int kTrash = 32;

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
using rsrc_t = __amdgpu_buffer_rsrc_t;

// Buffer resource over `bytes` bytes at p: loads beyond the range return 0 and stores are
// dropped by the hardware, so the tile kernel needs no index clamps, does its address
// arithmetic in 32 bits and cannot fault on a damaged plan.  (Every array of the path is
// far below 4 GiB; launch_tiles checks.)
__device__ __forceinline__ rsrc_t make_rsrc(const void *p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, int(bytes), 0x00020000);
}
template <typename T>
__device__ __forceinline__ void buf_load2(rsrc_t r, unsigned off, T &x, T &y) {
  if constexpr (sizeof(T) == 8) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    x = __builtin_bit_cast(double, u32x2{v.x, v.y});
    y = __builtin_bit_cast(double, u32x2{v.z, v.w});
  } else {
    // NB: __builtin_amdgcn_raw_buffer_load_b64 is miscompiled by this hipcc (ROCm 7.2): it
    // emits buffer_load_dword and leaves the second dword undefined.  Two dword loads.
    x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
    y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off + 4u, 0, 0));
  }
}
template <typename T>
__device__ __forceinline__ T buf_load1(rsrc_t r, unsigned off) {
  if constexpr (sizeof(T) == 8) {  // two dword loads: see buf_load2
    const unsigned lo = __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0);
    const unsigned hi = __builtin_amdgcn_raw_buffer_load_b32(r, off + 4u, 0, 0);
    return __builtin_bit_cast(double, u32x2{lo, hi});
  } else {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
  }
}
// Cache policy of the output stores.  Non-temporal stores (aux bit 1 = nt on gfx950) speed the
// ring kernel up by 15 % (tfem_rowkit.hpp) but slow this kernel down (K + f 242 -> 264 us at 1e7
// elements: its runs start at odd 8-byte entries and rely on L2 to merge partial lines).
constexpr int kTileStoreNT = 0;
template <typename T>
__device__ __forceinline__ void buf_store1(rsrc_t r, unsigned off, T x) {
  if constexpr (sizeof(T) == 8)
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, x), r, off, 0, kTileStoreNT);
  else
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x), r, off, 0, kTileStoreNT);
}
template <typename T>
__device__ __forceinline__ void buf_store2(rsrc_t r, unsigned off, T x, T y) {
  if constexpr (sizeof(T) == 8) {
    const u32x2 a = __builtin_bit_cast(u32x2, x), b = __builtin_bit_cast(u32x2, y);
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{a.x, a.y, b.x, b.y}, r, off, 0, kTileStoreNT);
  } else {
    __builtin_amdgcn_raw_buffer_store_b64(
        u32x2{__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, y)}, r, off, 0, kTileStoreNT);
  }
}

template <typename T>
struct TileArgs {
  const T *coords;
  const int32_t *desc;     // = plan + off_desc
  const unsigned char *plan;  // packed plan (tfem_tile_plan_pack)
  const T *fq;             // (n_elems, Q) source values, load vector only
  T *vals;
  T *fout;
  // array extents in bytes (buffer resources)
  unsigned coords_bytes, plan_bytes, fq_bytes, vals_bytes, fout_bytes;
  unsigned off_rec, off_gid, off_loff, off_rund, off_runl, off_eid;  // byte offsets inside plan
  int n_tiles;
  int lds_acc;   // accumulator entries reserved in LDS
  int lds_vert;  // vertex slots reserved in LDS
  int lds_own;   // owned-row slots reserved in LDS
  int lds_run;   // run slots reserved in LDS (per buffer), >= max runs + 1
  T stiff_w;     // alpha * sum_q w_q / 2
  T mass_w[6];   // beta * sum_q (w_q/2) l_i l_j for (i,j) = 00 01 02 11 12 22
  T lamw[kMaxQuad][3];  // l_i(q) * w_q / 2
  int flags;     // diagnostic build only (tfem_p1_tiles_debug)
  unsigned long long *stamps;  // diagnostic build, flag 16: 8 cycle sums per wave
};

struct TileDesc {
  int elem_off, n_elem, vert_off, n_vert, n_own, acc_size, loff_off, run_off, n_runs, lrun_off;
};

// The plan is immutable during the launch: read descriptors through the constant address
// space so that a wave-uniform descriptor is scalar loads (lgkmcnt) even after the kernel
// has stored to `vals` -- a global_load here would put a vmcnt(0) wait, and with it the
// latency of every load and store in flight, on the critical path.
typedef const int32_t __attribute__((address_space(4))) *const_i32_ptr;

__device__ __forceinline__ TileDesc load_desc(const int32_t *desc, int tile) {
  const_i32_ptr d = (const_i32_ptr)(uintptr_t)(desc + kDescStride * tile);
  return TileDesc{d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], d[8], d[9]};
}

// Diagnostic build only: shader-clock stamp (cdna_hip_programming.md section 7).
__device__ __forceinline__ unsigned long long stamp() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

// LDS-only workgroup barrier: outstanding global loads (the prefetch) stay in flight.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// One element (basis.py:87-88, element_tri.py:132-145 and :41, abstract_basis.py:83/:104):
// its 3x3 block goes to A[row v_j][col v_i] (basis.py:73-76), its 3 load entries to f[v_i].
// 12-byte record: word j = 16 * local id of vertex j | positions << 16; 8-byte record
// (REC8): word 0 = three 10-bit local ids, word 1 = nine 3-bit positions.  `loff` maps EVERY local
// vertex to the byte offset of its row's accumulators: rows this tile does not own (halo
// vertices) map to a small trash area that is never written out, so the element phase has
// no branches.  Padding lanes hold a null record (a dummy vertex whose row is trash too).
// `srcw[i]` = sum_q f_q l_i(q) w_q/2 of this element (load vector).
template <typename T, bool KMAT, bool MASS, bool LOAD, bool REC8, bool DBG>
__device__ __forceinline__ void element_to_lds(const TileArgs<T> &a, const uint32_t (&rec)[3],
                                               const T (&srcw)[3], const unsigned char *xy_bytes,
                                               const unsigned char *loff_bytes,
                                               unsigned char *acc_bytes, int n_own,
                                               int facc_byte0, int trash_byte0) {
  T x[3], y[3];
  int base[3];
  uint32_t off16[3];
#pragma unroll
  for (int v = 0; v < 3; ++v) {
    // 16 * local id: stored as such in the 12-byte record, a 10-bit field of word 0 in the
    // 8-byte record
    off16[v] = REC8 ? ((rec[0] >> (10 * v)) & 0x3FFu) << 4 : rec[v] & 0xFFFFu;
    const T *p = reinterpret_cast<const T *>(xy_bytes + (sizeof(T) == 8 ? off16[v] : off16[v] >> 1));
    x[v] = p[0];
    y[v] = p[1];
    if (KMAT) base[v] = *reinterpret_cast<const int *>(loff_bytes + (off16[v] >> 2));
  }
  const T ja = x[1] - x[0], jb = x[2] - x[0];
  const T jc = y[1] - y[0], jd = y[2] - y[0];
  const T det = ja * jd - jb * jc;  // signed (element_tri.py:139)
  if (KMAT) {
    const T r = T(1) / det;
    const T i00 = r * jd, i01 = -(r * jb), i10 = -(r * jc), i11 = r * ja;
    const T g[3][2] = {{-(i00 + i10), -(i01 + i11)}, {i00, i01}, {i10, i11}};
    const T wdet = a.stiff_w * det;
    T loc[3][3];
    int m = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = i; j < 3; ++j, ++m) {
        T v = (g[i][0] * g[j][0] + g[i][1] * g[j][1]) * wdet;
        if (MASS) v = v + a.mass_w[m] * det;
        loc[i][j] = v;
        loc[j][i] = v;
      }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int pos = REC8 ? int((rec[1] >> (3 * (3 * j + i))) & 0x7u)
                             : int((rec[j] >> (16 + 4 * i)) & 0xFu);
        T *slot = reinterpret_cast<T *>(acc_bytes + base[j] + pos * int(sizeof(T)));
        if (DBG && (a.flags & 1)) {
          if (loc[i][j] == T(-1.2345e300)) *slot = loc[i][j];  // keeps the math alive
        } else {
          atomicAdd(slot, loc[i][j]);  // ds_add_f64 / ds_add_f32
        }
      }
    }
  }
  if (LOAD) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int lid = int(off16[i] >> 4);
      const int off = lid < n_own ? facc_byte0 + lid * int(sizeof(T))
                                  : trash_byte0 + (lid & 15) * int(sizeof(T));
      atomicAdd(reinterpret_cast<T *>(acc_bytes + off), srcw[i] * det);
    }
  }
}

// ---------------------------------------------------------------------------------------
// Iteration k of a persistent workgroup (tile k current; its coordinates, row offsets, runs
// are in LDS, its records and source weights in registers):
//   S2  element phase of tile k: LDS reads, fp64, ds_add
//   --  LDS barrier, then ONE s_waitcnt vmcnt(0): the loads of tile k+1 (issued a whole
//       element phase ago) and the stores of tile k-1 have landed long since
//   S5  stream tile k's accumulators out (a wave per 64-entry chunk, lanes along the CSR
//       array: contiguous 512-byte stores), clearing each entry after reading it
//   S3  take over tile k+1 from the load registers: records (register copies, with the null
//       record for padding lanes), source weights (Q fused multiply-adds per entry),
//       coordinates and row offsets -> LDS (single buffers: S2 of tile k is over), runs
//       -> LDS (double buffer: S5 of tile k may still be reading)
//   S4  issue the loads of tile k+2 into the registers just freed (records, rows, runs,
//       coordinates and source values by the vertex / element ids already in registers)
//       + vertex and element ids of tile k+3
//   --  LDS barrier; build the output chunk table of tile k+1
// The only vector-memory wait in the loop is the explicit vmcnt(0), one full element phase
// after the loads were issued: neither load latency nor store acknowledgements sit on the
// critical path.  Rules that keep hipcc from adding more: loads are unconditional with
// clamped indices; no loop follows loads or stores inside an iteration (it drains vmcnt in
// front of loops); no select between an LDS value and a register value (it can become a
// flat_load, which waits on vmcnt too); registers written by loads are only ever read
// right after the explicit wait (copied if they must live longer).
// DBG = true is the ablation build used by tools/ablate_tiles.py: bits of a.flags switch
// off 1 = LDS atomics, 2 = the whole element phase, 4 = the value stores, 8 = the
// coordinate gather, 16 = in-kernel stamps.  Its results are wrong by design; the product
// path never uses it.
// ---------------------------------------------------------------------------------------
template <typename T, bool KMAT, bool MASS, int QL, bool REC8, bool DBG>
__global__ __launch_bounds__(kTileBlock, QL > 0 ? 4 : 6) void k_p1_tiles_pipe(const TileArgs<T> a) {
  constexpr bool LOAD = QL > 0;
  static_assert(!REC8 || kElemPerLane == 2, "8-byte records are loaded two per lane");
  // element slot e of this lane: 8-byte records are read two per lane (one 16-byte load),
  // 12-byte records one per lane and round
  auto elem_index = [&](int e) { return REC8 ? 2 * int(threadIdx.x) + e : int(threadIdx.x) + e * kTileBlock; };
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T *acc = reinterpret_cast<T *>(smem_raw);        // [lds_acc] + trash [kTrash] + facc [lds_own]
  T *xy = acc + a.lds_acc + kTrash + a.lds_own;                          // [2 * (lds_vert + 1)]
  int *loff = reinterpret_cast<int *>(xy + 2 * (a.lds_vert + 1));        // [lds_vert + 1]
  int *gid_buf = loff + a.lds_vert + 1;                                  // [3][lds_own]
  int *runl_buf = gid_buf + 3 * a.lds_own;                               // [2][lds_run]
  int *rund_buf = runl_buf + 2 * a.lds_run;                              // [2][lds_run]
  // [2][kChunks], 16-byte aligned (same rounding as the host's size computation)
  const int ctab_off = (int(reinterpret_cast<unsigned char *>(rund_buf + 2 * a.lds_run) - smem_raw) + 15) & ~15;
  int4 *ctab_buf = reinterpret_cast<int4 *>(smem_raw + ctab_off);
  const int trash = a.lds_acc * int(sizeof(T));
  const int facc0 = (a.lds_acc + kTrash) * int(sizeof(T));

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int per = (a.n_tiles + 7) / 8;
  const int xcd = blockIdx.x & 7;
  const int j0 = blockIdx.x >> 3;
  const int stride = gridDim.x >> 3;
  auto tile_at = [&](int k) {
    const int j = j0 + k * stride;
    const int t = xcd * per + j;
    return __builtin_amdgcn_readfirstlane((j < per && t < a.n_tiles) ? t : -1);
  };
  const rsrc_t r_coords = make_rsrc(a.coords, a.coords_bytes);
  // all plan arrays live in ONE allocation (tfem_tile_plan_pack): one resource, five offsets
  const rsrc_t r_plan = make_rsrc(a.plan, a.plan_bytes);
  const rsrc_t r_fq = make_rsrc(a.fq, a.fq_bytes);
  const rsrc_t r_vals = make_rsrc(a.vals, a.vals_bytes);
  const rsrc_t r_fout = make_rsrc(a.fout, a.fout_bytes);

  // registers written by loads (S4) and read right after the vmcnt(0) of the next iteration
  uint32_t rec_ld[kElemPerLane][3];
  T fq_ld[kElemPerLane][QL > 0 ? QL : 1];
  T xy_ld[kVertPerLane][2];
  int loff_ld[kRowPerLane], runl_ld[kRowPerLane], rund_ld[kRowPerLane];
  int gid[kVertPerLane];   // vertex ids, one tile further ahead
  int eid[kElemPerLane];   // element ids, one tile further ahead (load vector)
  // the current tile, as plain values
  uint32_t rec[kElemPerLane][3];
  T srcw[kElemPerLane][3];

  // Lanes past the end of a tile read the next tile's entries (or 0 past the array); what
  // they load is never used (null records, l < n_vert / n_own guards when parking).
  auto load_ids = [&](const TileDesc &d) {
#pragma unroll
    for (int v = 0; v < kVertPerLane; ++v)
      gid[v] = int(__builtin_amdgcn_raw_buffer_load_b32(
          r_plan, a.off_gid + unsigned(d.vert_off + tid + v * kTileBlock) * 4u, 0, 0));
    if (LOAD) {
#pragma unroll
      for (int e = 0; e < kElemPerLane; ++e)
        eid[e] = int(__builtin_amdgcn_raw_buffer_load_b32(
            r_plan, a.off_eid + unsigned(d.elem_off + elem_index(e)) * 4u, 0, 0));
    }
  };
  // `tile_slot` = k mod 3 of the tile being loaded: its owned vertex ids go to that LDS slot
  auto load_tile = [&](const TileDesc &d, int tile_slot) {
    if constexpr (REC8) {
      const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(
          r_plan, a.off_rec + unsigned(d.elem_off + 2 * tid) * 8u, 0, 0);
      rec_ld[0][0] = r.x;
      rec_ld[0][1] = r.y;
      rec_ld[1][0] = r.z;
      rec_ld[1][1] = r.w;
    } else {
#pragma unroll
      for (int e = 0; e < kElemPerLane; ++e) {
        const u32x3 r = __builtin_amdgcn_raw_buffer_load_b96(
            r_plan, a.off_rec + unsigned(d.elem_off + tid + e * kTileBlock) * 12u, 0, 0);
        rec_ld[e][0] = r.x;
        rec_ld[e][1] = r.y;
        rec_ld[e][2] = r.z;
      }
    }
#pragma unroll
    for (int r = 0; r < kRowPerLane; ++r) {
      const unsigned l = unsigned(tid + r * kTileBlock);
      loff_ld[r] = int(__builtin_amdgcn_raw_buffer_load_b16(r_plan, a.off_loff + (unsigned(d.loff_off) + l) * 2u, 0, 0));
      runl_ld[r] = int(__builtin_amdgcn_raw_buffer_load_b16(r_plan, a.off_runl + (unsigned(d.lrun_off) + l) * 2u, 0, 0));
      rund_ld[r] = int(__builtin_amdgcn_raw_buffer_load_b32(r_plan, a.off_rund + (unsigned(d.run_off) + l) * 4u, 0, 0));
    }
    if (!(DBG && (a.flags & 8))) {
#pragma unroll
      for (int v = 0; v < kVertPerLane; ++v)
        buf_load2<T>(r_coords, unsigned(gid[v]) * unsigned(2 * sizeof(T)), xy_ld[v][0], xy_ld[v][1]);
    }
    if (LOAD) {
#pragma unroll
      for (int e = 0; e < kElemPerLane; ++e) {
        const unsigned base = unsigned(eid[e]) * unsigned((QL > 0 ? QL : 1) * sizeof(T));
#pragma unroll
        for (int q = 0; q + 1 < QL; q += 2)
          buf_load2<T>(r_fq, base + unsigned(q * sizeof(T)), fq_ld[e][q], fq_ld[e][q + 1]);
        if (QL & 1) fq_ld[e][QL > 0 ? QL - 1 : 0] = buf_load1<T>(r_fq, base + unsigned((QL - 1) * sizeof(T)));
      }
      int *gid_w = gid_buf + tile_slot * a.lds_own;  // owned rows come first in the id list
#pragma unroll
      for (int v = 0; v < kVertPerLane; ++v) {
        const int l = tid + v * kTileBlock;
        if (l < d.n_own) gid_w[l] = gid[v];
      }
    }
  };
  auto take_tile = [&](const TileDesc &d, int run_buf) {
    // padding lanes get a null record: the dummy vertex three times, positions spread over
    // the trash entries
    const uint32_t dv = uint32_t(a.lds_vert);
    const uint32_t null_word[3] = {
        REC8 ? dv | dv << 10 | dv << 20 : (dv << 4) | (uint32_t(lane & 15) << 16),
        REC8 ? uint32_t(lane & 7) * 0x1249249u : (dv << 4) | (uint32_t(lane & 15) << 16),
        (dv << 4) | (uint32_t(lane & 15) << 16)};
#pragma unroll
    for (int e = 0; e < kElemPerLane; ++e) {
      const bool real = elem_index(e) < d.n_elem;
#pragma unroll
      for (int j = 0; j < (REC8 ? 2 : 3); ++j) rec[e][j] = real ? rec_ld[e][j] : null_word[j];
      if (LOAD) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          T s = T(0);
#pragma unroll
          for (int q = 0; q < QL; ++q) s = s + fq_ld[e][q] * a.lamw[q][i];
          srcw[e][i] = s;
        }
      }
    }
#pragma unroll
    for (int v = 0; v < kVertPerLane; ++v) {
      const int l = tid + v * kTileBlock;
      if (l < d.n_vert) {
        xy[2 * l] = xy_ld[v][0];
        xy[2 * l + 1] = xy_ld[v][1];
        loff[l] = (v < kRowPerLane && l < d.n_own)
                      ? loff_ld[v < kRowPerLane ? v : 0] * int(sizeof(T))
                      : trash + (l & 15) * int(sizeof(T));
      }
    }
    int *runl_w = runl_buf + run_buf * a.lds_run;
    int *rund_w = rund_buf + run_buf * a.lds_run;
#pragma unroll
    for (int r = 0; r < kRowPerLane; ++r) {
      const int lr = tid + r * kTileBlock;
      if (lr < d.n_runs) {
        runl_w[lr] = runl_ld[r];
        rund_w[lr] = rund_ld[r];
      }
    }
    if (tid == 0) runl_w[d.n_runs] = d.acc_size;  // sentinel: end of the last run
  };
  // Output chunk table (needs the parked runs: call after a barrier).  Chunk c = accumulator
  // entries [128c, 128c+128): x = end of the run holding entry 128c, y = delta of that run,
  // z = delta of the next run, w = index of the first run | slow << 16 (a third run starts
  // inside the chunk: lanes then search).  Fixed-trip binary search: no loop in the code.
  auto build_chunks = [&](const TileDesc &d, int buf) {
    if (!KMAT) return;
    const int n_chunks = (d.acc_size + kChunkLen - 1) / kChunkLen;
    if (tid < n_chunks) {
      const int *runl = runl_buf + buf * a.lds_run;
      const int *rund = rund_buf + buf * a.lds_run;
      const int target = tid * kChunkLen;
      int lo = 0, hi = d.n_runs;
#pragma unroll
      for (int it = 0; it < 10; ++it) {  // 2^10 >= kTileBlock * kRowPerLane runs
        const int mid = (lo + hi) >> 1;
        const bool go = hi - lo > 1;
        const bool up = go && runl[mid] <= target;
        lo = up ? mid : lo;
        hi = (go && !up) ? mid : hi;
      }
      const int nxt = lo + 1 < d.n_runs ? lo + 1 : lo;
      const bool slow = lo + 2 < d.n_runs && runl[lo + 2] < target + kChunkLen;
      ctab_buf[buf * kChunks + tid] =
          make_int4(runl[lo + 1], rund[lo], rund[nxt], lo | (slow ? 1 << 16 : 0));
    }
  };

  int t_c = tile_at(0);
  if (t_c < 0) return;  // whole workgroup, before any barrier
  int t_n = tile_at(1), t_nn = tile_at(2), t_nnn = tile_at(3);
  TileDesc dc = load_desc(a.desc, t_c);
  TileDesc dn = load_desc(a.desc, t_n >= 0 ? t_n : t_c);
  TileDesc dnn = load_desc(a.desc, t_nn >= 0 ? t_nn : t_c);
  TileDesc dnnn = load_desc(a.desc, t_nnn >= 0 ? t_nnn : t_c);

  // ---- prologue: tile 0 taken over, tile 1 in the load registers, ids of tile 2 -------------
  for (int s = tid; s < a.lds_acc + kTrash + a.lds_own; s += kTileBlock) acc[s] = T(0);
  if (tid < 2) {  // the dummy vertex (slot lds_vert): finite coordinates, trash row
    xy[2 * a.lds_vert + tid] = T(tid);
    loff[a.lds_vert] = trash;
  }
  load_ids(dc);
  load_tile(dc, 0);
  take_tile(dc, 0);
  if (t_n >= 0) {
    load_ids(dn);
    load_tile(dn, 1);
  }
  if (t_nn >= 0) load_ids(dnn);
  __syncthreads();
  build_chunks(dc, 0);
  __syncthreads();

  if (DBG && (a.flags & 32)) {  // experiment: spread the workgroups' phases over one tile period
    const int steps = (blockIdx.x >> 3) & 15;
    for (int i = 0; i < steps; ++i) __builtin_amdgcn_s_sleep(15);  // ~1000 cycles per step
  }
  int cur = 0;   // run / chunk-table buffer of the current tile
  int slot = 0;  // k mod 3: vertex-id buffer of the current tile
  unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int k = 0;; ++k) {
    const int *runl = runl_buf + cur * a.lds_run;
    const int *rund = rund_buf + cur * a.lds_run;
    const bool timing = DBG && (a.flags & 16);
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, t6 = 0;
    if (timing) t0 = stamp();
    // ---- S2 ----------------------------------------------------------------------------------
    if (!(DBG && (a.flags & 2))) {
      // waves whose 64 slots of a round lie past the end of the tile skip the round (scalar
      // branch); inside the last partial wave the padding lanes process null records
#pragma unroll
      for (int e = 0; e < kElemPerLane; ++e) {
        if ((REC8 ? 128 * wave + e : e * kTileBlock + wave * 64) < dc.n_elem)
          element_to_lds<T, KMAT, MASS, LOAD, REC8, DBG>(
              a, rec[e], srcw[e], reinterpret_cast<const unsigned char *>(xy),
              reinterpret_cast<const unsigned char *>(loff), reinterpret_cast<unsigned char *>(acc),
              dc.n_own, facc0, trash);
      }
    }
    if (timing) t1 = stamp();
    lds_barrier();
    if (timing) t2 = stamp();
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): see the comment above the kernel
    if (timing) t3 = stamp();
    // ---- S5 ----------------------------------------------------------------------------------
    // wave w streams chunks w, w+8, ...: lane l owns accumulator entries 128c + 2l, + 2l + 1;
    // their run (hence their place in the CSR array) comes from the chunk table.  Both in one
    // run: one 16-byte store, i.e. 1 KiB contiguous per wave instruction.
    if (KMAT) {
      const int4 *ctab = ctab_buf + cur * kChunks;
      const int n_chunks = (dc.acc_size + kChunkLen - 1) / kChunkLen;
#pragma unroll
      for (int u = 0; u < kChunks / kWaves; ++u) {
        const int c = wave + kWaves * u;
        if (c < n_chunks) {  // wave-uniform
          const int s0 = c * kChunkLen + 2 * lane;
          const int4 tab = ctab[c];
          const T v0 = acc[s0], v1 = acc[s0 + 1];  // entries past acc_size exist in LDS (unused)
          if (!(tab.w >> 16)) {
            const int d0 = s0 < tab.x ? tab.y : tab.z;
            const int d1 = s0 + 1 < tab.x ? tab.y : tab.z;
            unsigned long long ts0 = 0;
            if (timing) ts0 = stamp();
            if (!(DBG && (a.flags & 4))) {
              if (d0 == d1 && s0 + 1 < dc.acc_size) {
                buf_store2<T>(r_vals, (unsigned(s0 + d0) * unsigned(sizeof(T))) & ((DBG && (a.flags & 64)) ? 0xFFFF0u : ~0u), v0, v1);
              } else {  // the pair straddles two runs, or is the odd tail
                if (s0 < dc.acc_size) buf_store1<T>(r_vals, unsigned(s0 + d0) * unsigned(sizeof(T)), v0);
                if (s0 + 1 < dc.acc_size) buf_store1<T>(r_vals, unsigned(s0 + 1 + d1) * unsigned(sizeof(T)), v1);
              }
            }
            if (timing) tsum[7] += stamp() - ts0;  // cycles inside the store statements (fast chunks)
          } else {  // more than two runs meet in this chunk: each entry searches the next 128
            // runs with a fixed-trip binary search (no loop: hipcc drains vmcnt before loops)
            int lo0 = tab.w & 0xFFFF, lo1 = lo0;
            int hi0 = lo0 + kChunkLen < dc.n_runs ? lo0 + kChunkLen : dc.n_runs, hi1 = hi0;
#pragma unroll
            for (int it = 0; it < 7; ++it) {
              const int m0 = (lo0 + hi0) >> 1, m1 = (lo1 + hi1) >> 1;
              const int st0 = runl[m0], st1 = runl[m1];
              const bool g0 = hi0 - lo0 > 1, g1 = hi1 - lo1 > 1;
              lo0 = (g0 && st0 <= s0) ? m0 : lo0;
              hi0 = (g0 && st0 > s0) ? m0 : hi0;
              lo1 = (g1 && st1 <= s0 + 1) ? m1 : lo1;
              hi1 = (g1 && st1 > s0 + 1) ? m1 : hi1;
            }
            const int e0 = rund[lo0], e1 = rund[lo1];
            if (!(DBG && (a.flags & 4))) {
              if (s0 < dc.acc_size) buf_store1<T>(r_vals, unsigned(s0 + e0) * unsigned(sizeof(T)), v0);
              if (s0 + 1 < dc.acc_size) buf_store1<T>(r_vals, unsigned(s0 + 1 + e1) * unsigned(sizeof(T)), v1);
            }
          }
          if (s0 < dc.acc_size) {  // never clear beyond the tile's entries (+1: rounding slack)
            acc[s0] = T(0);
            acc[s0 + 1] = T(0);
          }
        }
      }
    }
    if (LOAD) {  // load entries of the owned vertices -> f[global id]
      const int *gid_r = gid_buf + slot * a.lds_own;
      T *facc = acc + a.lds_acc + kTrash;
#pragma unroll
      for (int r = 0; r < kRowPerLane; ++r) {
        const int l = tid + r * kTileBlock;
        if (l < dc.n_own) {
          if (!(DBG && (a.flags & 4)))
            buf_store1<T>(r_fout, unsigned(gid_r[l]) * unsigned(sizeof(T)), facc[l]);
          facc[l] = T(0);
        }
      }
    }
    if (timing) t4 = stamp();
    // ---- S3 + S4 ---------------------------------------------------------------------------------
    if (t_n >= 0) take_tile(dn, cur ^ 1);
    if (t_nn >= 0) {
      load_tile(dnn, slot == 0 ? 2 : slot - 1);  // (k + 2) mod 3
      if (t_nnn >= 0) load_ids(dnnn);
    }
    if (timing) t5 = stamp();
    lds_barrier();
    if (timing) {
      t6 = stamp();
      tsum[0] += t1 - t0;  // S2 element phase
      tsum[1] += t2 - t1;  // barrier after S2
      tsum[2] += t3 - t2;  // vmcnt(0)
      tsum[3] += t5 - t4;  // S3 + S4
      tsum[4] += t4 - t3;  // S5
      tsum[5] += t6 - t5;  // barrier after S3/S4
      tsum[6] += 1;
    }
    if (t_n < 0) break;
    build_chunks(dn, cur ^ 1);  // visible to S5 of the next tile through the barrier after its S2
    // ---- advance the tile window --------------------------------------------------------------
    t_c = t_n;
    dc = dn;
    t_n = t_nn;
    dn = dnn;
    t_nn = t_nnn;
    dnn = dnnn;
    t_nnn = tile_at(k + 4);
    if (t_nnn >= 0) dnnn = load_desc(a.desc, t_nnn);
    cur ^= 1;
    slot = slot == 2 ? 0 : slot + 1;
  }
  if (DBG && (a.flags & 16) && a.stamps && (tid & 63) == 0) {
    unsigned long long *o = a.stamps + 8 * (size_t(blockIdx.x) * kWaves + (tid >> 6));
    for (int i = 0; i < 8; ++i) o[i] = tsum[i];
  }
}

struct TileLaunch {
  const void *coords;
  int quad_order;
  double alpha, beta;
  const unsigned char *plan;     // packed plan, device
  const int64_t *layout;         // host: kPlanLayoutLen entries written by tfem_tile_plan_pack
  int64_t n_verts, n_elems, nnz;
  void *vals;       // nullptr: no matrix
  const void *fq;   // nullptr: no load vector
  void *fout;
  hipStream_t stream;
  int flags = -1;   // >= 0: diagnostic build
  unsigned long long *stamps = nullptr;
};

static int cu_count() {
  static int cached = 0;
  if (cached == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      cached = prop.multiProcessorCount;
    else
      cached = 256;
  }
  return cached;
}

// The ablation build exists for fp64 stiffness (K only, and K + f at Q = 4) alone.
template <typename T, bool KMAT, bool MASS, int QL, bool REC8>
static void *pick_kernel(bool dbg) {
  if constexpr (sizeof(T) == 8 && KMAT && !MASS && (QL == 0 || QL == 4)) {
    if (dbg) return reinterpret_cast<void *>(k_p1_tiles_pipe<T, KMAT, MASS, QL, REC8, true>);
  }
  return reinterpret_cast<void *>(k_p1_tiles_pipe<T, KMAT, MASS, QL, REC8, false>);
}

template <typename T, bool KMAT, bool MASS, bool REC8>
static void *pick_q(int nq, bool load, bool dbg) {
  if (!load) return pick_kernel<T, KMAT, MASS, 0, REC8>(dbg);
  switch (nq) {
    case 1: return pick_kernel<T, KMAT, MASS, 1, REC8>(dbg);
    case 3: return pick_kernel<T, KMAT, MASS, 3, REC8>(dbg);
    case 4: return pick_kernel<T, KMAT, MASS, 4, REC8>(dbg);
    case 6: return pick_kernel<T, KMAT, MASS, 6, REC8>(dbg);
    default: return nullptr;
  }
}

template <typename T, bool REC8>
static void *pick_form(bool kmat, bool mass, int nq, bool load, bool dbg) {
  if (!kmat) return pick_q<T, false, false, REC8>(nq, true, dbg);
  return mass ? pick_q<T, true, true, REC8>(nq, load, dbg) : pick_q<T, true, false, REC8>(nq, load, dbg);
}

template <typename T>
static int launch_tiles(const TileLaunch &L) {
  TriTables tables;
  if (!build_tri_tables(L.quad_order, int(sizeof(T)), &tables))
    return fail(TFEM_ERR_UNSUPPORTED, "Integration order not implemented");
  if (!L.layout || L.layout[0] == 0) return TFEM_OK;
  const bool kmat = L.vals != nullptr;
  const bool load = L.fq != nullptr;
  if (!kmat && !load) return fail(TFEM_ERR_INVALID_ARGUMENT, "nothing to assemble");
  if (!L.coords || !L.plan || !L.layout || (load && !L.fout))
    return fail(TFEM_ERR_INVALID_ARGUMENT, "NULL pointer");
  const int64_t *z = L.layout;  // [0..11] = plan sizes, [12..18] = byte offsets, [19] = bytes
  TileArgs<T> a;
  std::memset(&a, 0, sizeof(a));
  a.coords = static_cast<const T *>(L.coords);
  a.plan = L.plan;
  a.desc = reinterpret_cast<const int32_t *>(L.plan + z[12]);
  a.fq = static_cast<const T *>(L.fq);
  a.vals = static_cast<T *>(L.vals);
  a.fout = static_cast<T *>(L.fout);
  const int64_t rb = int64_t(sizeof(T));
  const int64_t extents[5] = {L.n_verts * 2 * rb, z[19], L.n_elems * tables.nq * rb, L.nnz * rb,
                              L.n_verts * rb};
  for (int64_t e : extents)
    if (e < 0 || e >= (int64_t(1) << 32))
      return fail(TFEM_ERR_INDEX_RANGE, "an array of %lld bytes does not fit the 32-bit offsets "
                  "of the tile kernel", (long long)e);
  a.coords_bytes = unsigned(extents[0]);
  a.plan_bytes = unsigned(extents[1]);
  a.fq_bytes = load ? unsigned(extents[2]) : 0u;
  a.vals_bytes = kmat ? unsigned(extents[3]) : 0u;
  a.fout_bytes = load ? unsigned(extents[4]) : 0u;
  a.off_rec = unsigned(z[13]);
  a.off_gid = unsigned(z[14]);
  a.off_loff = unsigned(z[15]);
  a.off_rund = unsigned(z[16]);
  a.off_runl = unsigned(z[17]);
  a.off_eid = unsigned(z[18]);
  a.n_tiles = int(z[0]);
  a.lds_acc = (int(z[8]) + 3) & ~3;
  a.lds_vert = int(z[6]) | 1;  // odd: lds_vert + 1 is even, every LDS array stays 8-byte aligned
  a.lds_own = load ? ((int(z[7]) + 3) & ~3) : 0;
  a.lds_run = (int(z[10]) + 2) & ~1;  // even: the int4 chunk table behind it stays aligned
  a.flags = L.flags < 0 ? 0 : L.flags;
  a.stamps = L.stamps;
  // W = sum_q w_q/2, M_ij = sum_q (w_q/2) l_i l_j, l_i(q) w_q/2: formed in T, quadrature order
  T w = T(0);
  for (int q = 0; q < tables.nq; ++q) w = w + T(tables.hw[q]);
  a.stiff_w = T(L.alpha) * w;
  int m = 0;
  for (int i = 0; i < 3; ++i)
    for (int j = i; j < 3; ++j, ++m) {
      T s = T(0);
      for (int q = 0; q < tables.nq; ++q)
        s = s + T(tables.hw[q]) * (T(tables.lam[q][i]) * T(tables.lam[q][j]));
      a.mass_w[m] = T(L.beta) * s;
    }
  for (int q = 0; q < tables.nq; ++q)
    for (int i = 0; i < 3; ++i) a.lamw[q][i] = T(tables.lam[q][i]) * T(tables.hw[q]);
  size_t lds = size_t(a.lds_acc + kTrash + a.lds_own) * sizeof(T) +
               size_t(2 * (a.lds_vert + 1)) * sizeof(T) +
               size_t((a.lds_vert + 1) + 3 * a.lds_own + 4 * a.lds_run) * sizeof(int);
  lds = (lds + 15) & ~size_t(15);  // the chunk table starts 16-byte aligned
  lds += size_t(2 * kChunks) * sizeof(int4);
  if (lds > 160 * 1024) return fail(TFEM_ERR_INVALID_ARGUMENT, "tile needs %zu B of LDS", lds);
  const int per = int((z[0] + 7) / 8);
  int per_cu = int((160 * 1024) / lds);
  // 4 resp. 6 waves per SIMD (register budget of the instantiation)
  const int cap = (load ? 4 : 6) * 256 / kTileBlock;
  per_cu = per_cu < 1 ? 1 : (per_cu > cap ? cap : per_cu);
  int blocks = (cu_count() * per_cu / 8) * 8;
  if (blocks > per * 8) blocks = per * 8;
  const dim3 grid{unsigned(blocks)}, block{unsigned(kTileBlock)};
  const bool dbg = L.flags >= 0;
  const bool mass = kmat && L.beta != 0.0;
  if (z[20] != 2 && z[20] != 3)
    return fail(TFEM_ERR_INVALID_ARGUMENT, "plan layout: %lld words per record", (long long)z[20]);
  if (z[20] == 2 && (kElemPerLane != 2 || z[6] > 1022))
    return fail(TFEM_ERR_INVALID_ARGUMENT, "8-byte records need <= 1022 vertices per tile");
  void *kernel = z[20] == 2 ? pick_form<T, true>(kmat, mass, tables.nq, load, dbg)
                            : pick_form<T, false>(kmat, mass, tables.nq, load, dbg);
  if (!kernel) return fail(TFEM_ERR_UNSUPPORTED, "Integration order not implemented");
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
    if (e != hipSuccess) return fail(TFEM_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  void *params[] = {&a};
  hipError_t e = hipLaunchKernel(kernel, grid, block, params, lds, L.stream);
  if (e != hipSuccess) return fail(TFEM_ERR_HIP, "tile kernel launch: %s", hipGetErrorString(e));
  return TFEM_OK;
}

static int check_plan_limits(int64_t n_tiles, int max_n_elem, int max_n_vert, int max_n_own,
                             int max_acc, int max_n_runs) {
  if (n_tiles < 0 || max_n_elem > kTileBlock * kElemPerLane ||
      max_n_vert > kTileBlock * kVertPerLane || max_n_own > kTileBlock * kRowPerLane ||
      max_acc > kTileBlock * kAccPerLane || max_n_own > max_n_vert || max_n_runs > max_n_own ||
      max_n_runs < 0)
    return fail(TFEM_ERR_INVALID_ARGUMENT, "tile plan exceeds the kernel's capacities");
  return TFEM_OK;
}

}  // namespace tfem

extern "C" {

int tfem_tile_capacity(int what) {
  using namespace tfem;
  switch (what) {
    case 0: return kTileBlock * kElemPerLane;
    case 1: return kTileBlock * kVertPerLane;
    case 2: return kTileBlock * kRowPerLane;
    case 3: return kTileBlock * kAccPerLane;
    default: return 0;
  }
}

int tfem_p1_assemble_tiles(const void *coords, int real_bytes, int64_t n_verts, int quad_order,
                           double alpha, double beta, const void *plan_device,
                           const int64_t *plan_layout_host, void *vals, int64_t nnz,
                           const void *fq, int64_t n_elems, void *fout, void *stream) {
  using namespace tfem;
  if (real_bytes != 4 && real_bytes != 8)
    return fail(TFEM_ERR_INVALID_ARGUMENT, "real_bytes must be 4 or 8");
  if (!plan_layout_host) return fail(TFEM_ERR_INVALID_ARGUMENT, "plan_layout_host is NULL");
  const int64_t *z = plan_layout_host;
  if (int st = check_plan_limits(z[0], int(z[5]), int(z[6]), int(z[7]), int(z[8]), int(z[10])))
    return st;
  TileLaunch L{coords, quad_order, alpha, beta, static_cast<const unsigned char *>(plan_device),
               plan_layout_host, n_verts, n_elems, nnz, vals, fq, fout,
               static_cast<hipStream_t>(stream)};
  return real_bytes == 8 ? launch_tiles<double>(L) : launch_tiles<float>(L);
}

// Ablation build (fp64) for tools/ablate_tiles.py.
int tfem_p1_tiles_debug(const void *coords, int64_t n_verts, int quad_order,
                        const void *plan_device, const int64_t *plan_layout_host, void *vals,
                        int64_t nnz, const void *fq, int64_t n_elems, void *fout, void *stream,
                        int flags, unsigned long long *stamps) {
  using namespace tfem;
  const int64_t *z = plan_layout_host;
  if (int st = check_plan_limits(z[0], int(z[5]), int(z[6]), int(z[7]), int(z[8]), int(z[10])))
    return st;
  TileLaunch L{coords, quad_order, 1.0, 0.0, static_cast<const unsigned char *>(plan_device),
               plan_layout_host, n_verts, n_elems, nnz, vals, fq, fout,
               static_cast<hipStream_t>(stream), flags & 0xFF, stamps};
  return launch_tiles<double>(L);
}

}  // extern "C"
