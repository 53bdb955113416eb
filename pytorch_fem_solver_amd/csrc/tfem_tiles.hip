// P1 bilinear assembly over a tile plan (tfem_tiles_host.cpp): the headline kernel.
//
// A tile owns a compact set of CSR rows (vertices).  Per tile:
//   - 12-byte element records (three tile-local vertex ids + 9 four-bit column positions)
//     are read coalesced, the tile's vertex coordinates are gathered into LDS once;
//   - per element: coordinates from LDS, Jacobian, signed det, inverse, gradients and the
//     3x3 block, then ds_add_f64 of the entries whose ROW this tile owns (LDS atomics only;
//     nothing global);
//   - the accumulators are streamed to the CSR value array lane-contiguously through the
//     tile's OUTPUT RUNS (groups of owned rows that are contiguous in the CSR array), each
//     value written exactly once with a plain, fully coalesced store.
//
// k_p1_tiles_pipe: persistent workgroups, two per CU, each walking a strided list of tiles
// inside its XCD's contiguous Z-order range.  Everything the NEXT tiles need is loaded into
// registers while the current tile is computed and parked in the second LDS buffer
// afterwards, so HBM loads, fp64 arithmetic and value stores of one CU overlap instead of
// running phase by phase.  Barriers wait for LDS only (s_waitcnt lgkmcnt(0); s_barrier) --
// a __syncthreads() would drain the prefetch loads as well.
//
// Arithmetic: the P1 gradients are constant on an element, so the quadrature sum
// sum_q (alpha g_i.g_j + beta l_i(q) l_j(q)) w_q det/2 (abstract_basis.py:83) is evaluated
// as  alpha (g_i.g_j) (det W) + (beta M_ij) det  with W = sum_q w_q/2 and
// M_ij = sum_q (w_q/2) l_i(q) l_j(q) formed once on the host in the same precision -- the
// same numbers up to a few units of rounding (checked against the oracle at 1e-12; the
// strict operation-order version is k_p1_bilinear_atomic in tfem_kernels.hip).
//
// HBM traffic per element ~ 13.6 B records + ~12.7 B coordinates/ids + ~1 B row/run info +
// 28 B values = ~55 B, against 48 B algorithmic.
#include <hip/hip_runtime.h>

#include <cstring>

#include "tfem_common.hpp"

// This kernel trades the reference's exact operation order for fewer fp64 instructions
// (see "Arithmetic" above); let the compiler fuse multiply-adds here.
#pragma clang fp contract(fast)

namespace tfem {

// 512 lanes (8 waves) share one tile: with two workgroups per CU that is 4 waves per SIMD,
// which is what hides the dependent fp64 / LDS latency chains of the element phase.
constexpr int kTileBlock = 512;
constexpr int kElemPerLane = 3;  // tile element capacity   = 1536
constexpr int kVertPerLane = 2;  // tile vertex capacity    = 1024
constexpr int kRowPerLane = 1;   // tile owned-row capacity = 512
constexpr int kAccPerLane = 8;   // tile accumulator capacity = 4096 entries
constexpr int kChunks = kTileBlock * kAccPerLane / 64;  // 64-entry output chunks per tile
constexpr int kElemCap = kTileBlock * kElemPerLane;
constexpr int kDescStride = 12;  // ints per tile descriptor (tfem_tiles_host.cpp)

template <typename T>
struct TileArgs {
  const T *coords;
  const int32_t *desc;
  const uint32_t *records;
  const int32_t *vert_gid;
  const uint16_t *row_loff;
  const int32_t *run_delta;
  const uint16_t *run_lstart;
  T *vals;
  int n_tiles;
  int lds_acc;   // accumulator entries reserved in LDS
  int lds_vert;  // vertex slots reserved in LDS (per buffer)
  int lds_own;   // owned-row slots reserved in LDS (per buffer)
  int lds_run;   // run slots reserved in LDS (per buffer), >= max runs + 1
  T stiff_w;     // alpha * sum_q w_q / 2
  T mass_w[6];   // beta * sum_q (w_q/2) l_i l_j for (i,j) = 00 01 02 11 12 22
  int flags;     // diagnostic build only (tfem_p1_bilinear_tiles_debug)
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

// 3x3 block of one element (basis.py:87-88, element_tri.py:132-145 and :41,
// abstract_basis.py:83) and its scatter into the LDS accumulators:
// local[i][j] -> A[row v_j][col v_i] (basis.py:73-76).
// Record word j = 16 * local id of vertex j | positions << 16.  `loff` maps EVERY local
// vertex to the byte offset of its row's accumulators: rows this tile does not own (halo
// vertices) map to a small trash area that is never written out, so the element phase has
// no branches.  Padding lanes hold a null record (a dummy vertex whose row is trash too).
template <typename T, bool MASS, bool DBG>
__device__ __forceinline__ void element_to_lds(const TileArgs<T> &a, const uint32_t (&rec)[3],
                                               const unsigned char *xy_bytes,
                                               const unsigned char *loff_bytes,
                                               unsigned char *acc_bytes) {
  T x[3], y[3];
  int base[3];
#pragma unroll
  for (int v = 0; v < 3; ++v) {
    const uint32_t off16 = rec[v] & 0xFFFFu;  // 16 * lid
    const T *p = reinterpret_cast<const T *>(xy_bytes + (sizeof(T) == 8 ? off16 : off16 >> 1));
    x[v] = p[0];
    y[v] = p[1];
    base[v] = *reinterpret_cast<const int *>(loff_bytes + (off16 >> 2));
  }
  const T ja = x[1] - x[0], jb = x[2] - x[0];
  const T jc = y[1] - y[0], jd = y[2] - y[0];
  const T det = ja * jd - jb * jc;  // signed (element_tri.py:139)
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
      const int pos = int((rec[j] >> (16 + 4 * i)) & 0xFu);
      T *slot = reinterpret_cast<T *>(acc_bytes + base[j] + pos * int(sizeof(T)));
      if (DBG && (a.flags & 1)) {
        if (loc[i][j] == T(-1.2345e300)) *slot = loc[i][j];  // keeps the math alive
      } else {
        atomicAdd(slot, loc[i][j]);  // ds_add_f64 / ds_add_f32
      }
    }
  }
}

// Accumulator entries behind lds_acc that absorb the rows a tile does not own; halo vertex l
// uses entries (l & 15) .. (l & 15) + 15 so that neighbouring lanes do not pile up on one
// address.
constexpr int kTrash = 32;

// ---------------------------------------------------------------------------------------
// Iteration k of a persistent workgroup (tile k current, all of its data already in LDS):
//   S2  element phase of tile k: records, coordinates, row offsets from LDS; fp64; ds_add
//   --  LDS barrier, then ONE s_waitcnt vmcnt(0): the loads of tile k+1 (issued a whole
//       element phase ago) and the stores of tile k-1 have landed long since
//   S5  stream tile k's accumulators out (wave per 64-entry chunk, lanes along the CSR
//       array: contiguous 512-byte stores), clearing each entry after reading it
//   S3  park tile k+1 from registers into LDS: records, coordinates, row offsets (single
//       buffers: S2 of tile k is over) and runs (double buffer: S5 of tile k may still read)
//   S4  issue the loads of tile k+2 into the registers just freed + vertex ids of tile k+3
//   --  LDS barrier; build the output chunk table of tile k+1
// The only vector-memory wait in the loop is the explicit vmcnt(0), one full element phase
// after the loads were issued: neither load latency nor store acknowledgements sit on the
// critical path.  Loads are unconditional with clamped indices (straight-line code); no
// loop follows the loads inside an iteration (hipcc drains vmcnt in front of loops).
// DBG = true is the ablation build used by tools/ablate_tiles.py: bits of a.flags switch
// off 1 = LDS atomics, 2 = the whole element phase, 4 = the value stores, 8 = the
// coordinate gather, 16 = in-kernel stamps.  Its results are wrong by design; the product
// path never uses it.
// ---------------------------------------------------------------------------------------
template <typename T, bool MASS, bool DBG>
__global__ __launch_bounds__(kTileBlock, 2) void k_p1_tiles_pipe(const TileArgs<T> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T *acc = reinterpret_cast<T *>(smem_raw);                             // [lds_acc + kTrash]
  T *xy = acc + a.lds_acc + kTrash;                                     // [2 * (lds_vert + 1)]
  int *loff = reinterpret_cast<int *>(xy + 2 * (a.lds_vert + 1));       // [lds_vert + 1]
  uint32_t *rec_lds = reinterpret_cast<uint32_t *>(loff + a.lds_vert + 1);  // [3][kElemCap]
  int *runl_buf = reinterpret_cast<int *>(rec_lds + 3 * kElemCap);      // [2][lds_run]
  int *rund_buf = runl_buf + 2 * a.lds_run;                             // [2][lds_run]
  int4 *ctab_buf = reinterpret_cast<int4 *>(rund_buf + 2 * a.lds_run);  // [2][kChunks]
  const int trash = a.lds_acc * int(sizeof(T));

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
  auto clamp_lane = [&](int slot, int count) {  // slot-th item of this lane, clamped
    const int l = tid + slot * kTileBlock;
    return l < count ? l : count - 1;
  };

  // registers that carry one tile from its loads (S4) to its parking (S3, next iteration)
  uint32_t rec[kElemPerLane][3];
  T xyr[kVertPerLane][2];
  int loffr[kRowPerLane], runlr[kRowPerLane], rundr[kRowPerLane];
  int gid[kVertPerLane];  // vertex ids, one tile further ahead

  auto load_tile = [&](const TileDesc &d, bool gather_by_gid) {
#pragma unroll
    for (int e = 0; e < kElemPerLane; ++e) {
      const int idx = clamp_lane(e, d.n_elem > 0 ? d.n_elem : 1);
      const uint32_t *r = a.records + 3 * size_t(d.elem_off + idx);
      rec[e][0] = r[0];
      rec[e][1] = r[1];
      rec[e][2] = r[2];
    }
#pragma unroll
    for (int r = 0; r < kRowPerLane; ++r) {
      loffr[r] = a.row_loff[d.loff_off + clamp_lane(r, d.n_own)];
      const int lr = clamp_lane(r, d.n_runs > 0 ? d.n_runs : 1);  // arrays are padded by one
      runlr[r] = a.run_lstart[d.lrun_off + lr];
      rundr[r] = a.run_delta[d.run_off + lr];
    }
    if (!(DBG && (a.flags & 8))) {
#pragma unroll
      for (int v = 0; v < kVertPerLane; ++v) {
        const int64_t g = gather_by_gid ? int64_t(gid[v])
                                        : int64_t(a.vert_gid[d.vert_off + clamp_lane(v, d.n_vert)]);
        xyr[v][0] = a.coords[2 * g];
        xyr[v][1] = a.coords[2 * g + 1];
      }
    }
  };
  auto load_gid = [&](const TileDesc &d) {
#pragma unroll
    for (int v = 0; v < kVertPerLane; ++v)
      gid[v] = a.vert_gid[d.vert_off + clamp_lane(v, d.n_vert)];
  };
  auto park_tile = [&](const TileDesc &d, int buf) {
#pragma unroll
    for (int e = 0; e < kElemPerLane; ++e) {
      const int idx = tid + e * kTileBlock;
      // lanes past the end of the tile get a null record: dummy vertex, trash rows
      const bool real = idx < d.n_elem;
      // null record: the dummy vertex three times; positions spread over the trash entries
      const uint32_t null_word = (uint32_t(a.lds_vert) << 4) | (uint32_t(lane & 15) << 16);
      rec_lds[idx] = real ? rec[e][0] : null_word;
      rec_lds[kElemCap + idx] = real ? rec[e][1] : null_word;
      rec_lds[2 * kElemCap + idx] = real ? rec[e][2] : null_word;
    }
#pragma unroll
    for (int v = 0; v < kVertPerLane; ++v) {
      const int l = tid + v * kTileBlock;
      if (l < d.n_vert) {
        xy[2 * l] = xyr[v][0];
        xy[2 * l + 1] = xyr[v][1];
        loff[l] = (v < kRowPerLane && l < d.n_own)
                      ? loffr[v < kRowPerLane ? v : 0] * int(sizeof(T))
                      : trash + (l & 15) * int(sizeof(T));
      }
    }
    int *runl_w = runl_buf + buf * a.lds_run;
    int *rund_w = rund_buf + buf * a.lds_run;
#pragma unroll
    for (int r = 0; r < kRowPerLane; ++r) {
      const int lr = clamp_lane(r, d.n_runs > 0 ? d.n_runs : 1);
      runl_w[lr] = runlr[r];
      rund_w[lr] = rundr[r];
    }
    if (tid == 0) runl_w[d.n_runs] = d.acc_size;  // sentinel: end of the last run
  };
  // Output chunk table (needs the parked runs: call after a barrier).  Chunk c = accumulator
  // entries [64c, 64c+64); x = end of the run holding entry 64c, y = delta of that run,
  // z = delta of the next run, w = index of the first run | slow << 16 (a third run starts
  // inside the chunk: lanes then search).  Fixed-trip binary search: no loop in the code.
  auto build_chunks = [&](const TileDesc &d, int buf) {
    const int n_chunks = (d.acc_size + 63) >> 6;
    if (tid < n_chunks) {
      const int *runl = runl_buf + buf * a.lds_run;
      const int *rund = rund_buf + buf * a.lds_run;
      const int target = tid << 6;
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
      const bool slow = lo + 2 < d.n_runs && runl[lo + 2] < target + 64;
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

  // ---- prologue: tile 0 into LDS, tile 1 into registers, vertex ids of tile 2 ----------------
  for (int s = tid; s < a.lds_acc + kTrash; s += kTileBlock) acc[s] = T(0);
  if (tid < 2) {  // the dummy vertex (slot lds_vert): finite coordinates, trash row
    xy[2 * a.lds_vert + tid] = T(tid);
    loff[a.lds_vert] = trash;
  }
  load_tile(dc, false);
  park_tile(dc, 0);
  if (t_n >= 0) load_tile(dn, false);
  if (t_nn >= 0) load_gid(dnn);
  __syncthreads();
  build_chunks(dc, 0);
  __syncthreads();

  int cur = 0;
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
        const int idx = tid + e * kTileBlock;
        if (e * kTileBlock + wave * 64 < dc.n_elem) {
          const uint32_t r3[3] = {rec_lds[idx], rec_lds[kElemCap + idx],
                                  rec_lds[2 * kElemCap + idx]};
          element_to_lds<T, MASS, DBG>(a, r3, reinterpret_cast<const unsigned char *>(xy),
                                       reinterpret_cast<const unsigned char *>(loff),
                                       reinterpret_cast<unsigned char *>(acc));
        }
      }
    }
    if (timing) t1 = stamp();
    lds_barrier();
    if (timing) t2 = stamp();
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): see the comment above the kernel
    if (timing) t3 = stamp();
    // ---- S5 ----------------------------------------------------------------------------------
    // wave w streams chunks w, w+8, ...: lane l owns accumulator entry 64c + l, whose run
    // (hence its place in the CSR array) comes from the chunk table; 4 chunks in flight
    {
      const int4 *ctab = ctab_buf + cur * kChunks;
      const int n_chunks = (dc.acc_size + 63) >> 6;
      for (int c0 = wave; c0 < n_chunks; c0 += 4 * (kTileBlock / 64)) {
        T val[4];
        int4 tab[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int c = c0 + (kTileBlock / 64) * u < n_chunks ? c0 + (kTileBlock / 64) * u : n_chunks - 1;
          const int sidx = (c << 6) + lane;
          tab[u] = ctab[c];
          val[u] = acc[sidx < dc.acc_size ? sidx : dc.acc_size - 1];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int c = c0 + (kTileBlock / 64) * u;
          const int sidx = (c << 6) + lane;
          // run of this lane's entry: the chunk's first run, or the next one ...
          int run = (tab[u].w & 0xFFFF) + (sidx < tab[u].x ? 0 : 1);
          if (tab[u].w >> 16) {  // ... or, if more than two runs meet in this chunk (wave-
            // uniform), one of the next 64: fixed-trip binary search, no loop (hipcc drains
            // vmcnt, i.e. the stores, in front of loops)
            int lo = tab[u].w & 0xFFFF;
            int hi = lo + 64 < dc.n_runs ? lo + 64 : dc.n_runs;
#pragma unroll
            for (int it = 0; it < 6; ++it) {
              const int mid = (lo + hi) >> 1;
              const int start = runl[mid];
              const bool up = hi - lo > 1 && start <= sidx;
              const bool down = hi - lo > 1 && start > sidx;
              lo = up ? mid : lo;
              hi = down ? mid : hi;
            }
            run = lo;
          }
          // always an LDS read: a select between an LDS value and a register value can turn
          // into a flat_load, which waits on vmcnt too, i.e. on every store in flight
          const int delta = rund[run < dc.n_runs ? run : 0];
          if (c < n_chunks && sidx < dc.acc_size) {
            if (!(DBG && (a.flags & 4))) a.vals[int64_t(sidx) + delta] = val[u];
            acc[sidx] = T(0);
          }
        }
      }
    }
    if (timing) t4 = stamp();
    // ---- S3 + S4 ---------------------------------------------------------------------------------
    if (t_n >= 0) park_tile(dn, cur ^ 1);
    if (t_nn >= 0) {
      load_tile(dnn, true);
      if (t_nnn >= 0) load_gid(dnnn);
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
  }
  if (DBG && (a.flags & 16) && a.stamps && (tid & 63) == 0) {
    unsigned long long *o = a.stamps + 8 * (size_t(blockIdx.x) * (kTileBlock / 64) + (tid >> 6));
    for (int i = 0; i < 8; ++i) o[i] = tsum[i];
  }
}

struct TileLaunch {
  const void *coords;
  int quad_order;
  double alpha, beta;
  const int32_t *desc;
  int64_t n_tiles;
  const uint32_t *records;
  const int32_t *vert_gid;
  const uint16_t *row_loff;
  const int32_t *run_delta;
  const uint16_t *run_lstart;
  int max_n_elem, max_n_vert, max_n_own, max_acc, max_n_runs;
  void *vals;
  hipStream_t stream;
  int flags;  // < 0: production build
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

template <typename T>
static int launch_tiles(const TileLaunch &L) {
  TriTables tables;
  if (!build_tri_tables(L.quad_order, int(sizeof(T)), &tables))
    return fail(TFEM_ERR_UNSUPPORTED, "Integration order not implemented");
  if (L.n_tiles == 0) return TFEM_OK;
  if (!L.coords || !L.desc || !L.records || !L.vert_gid || !L.row_loff || !L.run_delta ||
      !L.run_lstart || !L.vals)
    return fail(TFEM_ERR_INVALID_ARGUMENT, "NULL pointer");
  TileArgs<T> a;
  std::memset(&a, 0, sizeof(a));
  a.coords = static_cast<const T *>(L.coords);
  a.desc = L.desc;
  a.records = L.records;
  a.vert_gid = L.vert_gid;
  a.row_loff = L.row_loff;
  a.run_delta = L.run_delta;
  a.run_lstart = L.run_lstart;
  a.vals = static_cast<T *>(L.vals);
  a.n_tiles = int(L.n_tiles);
  a.lds_acc = (L.max_acc + 3) & ~3;
  a.lds_vert = L.max_n_vert | 1;  // odd: lds_vert + 1 is even, every LDS array stays 8-byte aligned
  a.lds_own = L.max_n_own;
  a.lds_run = (L.max_n_runs + 2) & ~1;  // even: the int4 chunk table behind it stays aligned
  a.flags = L.flags < 0 ? 0 : L.flags;
  a.stamps = L.stamps;
  // W = sum_q w_q/2 and M_ij = sum_q (w_q/2) l_i l_j, accumulated in T in quadrature order
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
  const size_t lds = size_t(a.lds_acc + kTrash) * sizeof(T) +
                     size_t(2 * (a.lds_vert + 1)) * sizeof(T) +
                     size_t((a.lds_vert + 1) + 3 * kElemCap + 4 * a.lds_run) * sizeof(int) + 16 +
                     size_t(2 * kChunks) * sizeof(int4);
  if (lds > 160 * 1024) return fail(TFEM_ERR_INVALID_ARGUMENT, "tile needs %zu B of LDS", lds);
  const int per = int((L.n_tiles + 7) / 8);
  int per_cu = int((160 * 1024) / lds);
  per_cu = per_cu < 1 ? 1 : (per_cu > 2 ? 2 : per_cu);
  int blocks = (cu_count() * per_cu / 8) * 8;
  if (blocks > per * 8) blocks = per * 8;
  const dim3 grid{unsigned(blocks)}, block{unsigned(kTileBlock)};
  const bool dbg = L.flags >= 0;
  const bool mass = L.beta != 0.0;
  auto kernel = mass ? (dbg ? k_p1_tiles_pipe<T, true, true> : k_p1_tiles_pipe<T, true, false>)
                     : (dbg ? k_p1_tiles_pipe<T, false, true> : k_p1_tiles_pipe<T, false, false>);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
    if (e != hipSuccess) return fail(TFEM_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  hipLaunchKernelGGL(kernel, grid, block, lds, L.stream, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(TFEM_ERR_HIP, "tile kernel launch: %s", hipGetErrorString(e));
  return TFEM_OK;
}

static int check_plan_limits(int64_t n_tiles, int max_n_elem, int max_n_vert, int max_n_own,
                             int max_acc, int max_n_runs) {
  if (n_tiles < 0 || max_n_elem > kTileBlock * kElemPerLane ||
      max_n_vert > kTileBlock * kVertPerLane || max_n_own > kTileBlock * kRowPerLane ||
      max_acc > kTileBlock * kAccPerLane || max_n_own > max_n_vert || max_n_runs > max_n_own || max_n_runs < 0)
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
    default: return 0;
  }
}

int tfem_p1_bilinear_tiles(const void *coords, int real_bytes, int quad_order, double alpha,
                           double beta, const int32_t *desc, int64_t n_tiles,
                           const uint32_t *records, const int32_t *vert_gid,
                           const uint16_t *row_loff, const int32_t *run_delta,
                           const uint16_t *run_lstart, int max_n_elem, int max_n_vert,
                           int max_n_own, int max_acc, int max_n_runs, void *vals, void *stream) {
  using namespace tfem;
  if (real_bytes != 4 && real_bytes != 8)
    return fail(TFEM_ERR_INVALID_ARGUMENT, "real_bytes must be 4 or 8");
  if (int st = check_plan_limits(n_tiles, max_n_elem, max_n_vert, max_n_own, max_acc, max_n_runs))
    return st;
  TileLaunch L{coords, quad_order, alpha, beta, desc, n_tiles, records, vert_gid, row_loff,
               run_delta, run_lstart, max_n_elem, max_n_vert, max_n_own, max_acc, max_n_runs,
               vals, static_cast<hipStream_t>(stream), -1};
  return real_bytes == 8 ? launch_tiles<double>(L) : launch_tiles<float>(L);
}

// Ablation build (fp64, stiffness) for tools/ablate_tiles.py.
int tfem_p1_bilinear_tiles_debug(const void *coords, int quad_order, const int32_t *desc,
                                 int64_t n_tiles, const uint32_t *records,
                                 const int32_t *vert_gid, const uint16_t *row_loff,
                                 const int32_t *run_delta, const uint16_t *run_lstart,
                                 int max_n_elem, int max_n_vert, int max_n_own, int max_acc,
                                 int max_n_runs, void *vals, void *stream, int flags,
                                 unsigned long long *stamps) {
  using namespace tfem;
  if (int st = check_plan_limits(n_tiles, max_n_elem, max_n_vert, max_n_own, max_acc, max_n_runs))
    return st;
  TileLaunch L{coords, quad_order, 1.0, 0.0, desc, n_tiles, records, vert_gid, row_loff,
               run_delta, run_lstart, max_n_elem, max_n_vert, max_n_own, max_acc, max_n_runs,
               vals, static_cast<hipStream_t>(stream), flags & 0xFF, stamps};
  return launch_tiles<double>(L);
}

}  // extern "C"
