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
// HBM traffic per element with consecutive-vertex tiles (N_v = N_T / 2; measured):
// row records 8 B + halo vertex ids ~1 B + coordinates ~9.5 B (halo re-reads included) +
// values 28.6 B = 47.6 B, against 48 B algorithmic: the 16-byte row records replace the
// 24 bytes of connectivity the row's triangles take.
#include <cmath>
#include <cstdio>
#include <mutex>
#include <vector>

#include "tfem_rings_kernel.hpp"

namespace tfem {

// Rows of vertices with 8 .. 15 neighbours in a plan with long rows (tfem_rings_host.cpp):
// SIXTEEN lanes per row, lane i = slot i of the fan, coordinates by global ids.  A lane
// evaluates its slot's triangle (ring_row's formulas), takes what the previous slot's triangle
// adds to its own column from the lane before it (slot 0 from slot k - 1), and writes its entry;
// the diagonal follows from the sum over the sixteen lanes.
template <typename T, bool MASS>
__global__ __launch_bounds__(kRingBlock) void k_p1_long_rows(const T *coords, const unsigned char *plan,
                                                             unsigned off_long, int n_long, T *vals, T stiff_w,
                                                             T mass_d, T mass_o) {
  const int gtid = int(blockIdx.x) * kRingBlock + int(threadIdx.x);
  const int row = gtid >> 4, i = gtid & 15;
  const bool live = row < n_long;
  const uint32_t *rec = reinterpret_cast<const uint32_t *>(plan + off_long) + 24 * size_t(live ? row : 0);
  const uint32_t v = rec[0];
  const int k = int(rec[2] & 0xFFu);
  const int dpos = int(rec[2] >> 8);
  const bool slot = live && i < k;
  const uint32_t flag = slot ? (rec[3] >> (2 * i)) & 3u : 0u;
  const int nxt = i + 1 == k ? 0 : i + 1;
  const uint32_t g0 = rec[4 + (slot ? i : 0)], g1 = rec[4 + (slot ? nxt : 0)];
  const T xv = coords[2 * size_t(v)], yv = coords[2 * size_t(v) + 1];
  const T ecx = coords[2 * size_t(g0)] - xv, ecy = coords[2 * size_t(g0) + 1] - yv;
  const T enx = coords[2 * size_t(g1)] - xv, eny = coords[2 * size_t(g1) + 1] - yv;
  const T qc = ecx * ecx + ecy * ecy, qn = enx * enx + eny * eny;
  const T p = ecx * enx + ecy * eny;
  const T cross = ecx * eny - ecy * enx;
  const T cs = flag_weight<T>(stiff_w, flag) * fast_rcp<T>(flag ? cross : T(1));
  T here = cs * (p - qn), next = cs * (p - qc);  // to column n_i, to column n_next
  T sdet = T(0);
  if (MASS) {
    sdet = flag_weight<T>(T(1), flag) * cross;
    here = here + mass_o * sdet;
    next = next + mass_o * sdet;
  }
  const int lane = int(threadIdx.x) & 63;
  const int from = (lane & ~15) + (i == 0 ? (k > 0 ? k - 1 : 0) : i - 1);
  const T entry = here + __shfl(next, from, 64);
  T sum = here + next, dsum = sdet;
#pragma unroll
  for (int m = 8; m >= 1; m >>= 1) {
    sum = sum + __shfl_xor(sum, m, 64);
    if (MASS) dsum = dsum + __shfl_xor(dsum, m, 64);
  }
  if (!slot) return;
  T *out = vals + rec[1];
  out[int((rec[19 + i / 8] >> (4 * (i % 8))) & 15u)] = entry;
  // stiffness rows sum to zero; the mass part is taken out of the sum and added on the diagonal
  if (i == 0) out[dpos] = MASS ? mass_d * dsum - (sum - T(2) * mass_o * dsum) : -sum;
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
  const void *fq = nullptr;  // nullptr: no load vector
  int64_t n_elems = 0;
  void *fout = nullptr;
  const tfem_source_program *source = nullptr;  // load vector of this program instead of fq
  int64_t tile_first = 0, tile_count = -1;       // tile range of the plan (-1: to the end)
  int blocks_per_cu = 0;  // > 0: cap on resident workgroups per CU (tuning)
  int flags = 0;          // > 0: ablation build (wrong results by design)
  unsigned long long *stamps = nullptr;
};

static int ring_cu_count() {
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

// load vector alone (vals == NULL): the matrix part of the row is dead code
template <typename T, int SLOTS, bool CHUNK>
static void *pick_ring_load_only(int nq) {
  switch (nq) {
    case 1: return reinterpret_cast<void *>(k_p1_rings<T, SLOTS, false, CHUNK, 1, false, false>);
    case 3: return reinterpret_cast<void *>(k_p1_rings<T, SLOTS, false, CHUNK, 3, false, false>);
    case 4: return reinterpret_cast<void *>(k_p1_rings<T, SLOTS, false, CHUNK, 4, false, false>);
    case 6: return reinterpret_cast<void *>(k_p1_rings<T, SLOTS, false, CHUNK, 6, false, false>);
    default: return nullptr;
  }
}

template <typename T, int SLOTS, bool MASS, bool CHUNK>
static void *pick_ring_q(int nq) {
  switch (nq) {
    case 0: return reinterpret_cast<void *>(k_p1_rings<T, SLOTS, MASS, CHUNK, 0, false>);
    case 1: return reinterpret_cast<void *>(k_p1_rings<T, SLOTS, MASS, CHUNK, 1, false>);
    case 3: return reinterpret_cast<void *>(k_p1_rings<T, SLOTS, MASS, CHUNK, 3, false>);
    case 4: return reinterpret_cast<void *>(k_p1_rings<T, SLOTS, MASS, CHUNK, 4, false>);
    case 6: return reinterpret_cast<void *>(k_p1_rings<T, SLOTS, MASS, CHUNK, 6, false>);
    default: return nullptr;
  }
}

template <typename T, int SLOTS, bool CHUNK>
static void *pick_ring_mass(bool kmat, bool mass, int nq) {
  if (!kmat) return pick_ring_load_only<T, SLOTS, CHUNK>(nq);
  return mass ? pick_ring_q<T, SLOTS, true, CHUNK>(nq) : pick_ring_q<T, SLOTS, false, CHUNK>(nq);
}

// nq = 0: matrix only; kmat = false: the load vector alone; src: source program in the launch
// (those instantiations live in tfem_rings_src.hip)
template <typename T>
static void *pick_ring_kernel(int slots, bool mass, bool chunk, int nq, bool src, bool kmat, bool wide) {
  if (src) return pick_ring_src_kernel<T>(slots, mass, chunk, nq, kmat, wide);
  if (slots == 7)
    return chunk ? pick_ring_mass<T, 7, true>(kmat, mass, nq) : pick_ring_mass<T, 7, false>(kmat, mass, nq);
  return chunk ? pick_ring_mass<T, 15, true>(kmat, mass, nq) : pick_ring_mass<T, 15, false>(kmat, mass, nq);
}

template <typename T>
static int launch_rings(const RingLaunch &L) {
  TriTables tables;
  if (!build_tri_tables(L.quad_order, int(sizeof(T)), &tables))
    return fail(TFEM_ERR_UNSUPPORTED, "Integration order not implemented");
  const int64_t *z = L.layout;
  if (z[0] == 0) return TFEM_OK;
  const bool src = L.source != nullptr;
  const bool load = L.fq != nullptr || src;
  const bool kmat = L.vals != nullptr;
  if (src && L.fq) return fail(TFEM_ERR_INVALID_ARGUMENT, "source values AND a source program");
  if (!kmat && !load) return fail(TFEM_ERR_INVALID_ARGUMENT, "nothing to assemble");
  if (!L.coords || !L.plan || (load && !L.fout)) return fail(TFEM_ERR_INVALID_ARGUMENT, "NULL pointer");
  if (z[0] < 0 || z[4] > kRingBlock || z[3] > kRingVertCap || z[4] > z[3] || z[14] > kRingHaloCap ||
      !((z[6] == 7 && z[7] == 4) || (z[6] == 15 && z[7] == 8)) || (z[5] > z[6] + 1 && z[23] == 0) || z[5] > 16)
    return fail(TFEM_ERR_INVALID_ARGUMENT, "ring plan exceeds the kernel's capacities");
  RingArgs<T> a;
  std::memset(&a, 0, sizeof(a));
  a.coords = static_cast<const T *>(L.coords);
  a.plan = L.plan;
  a.vals = static_cast<T *>(L.vals);
  const int64_t rb = int64_t(sizeof(T));
  const int64_t extents[5] = {L.n_verts * 2 * rb, z[12], kmat ? L.nnz * rb : 0,
                              (load && !src) ? L.n_elems * tables.nq * rb : 0, load ? L.n_verts * rb : 0};
  for (int64_t e : extents)
    if (e < 0 || e >= (int64_t(1) << 32))
      return fail(TFEM_ERR_INDEX_RANGE, "an array of %lld bytes does not fit the 32-bit offsets "
                  "of the ring kernel", (long long)e);
  a.coords_bytes = unsigned(extents[0]);
  a.plan_bytes = unsigned(extents[1]);
  a.vals_bytes = unsigned(extents[2]);
  a.fq = static_cast<const T *>(L.fq);
  a.fout = static_cast<T *>(L.fout);
  a.fq_bytes = unsigned(extents[3]);
  a.fout_bytes = unsigned(extents[4]);
  a.off_elems = unsigned(z[15]);
  a.off_telems = unsigned(z[16]);
  // measured at 1e7 elements: the matrix-only launch is 4-6 % faster with one contiguous range
  // per XCD (halo coordinates shared in that XCD's L2), the launches that read the source values
  // 6-7 % faster with the tile list dealt to the XCDs in blocks of 4 (one front of reads across
  // the chip; blocks of 1, 2, 8 within 1.5 %)
  a.xcd_interleave = load ? 4 : 0;
  if (const char *v = std::getenv("TFEM_RINGS_XCD"))  // developer switch: block size, 0 = ranges
    a.xcd_interleave = std::strcmp(v, "interleave") == 0 ? 1 : std::atoi(v);
  if (load && !src && z[23] > 0)
    return fail(TFEM_ERR_UNSUPPORTED, "a ring plan with long rows takes the load vector of a source program, "
                "not of pre-evaluated source values");
  if (load && !src && (z[18] == 0 || z[17] > kRingElemPerLane * kRingBlock))
    return fail(TFEM_ERR_UNSUPPORTED, "a tile of the ring plan has %lld elements: the fused load "
                "vector stages at most %d", (long long)z[17], kRingElemPerLane * kRingBlock);
  if (src && z[27] > kRingElemPerLane * kRingBlock)
    return fail(TFEM_ERR_UNSUPPORTED, "a tile of the ring plan evaluates %lld elements: the launch "
                "takes at most %d", (long long)z[27], kRingElemPerLane * kRingBlock);
  for (int i = 0; i < 3; ++i)
    for (int q = 0; q < tables.nq; ++q) {
      a.lamw[i][q] = T(tables.lam[q][i]) * T(tables.hw[q]);
      a.lam[i][q] = T(tables.lam[q][i]);
      a.hw[q] = T(tables.hw[q]);
    }
  if (src && tables.nq == 4) {
    // the structure of the 4-point rule the QL = 4 instantiations use (RingArgs::qsym)
    // (c0, a, b from the rule's literals; 1 - xi - eta of the reference's table is the same to an ulp)
    const double c0 = tables.lam[0][1], aa = tables.lam[1][1], b = tables.lam[1][2], d = aa - b;
    const int at[4] = {0, 1, 2, 0};
    const double tol = sizeof(T) == 8 ? 1e-15 : 1e-6;  // the tables are rounded to the launch's real type
    auto near = [tol](double x, double y) { return std::fabs(x - y) <= tol; };
    bool ok = tables.hw[1] == tables.hw[2] && tables.hw[1] == tables.hw[3];
    for (int i = 0; i < 3; ++i) {
      ok = ok && near(tables.lam[0][i], c0);
      for (int q = 1; q < 4; ++q) ok = ok && near(tables.lam[q][i], i == at[q] ? aa : b);
    }
    if (!ok) return fail(TFEM_ERR_UNSUPPORTED, "the 4-point rule is not the reference's order-3 rule");
    a.qsym[0] = T(c0);
    a.qsym[1] = T(b);
    a.qsym[2] = T(d);
    a.qsym[3] = T(c0) * T(tables.hw[0]);
    a.qsym[4] = T(b) * T(tables.hw[1]);
    a.qsym[5] = T(d) * T(tables.hw[1]);
  }
  if (src) {
    if (z[20] == 0 || z[21] == 0)
      return fail(TFEM_ERR_UNSUPPORTED, "the ring plan carries no element vertex table (source programs)");
    a.off_tverts = unsigned(z[20]);
    a.off_chain = unsigned(z[24]);
    a.off_hin = unsigned(z[26]);
    a.chain_len = int(z[25]);
    if (a.chain_len < 1 || z[24] == 0 || z[26] == 0)
      return fail(TFEM_ERR_INVALID_ARGUMENT, "the ring plan carries no chain order (layout of an older build?)");
    const int st = src_convert<T>(L.source, &a.src);
    if (st != TFEM_OK) return st;
  }
  a.off_desc = unsigned(z[8]);
  a.off_rows = unsigned(z[9]);
  a.off_rowstart = unsigned(z[10]);
  a.off_gid = unsigned(z[11]);
  const int64_t t_first = L.tile_first, t_count = L.tile_count < 0 ? z[0] - L.tile_first : L.tile_count;
  if (t_first < 0 || t_count < 0 || t_first + t_count > z[0])
    return fail(TFEM_ERR_INVALID_ARGUMENT, "tile range [%lld, +%lld) outside the plan's %lld tiles",
                (long long)t_first, (long long)t_count, (long long)z[0]);
  if (kmat && z[23] > 0 && (t_first != 0 || t_count != z[0]))
    return fail(TFEM_ERR_UNSUPPORTED, "a ring plan with long rows is launched over all of its tiles: the rows of "
                "the vertices with 8 .. 15 neighbours are written by one launch over all of them");
  if (t_count == 0) return TFEM_OK;
  a.n_tiles = int(t_count);
  // a tile's index only addresses its descriptor; the launches of a source program walk positions
  // [t_first, t_first + t_count) of the chain order instead (the same tiles for the ranges the
  // plan knows: the tiles owning flagged vertices come first in both orders)
  a.n_runs = 0;
  if (src) {
    a.u_first = int(t_first);
    if (z[28] > 0 && (t_first != 0 || t_count != z[0]))
      return fail(TFEM_ERR_UNSUPPORTED, "tile ranges need a plan built with flagged vertices (its blocks break between the ranges)");
    if (z[28] > 0) {  // runs (plans without flagged vertices): the whole plan in one launch
      a.n_runs = int(z[28]);
      a.off_runs = unsigned(z[30]);
    }
  } else {
    a.off_desc += 80u * unsigned(t_first);
  }
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
  a.lds_elem = load ? int(z[17]) : 0;
  const size_t lds = size_t(4 * a.lds_vert) * sizeof(T) +
                     (kmat ? size_t(kRingWaves) * size_t(64 * (slots + 1) + 2) * sizeof(T) : 0) +
                     (src ? size_t(3 * (a.lds_vert + 2)) * sizeof(T)  // three buffers of sums per local vertex
                          : load ? size_t(3 * a.lds_elem + 4) * sizeof(T) : 0);
  const bool chunk = z[13] != 0;
  a.flags = L.flags > 0 ? L.flags : 0;
  // Store policy of the CSR values (TFEM_RINGS_STORES=nt | plain overrides).  Measured at S(2236)
  // and S(3162) (profiles/r03_k_store_policy.log): the matrix-only launch from COLD caches (behind
  // 512 MB of unrelated reads) takes 81 us with plain stores against 99 us with non-temporal ones
  // (74 % against 60 % of the roofline; 68 % against 59 % at 2e7 elements), behind unrelated writes
  // both take 112-113 us; only launches of the SAME mesh in a row, whose read set then survives in
  // the memory-side cache, prefer non-temporal stores (81-93 us against 100).  The launches that
  // also form a load vector are bound by vector issue and run 3-20 % faster with non-temporal
  // stores.  So: plain for the matrix alone, non-temporal with a load vector.
  bool plain_stores = !load;
  if (const char *v = std::getenv("TFEM_RINGS_STORES")) plain_stores = std::strcmp(v, "plain") == 0;
  if (plain_stores) a.flags |= 1024;
  a.stamps = L.stamps;
  // programs that never hold more than two values: three elements per pass of the interpreter
  bool wide = src && src_depth(L.source) <= 2;
  if (const char *v = std::getenv("TFEM_SRC_WIDE")) wide = wide && std::atoi(v) != 0;  // developer switch
  void *kernel = pick_ring_kernel<T>(slots, mass, chunk, load ? tables.nq : 0, src, kmat, wide);
  if (!kernel) return fail(TFEM_ERR_UNSUPPORTED, "Integration order not implemented");
  if constexpr (sizeof(T) == 8) {  // the ablation build exists for fp64 stiffness, 7 slots
    if (kmat && !src && L.flags > 0 && slots == 7 && !mass && (!load || tables.nq == 4)) {
      if (load)
        kernel = chunk ? reinterpret_cast<void *>(k_p1_rings<T, 7, false, true, 4, true>)
                       : reinterpret_cast<void *>(k_p1_rings<T, 7, false, false, 4, true>);
      else
        kernel = chunk ? reinterpret_cast<void *>(k_p1_rings<T, 7, false, true, 0, true>)
                       : reinterpret_cast<void *>(k_p1_rings<T, 7, false, false, 0, true>);
    }
  }
  // resident workgroups: what LDS and registers allow per CU, on every CU.  The answer (and the
  // one-off attribute for more than 64 KB of LDS) is kept per (kernel, LDS size): the launch path
  // of a prepared step does no runtime query
  struct Occupancy { void *kernel; size_t lds; int per_cu; };
  static Occupancy occ_cache[16];
  static int occ_used = 0;
  static std::mutex occ_mutex;
  int per_cu = 0;
  {
    std::lock_guard<std::mutex> guard(occ_mutex);
    for (int i = 0; i < occ_used; ++i)
      if (occ_cache[i].kernel == kernel && occ_cache[i].lds == lds) per_cu = occ_cache[i].per_cu;
    if (per_cu == 0) {
      if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        if (e != hipSuccess) return fail(TFEM_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
      }
      hipError_t oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kRingBlock, lds);
      if (oe != hipSuccess || per_cu < 1) per_cu = 1;
      if (occ_used < 16) occ_cache[occ_used++] = {kernel, lds, per_cu};
    }
  }
  const int deal = a.xcd_interleave > 0 ? a.xcd_interleave : 1;
  int per = int((t_count + 8 * deal - 1) / (8 * deal)) * deal;
  if (src) {  // blocks of the chain order, dealt to the XCDs round-robin: workgroups per XCD that get one
    const int64_t first_block = t_first / a.chain_len, last_block = (t_first + t_count - 1) / a.chain_len;
    per = int((last_block - first_block + 1 + 7) / 8);
    if (a.n_runs > 0) per = (a.n_runs + 7) / 8;
  }
  if (L.blocks_per_cu > 0 && L.blocks_per_cu < per_cu) per_cu = L.blocks_per_cu;
  // TFEM_RINGS_RESERVE_CUS: CUs per XCD this launch leaves free (a sharded step: the kernels of
  // the interface exchange of the previous step -- pack, RCCL's all-reduce, unpack -- find room
  // beside the persistent workgroups of this one)
  int cus = ring_cu_count();
  if (const char *v = std::getenv("TFEM_RINGS_RESERVE_CUS")) cus = std::max(8, cus - 8 * std::max(0, std::atoi(v)));
  const int blocks = std::min(per * 8, (cus * per_cu / 8) * 8);
  const dim3 grid{unsigned(blocks)}, block{unsigned(kRingBlock)};
#ifdef TFEM_SRC_TIMING
  // developer build (tools/ablate_src.py): phase stamps of launch number 300 of a source program
  static unsigned long long *dev_stamps = nullptr;
  static int n_launch = 0;
  const bool stamp_now = src && kmat && ++n_launch == 300;
  if (stamp_now) {
    if (!dev_stamps) (void)hipMalloc(&dev_stamps, sizeof(unsigned long long) * 12 * kRingWaves * 8192);
    a.stamps = dev_stamps;
  }
#endif
  void *params[] = {&a};
  hipError_t e = hipLaunchKernel(kernel, grid, block, params, lds, L.stream);
  if (e != hipSuccess) return fail(TFEM_ERR_HIP, "ring kernel launch: %s", hipGetErrorString(e));
#ifdef TFEM_SRC_TIMING
  if (stamp_now) {
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(size_t(12) * kRingWaves * size_t(blocks));
    (void)hipMemcpy(h.data(), dev_stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double sum[10] = {0};
    double clk = 0, real = 0, real_max = 0, real_min = 1e300, tiles_max = 0;
    for (size_t w = 0; w < size_t(kRingWaves) * size_t(blocks); ++w) {
      for (int i = 0; i < 10; ++i) sum[i] += double(h[12 * w + size_t(i)]);
      clk += double(h[12 * w + 10]);
      real += double(h[12 * w + 11]);
      real_max = std::max(real_max, double(h[12 * w + 11]));
      real_min = std::min(real_min, double(h[12 * w + 11]));
      tiles_max = std::max(tiles_max, double(h[12 * w + 7]));
    }
    const char *names[10] = {"A loads", "B rows", "stage", "vmcnt", "park", "stores", "barrier E", "tiles", "G", "barrier G"};
    std::fprintf(stderr, "[stamps] %d workgroups, shader cycles per tile and wave (s_memtime):", blocks);
    for (int i = 0; i < 10; ++i)
      if (i != 7) std::fprintf(stderr, "  %s %.0f", names[i], sum[i] / sum[7]);
    std::fprintf(stderr, "  tiles/wave %.1f  |  in-kernel clock %.3f GHz (shader cycles / 100 MHz ticks over the waves' tile loops), loop %.1f us\n",
                 sum[7] / (double(kRingWaves) * blocks), clk / real * 0.1, real / (double(kRingWaves) * blocks) * 0.01);
    std::fprintf(stderr, "[stamps] loop of the shortest / longest wave %.1f / %.1f us, most tiles of a wave %.0f\n",
                 real_min * 0.01, real_max * 0.01, tiles_max);
    // who is fast: mean loop by the workgroup's place in the launch order (blocks of 256 = one per CU)
    // and by XCD
    std::fprintf(stderr, "[stamps] mean loop by blockIdx / 256:");
    for (int layer = 0; layer * 256 < blocks; ++layer) {
      double t = 0;
      int n = 0;
      for (int b = layer * 256; b < std::min(blocks, (layer + 1) * 256); ++b, ++n) t += double(h[12 * size_t(b) * kRingWaves + 11]);
      std::fprintf(stderr, " %.1f", t / n * 0.01);
    }
    std::fprintf(stderr, " us;  by XCD (blockIdx & 7):");
    for (int x = 0; x < 8; ++x) {
      double t = 0;
      int n = 0;
      for (int b = x; b < blocks; b += 8, ++n) t += double(h[12 * size_t(b) * kRingWaves + 11]);
      std::fprintf(stderr, " %.1f", t / n * 0.01);
    }
    std::fprintf(stderr, " us\n");
  }
#endif
  if (kmat && z[23] > 0) {  // the rows of the vertices with 8 .. 15 neighbours
    const dim3 lgrid{unsigned((16 * z[23] + kRingBlock - 1) / kRingBlock)};  // sixteen lanes per row
    if (mass)
      hipLaunchKernelGGL((k_p1_long_rows<T, true>), lgrid, block, 0, L.stream, a.coords, a.plan, unsigned(z[22]),
                         int(z[23]), a.vals, a.stiff_w, a.mass_d, a.mass_o);
    else
      hipLaunchKernelGGL((k_p1_long_rows<T, false>), lgrid, block, 0, L.stream, a.coords, a.plan, unsigned(z[22]),
                         int(z[23]), a.vals, a.stiff_w, a.mass_d, a.mass_o);
    e = hipGetLastError();
    if (e != hipSuccess) return fail(TFEM_ERR_HIP, "long-row kernel launch: %s", hipGetErrorString(e));
  }
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
                           const void *fq, int64_t n_elems, void *fout, void *stream) {
  using namespace tfem;
  if (real_bytes != 4 && real_bytes != 8)
    return fail(TFEM_ERR_INVALID_ARGUMENT, "real_bytes must be 4 or 8");
  if (!plan_layout_host) return fail(TFEM_ERR_INVALID_ARGUMENT, "plan_layout_host is NULL");
  RingLaunch L{coords, quad_order, alpha, beta, static_cast<const unsigned char *>(plan_device),
               plan_layout_host, n_verts, nnz, vals, static_cast<hipStream_t>(stream)};
  L.fq = fq;
  L.n_elems = n_elems;
  L.fout = fout;
  // developer switches (tools/time_rings.py)
  if (const char *v = std::getenv("TFEM_RINGS_PER_CU")) L.blocks_per_cu = std::atoi(v);
  if (const char *v = std::getenv("TFEM_RINGS_DEBUG")) L.flags = std::atoi(v);  // ablation build
  return real_bytes == 8 ? launch_rings<double>(L) : launch_rings<float>(L);
}

int tfem_p1_assemble_rings_source(const void *coords, int real_bytes, int64_t n_verts, int quad_order,
                                  double alpha, double beta, const void *plan_device,
                                  const int64_t *plan_layout_host, void *vals, int64_t nnz,
                                  const tfem_source_program *source, int64_t n_elems, void *fout,
                                  void *stream) {
  using namespace tfem;
  if (real_bytes != 4 && real_bytes != 8)
    return fail(TFEM_ERR_INVALID_ARGUMENT, "real_bytes must be 4 or 8");
  if (!plan_layout_host || !source) return fail(TFEM_ERR_INVALID_ARGUMENT, "NULL pointer");
  RingLaunch L{coords, quad_order, alpha, beta, static_cast<const unsigned char *>(plan_device),
               plan_layout_host, n_verts, nnz, vals, static_cast<hipStream_t>(stream)};
  L.source = source;
  L.n_elems = n_elems;
  L.fout = fout;
  if (const char *v = std::getenv("TFEM_RINGS_PER_CU")) L.blocks_per_cu = std::atoi(v);
  return real_bytes == 8 ? launch_rings<double>(L) : launch_rings<float>(L);
}

int tfem_p1_assemble_rings_range(const void *coords, int real_bytes, int64_t n_verts, int quad_order,
                                 double alpha, double beta, const void *plan_device,
                                 const int64_t *plan_layout_host, void *vals, int64_t nnz, const void *fq,
                                 const tfem_source_program *source, int64_t n_elems, void *fout,
                                 int64_t tile_first, int64_t tile_count, void *stream) {
  using namespace tfem;
  if (real_bytes != 4 && real_bytes != 8)
    return fail(TFEM_ERR_INVALID_ARGUMENT, "real_bytes must be 4 or 8");
  if (!plan_layout_host) return fail(TFEM_ERR_INVALID_ARGUMENT, "plan_layout_host is NULL");
  RingLaunch L{coords, quad_order, alpha, beta, static_cast<const unsigned char *>(plan_device),
               plan_layout_host, n_verts, nnz, vals, static_cast<hipStream_t>(stream)};
  L.fq = fq;
  L.source = source;
  L.n_elems = n_elems;
  L.fout = fout;
  L.tile_first = tile_first;
  L.tile_count = tile_count;
  if (const char *v = std::getenv("TFEM_RINGS_PER_CU")) L.blocks_per_cu = std::atoi(v);
  return real_bytes == 8 ? launch_rings<double>(L) : launch_rings<float>(L);
}

// Ablation build (fp64 stiffness, 7 slots) for tools/time_rings.py; `stamps` = 8 * 4 * grid
// 64-bit words or NULL.
int tfem_p1_rings_debug(const void *coords, int64_t n_verts, int quad_order, const void *plan_device,
                        const int64_t *plan_layout_host, void *vals, int64_t nnz, void *stream,
                        int flags, int blocks_per_cu, unsigned long long *stamps, const void *fq,
                        int64_t n_elems, void *fout) {
  using namespace tfem;
  RingLaunch L{coords, quad_order, 1.0, 0.0, static_cast<const unsigned char *>(plan_device),
               plan_layout_host, n_verts, nnz, vals, static_cast<hipStream_t>(stream)};
  L.flags = flags;
  L.blocks_per_cu = blocks_per_cu;
  L.stamps = stamps;
  L.fq = fq;
  L.n_elems = n_elems;
  L.fout = fout;
  return launch_rings<double>(L);
}

}  // extern "C"
