// Ring plan for the P1 stiffness/mass headline kernel (host, once per mesh).
//
// Owner-computes, row-centric form of the element loop (abstract_basis.py:74-93 with the
// scatter of basis.py:64-85): the CSR rows (= vertices) are cut into spatially compact tiles
// along a Z-order curve; one lane owns one row.  A row's record lists the row's neighbours
// IN FAN ORDER (the order in which the incident triangles chain around the vertex) as
// tile-local vertex ids; slot i also says whether a triangle (v, n_i, n_{i+1}) exists and
// with which orientation it is stored in the connectivity (the reference integrates with the
// SIGNED determinant, element_tri.py:139).  The lane evaluates every incident triangle and
// keeps its row's entries in registers: no atomics of any kind, every CSR value written once.
//
// Row record, SLOTS = 7 (rows of <= 8 entries; 4 dwords) / SLOTS = 15 (<= 16 entries; 8 dwords):
//   local id of neighbour i : 10 bits, dword i / 3, shift 10 * (i % 3)
//   k = number of neighbours: SLOTS 7 : (w0 >> 30) | ((w1 >> 30) & 1) << 2
//                             SLOTS 15: (w0 >> 30) | (w1 >> 30) << 2
//   position of the diagonal inside the row
//                           : SLOTS 7 : (w2 >> 24) & 7;  SLOTS 15: (w2 >> 30) | (w3 >> 30) << 2
//   triangle flag of slot i (0 none, 1 = connectivity holds (v, n_i, n_next) up to rotation,
//                            2 = it holds (v, n_next, n_i); next = i + 1, or 0 from slot k-1)
//                           : SLOTS 7 : (w2 >> (10 + 2 i)) & 3;  SLOTS 15: (w5 >> 2 i) & 3
//   position of column n_i inside the row
//                           : SLOTS 7 : (w3 >> 3 i) & 7
//                             SLOTS 15: i < 8 ? (w6 >> 4 i) & 15 : (w7 >> 4 (i - 8)) & 15
// LONG ROWS (plans of unstructured meshes, TFEM_RING_LONG=1; not the default, see build_rings):
// the tiles keep the 4-dword records; a vertex with 8 .. 15 neighbours keeps its place
// in its tile as a row of length zero (k = 0, w3 = 0x80000000 | entries of the row: the rows
// behind it in the wave start that many entries later) and is listed in `long_rows`, 24 dwords
// each, written by a launch of its own (one lane per long row, global ids):
//   vertex id, CSR offset of the row, k | position of the diagonal << 8, triangle flags (2 bits per
//   slot), 15 neighbour vertex ids, positions of the 15 neighbour columns (4 bits each, 2 dwords)
// Supported fans: one closed cycle (interior vertex) or any number of open chains (boundary
// vertex, several fans meeting in a vertex).  An edge with three or more triangles, a
// duplicated or degenerate triangle, or a closed cycle beside another fan is reported as
// TFEM_ERR_UNSUPPORTED; the caller then uses the element-record tile plan instead.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <numeric>
#include <vector>
#include <chrono>
#include <cstdio>

#include "tfem_common.hpp"
#include "tfem_threads.hpp"

namespace tfem {

constexpr int kRingDescStride = 20;
constexpr int kRingLayoutLen = 32;
constexpr int kRingElemCap = 768;  // elements staged per tile: three per lane of the kernel
constexpr int kRingElemRuns = 8;   // runs of consecutive element ids a tile may store instead of its list
constexpr int kRingHaloCapHost = 256;  // halo vertices per tile: one per lane of the kernel

struct RingPlan {
  int slots = 7, words = 4;
  // per tile (20 ints): vert_off, n_vert, row_off, then start_0 = 0, start_1, start_2, start_3,
  // start_4 = n_own (wave w of the workgroup owns the tile's rows [start_w, start_{w+1})),
  // global id of the first row of wave 0..3, CSR offset of the first row of wave 0..3,
  // offset into tile_elems, number of elements of the tile, element-list mode, offset into
  // tile_tverts
  std::vector<int32_t> desc;
  std::vector<uint32_t> rows;     // `words` dwords per owned row
  std::vector<int32_t> rowstart;  // rowptr[g] of every owned row
  // load vector only.  tile_elems: the elements of every tile's fans (ascending per tile: their
  // source values are fetched once per tile, coalesced, and staged in LDS); row_ecodes: per
  // owned row one 12-bit code per slot, packed: tile-local element index | local index
  // of the row's vertex in that element << 10 (0xFFF: no triangle); 12 bits per slot, packed.
  std::vector<int32_t> tile_elems;
  std::vector<uint32_t> row_ecodes;
  // per tile element (order of the tile's ascending element list): the three tile-local vertex
  // ids of the element in its own local order, 10 bits each -- what the kernel needs to form the
  // integration points of the element from the coordinates it holds in LDS (source programs)
  std::vector<uint32_t> tile_tverts;
  std::vector<uint32_t> long_rows;  // 24 dwords per vertex with 8 .. 15 neighbours (slots == 7 plans)
  int32_t max_n_elem = 0;
  bool elems_staged = true;  // false: some tile has more than kRingElemCap elements
  std::vector<int32_t> vert_gid;  // global id of every tile-local vertex, owned rows first
  int32_t max_n_vert = 0, max_n_own = 0, max_row_len = 0, max_n_halo = 0;
  int64_t n_tiles = 0;
  bool chunked = false;           // every wave's 64 rows are 64 consecutive vertices
  bool long_mode = false;         // 4-dword records + long rows instead of 8-dword records
  int64_t n_priority = 0;         // leading tiles that own a flagged vertex (tfem_ring_plan_create_priority)
  // Source-program launches (element form of the load vector) walk the tiles in CHAIN ORDER: the
  // tiles along the space-filling curve (spatial neighbours one after the other; the tiles owning
  // flagged vertices first), cut into blocks of `chain_len` positions that one workgroup takes in
  // a row.  Inside a block an element shared by consecutive tiles is evaluated ONCE, by the earlier
  // tile, which sums the shares of ALL its local vertices; a row of the later tile adds what the
  // earlier tile summed for its vertex (`hand_in`: per owned row the local id its vertex has in the
  // previous tile of the block, 0xFFFF: none) to its own sum; tile_tverts lists per tile only the
  // elements the tile evaluates itself.
  std::vector<int32_t> chain_order;  // position in the chain order -> tile
  std::vector<uint16_t> hand_in;     // parallel to rowstart
  int32_t chain_len = 1;
  // RUNS (plans without flagged vertices): the chain order is cut into chain_wgs runs, one per
  // resident workgroup of a launch (`runs`: first position of every run, + n_tiles); a run is one
  // block: hand-over from every tile to the next.  The runs' lengths follow `shares`: the vector
  // pipe serves a SIMD's OLDEST wave first, so the workgroups placed first on their CUs -- the first
  // quarter of the launch order -- progress fastest (equal shares: their loops end after 105, 127,
  // 154 and 181 us, profiles/r03_wave_loop_spread.log) and take more tiles, so that all finish
  // together.  No runs (plans with flagged vertices: the two launches of a sharded step): blocks of
  // chain_len positions throughout.
  std::vector<int32_t> runs;
  int32_t chain_wgs = 1024;
  int32_t max_n_tv = 0;              // most elements a tile evaluates itself
};

namespace {

inline uint64_t ring_spread_bits(uint64_t x) {
  x &= 0xFFFFFFFFull;
  x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
  x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
  x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
  x = (x | (x << 2)) & 0x3333333333333333ull;
  x = (x | (x << 1)) & 0x5555555555555555ull;
  return x;
}

struct Fan {
  int k = 0;
  int32_t nb[16];   // neighbours in fan order (global ids)
  int flag[16];     // triangle flag of every slot
  int32_t elem[16]; // element of every slot's triangle (-1: none)
  int loc[16];      // local index of the fan's vertex inside that element
};

// Fan of vertex v from its incident elements.  Returns false when the fan has no ring form.
template <typename I>
bool build_fan(const I *conn, int32_t v, const int32_t *adj_first, const int32_t *adj_last,
               Fan &fan) {
  const int nt = int(adj_last - adj_first);
  fan.k = 0;
  if (nt == 0) return true;
  if (nt > 16) return false;
  int32_t ta[16], tb[16];   // triangle t = (v, ta, tb) in connectivity order (rotated)
  int tj[16];               // local index of v in triangle t
  int32_t nb[17];
  int cnt[17], tri[17][2];
  int n_nb = 0;
  for (int t = 0; t < nt; ++t) {
    const I *c = conn + 3 * int64_t(adj_first[t]);
    int j = -1, hits = 0;
    for (int a = 0; a < 3; ++a)
      if (int32_t(c[a]) == v) {
        j = a;
        ++hits;
      }
    if (hits != 1) return false;  // degenerate element
    tj[t] = j;
    ta[t] = int32_t(c[(j + 1) % 3]);
    tb[t] = int32_t(c[(j + 2) % 3]);
    if (ta[t] == tb[t]) return false;
    for (int side = 0; side < 2; ++side) {
      const int32_t w = side ? tb[t] : ta[t];
      int i = 0;
      while (i < n_nb && nb[i] != w) ++i;
      if (i == n_nb) {
        if (n_nb == 16) return false;
        nb[n_nb] = w;
        cnt[n_nb] = 0;
        ++n_nb;
      }
      if (cnt[i] == 2) return false;  // an edge with three triangles
      tri[i][cnt[i]++] = t;
    }
  }
  // a triangle listed twice shows up as two triangles over the same neighbour pair
  for (int t = 0; t < nt; ++t)
    for (int u = t + 1; u < nt; ++u)
      if ((ta[t] == ta[u] && tb[t] == tb[u]) || (ta[t] == tb[u] && tb[t] == ta[u])) return false;
  bool used[16] = {false};
  bool seen[17] = {false};
  auto index_of = [&](int32_t w) {
    int i = 0;
    while (nb[i] != w) ++i;
    return i;
  };
  // walks from neighbour `c`; returns the neighbour index it stops at
  auto walk = [&](int c) {
    for (;;) {
      seen[c] = true;
      fan.nb[fan.k] = nb[c];
      fan.flag[fan.k] = 0;
      fan.elem[fan.k] = -1;
      fan.loc[fan.k] = 0;
      int t = -1;
      for (int s = 0; s < cnt[c]; ++s)
        if (!used[tri[c][s]]) {
          t = tri[c][s];
          break;
        }
      if (t < 0) {
        ++fan.k;
        return c;  // chain end: no triangle behind this slot
      }
      used[t] = true;
      const bool forward = ta[t] == nb[c];
      fan.flag[fan.k] = forward ? 1 : 2;
      fan.elem[fan.k] = adj_first[t];
      fan.loc[fan.k] = tj[t];
      ++fan.k;
      const int o = index_of(forward ? tb[t] : ta[t]);
      if (seen[o]) return o;  // closed the cycle: slot k-1 links to the start
      c = o;
    }
  };
  bool any_end = false;
  for (int i = 0; i < n_nb; ++i) any_end = any_end || cnt[i] == 1;
  if (!any_end) {  // one closed cycle, started at the smallest neighbour for determinism
    int start = 0;
    for (int i = 1; i < n_nb; ++i)
      if (nb[i] < nb[start]) start = i;
    const int stop = walk(start);
    if (stop != start || fan.k != n_nb) return false;  // several cycles
    for (int t = 0; t < nt; ++t)
      if (!used[t]) return false;
    return true;
  }
  for (;;) {  // open chains, each started at its smaller end
    int start = -1;
    for (int i = 0; i < n_nb; ++i)
      if (!seen[i] && cnt[i] == 1 && (start < 0 || nb[i] < nb[start])) start = i;
    if (start < 0) break;
    const int stop = walk(start);
    if (fan.flag[fan.k - 1] != 0) return false;  // ran into a visited vertex: not a chain
    (void)stop;
  }
  if (fan.k != n_nb) return false;  // a closed cycle beside the chains
  for (int t = 0; t < nt; ++t)
    if (!used[t]) return false;
  return true;
}

// Small open-addressing map global id -> tile-local number (one per emitting thread; cleared by
// bumping a generation counter).
struct TileMap {
  std::vector<int32_t> key, val;
  std::vector<uint32_t> gen;
  uint32_t now = 0;
  uint32_t mask;
  explicit TileMap(uint32_t capacity) : key(capacity), val(capacity), gen(capacity, 0), mask(capacity - 1) {}
  void clear() {
    if (++now == 0) {
      std::fill(gen.begin(), gen.end(), 0u);
      now = 1;
    }
  }
  static uint32_t hash(int32_t k) { return uint32_t(k) * 2654435761u; }
  // returns the slot of `k` (present or free)
  uint32_t slot(int32_t k) const {
    uint32_t s = (hash(k) >> 7) & mask;
    while (gen[s] == now && key[s] != k) s = (s + 1) & mask;
    return s;
  }
  bool has(int32_t k) const { return gen[slot(k)] == now; }
  int32_t get(int32_t k) const { return val[slot(k)]; }
  void put(int32_t k, int32_t v) {
    const uint32_t s = slot(k);
    gen[s] = now;
    key[s] = k;
    val[s] = v;
  }
};

// What one tile owns: its rows in order, and which wave of the workgroup takes which of them.
struct TileSpec {
  std::vector<int32_t> owned;
  int32_t wave_start[5] = {0, 0, 0, 0, 0};
  int32_t curve_rank = 0;  // position of the tile along the space-filling curve
};

// What the tiles of one thread add to the plan arrays (a thread takes a contiguous range of
// tiles, so the plan arrays are the threads' arenas one after the other), and per tile where its
// share starts inside the arena.
struct TileArena {
  std::vector<int32_t> vert_gid, rowstart, tile_elems;
  std::vector<uint32_t> rows, row_ecodes, tile_tverts, long_rows;
};

struct TileOut {
  int64_t off_vert = 0, off_row = 0, off_elem = 0, off_tv = 0;  // arena-local
  int32_t n_vert = 0, n_own = 0, n_elem = 0, elem_mode = 0, gid0[4] = {0, 0, 0, 0}, rs0[4] = {0, 0, 0, 0};
  int32_t wave_start[5] = {0, 0, 0, 0, 0};
  int status = TFEM_OK;
  int32_t bad_vertex = -1, bad_len = 0, bad_k = 0;
};

struct EmitScratch {
  TileMap verts{4096}, elems{8192};
  std::vector<int32_t> fan_elem, fan_loc, elems_here;
  Fan fan;
};

// Numbers the tile's vertices (owned rows first, in the order given, then the halo in order of
// first reference) and writes its row records, element list, element vertex table and slot codes.
template <typename I>
void emit_tile(const I *conn, const int64_t *adj_ptr, const int32_t *adj, const int64_t *rowptr,
               const int32_t *colind, int slots, int words, bool elem_ranges, bool long_mode,
               const TileSpec &spec, EmitScratch &sc, TileArena &ar, TileOut &out) {
  const std::vector<int32_t> &owned = spec.owned;
  const int n_own = int(owned.size());
  out.n_own = n_own;
  for (int w = 0; w < 5; ++w) out.wave_start[w] = spec.wave_start[w];
  out.off_vert = int64_t(ar.vert_gid.size());
  out.off_row = int64_t(ar.rowstart.size());
  out.off_elem = int64_t(ar.tile_elems.size());
  out.off_tv = int64_t(ar.tile_tverts.size());
  sc.verts.clear();
  for (int l = 0; l < n_own; ++l) sc.verts.put(owned[size_t(l)], l);
  ar.vert_gid.insert(ar.vert_gid.end(), owned.begin(), owned.end());
  int next_local = n_own;
  for (int l = 0; l < n_own; ++l) {
    const int32_t u = owned[size_t(l)];
    for (int64_t p = rowptr[u]; p < rowptr[u + 1]; ++p) {
      const int32_t w = colind[p];
      if (!sc.verts.has(w)) {
        sc.verts.put(w, next_local++);
        ar.vert_gid.push_back(w);
      }
    }
  }
  out.n_vert = next_local;
  sc.fan_elem.assign(size_t(n_own) * size_t(slots), -1);
  sc.fan_loc.assign(size_t(n_own) * size_t(slots), 0);
  sc.elems_here.clear();
  sc.elems.clear();
  Fan &fan = sc.fan;
  for (int l = 0; l < n_own; ++l) {
    const int32_t u = owned[size_t(l)];
    const int len = int(rowptr[u + 1] - rowptr[u]);
    if (!build_fan(conn, u, adj + adj_ptr[size_t(u)], adj + adj_ptr[size_t(u) + 1], fan)) {
      out.status = TFEM_ERR_UNSUPPORTED;
      out.bad_vertex = u;
      out.bad_len = -1;
      return;
    }
    if (len != (fan.k ? fan.k + 1 : 0) || (fan.k > slots && !(long_mode && fan.k <= 15))) {
      out.status = TFEM_ERR_UNSUPPORTED;
      out.bad_vertex = u;
      out.bad_len = len;
      out.bad_k = fan.k;
      return;
    }
    uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int32_t *first = colind + rowptr[u];
    const int32_t *last = colind + rowptr[u + 1];
    const uint32_t k = uint32_t(fan.k);
    const uint32_t dpos = len ? uint32_t(std::lower_bound(first, last, u) - first) : 0u;
    if (fan.k > slots) {
      // a long row: a record of length zero in the tile, the row itself in the long-row list
      w[3] = 0x80000000u | uint32_t(len);
      uint32_t r[24] = {0};
      r[0] = uint32_t(u);
      r[1] = uint32_t(rowptr[u]);
      r[2] = k | dpos << 8;
      for (int i = 0; i < fan.k; ++i) {
        r[3] |= uint32_t(fan.flag[i]) << (2 * i);
        r[4 + i] = uint32_t(fan.nb[i]);
        const uint32_t pos = uint32_t(std::lower_bound(first, last, fan.nb[i]) - first);
        r[19 + i / 8] |= pos << (4 * (i % 8));
      }
      ar.long_rows.insert(ar.long_rows.end(), r, r + 24);
      ar.rows.insert(ar.rows.end(), w, w + words);
      ar.rowstart.push_back(int32_t(rowptr[u]));
      for (int i = 0; i < fan.k; ++i)  // its elements still belong to the tile (load vector)
        if (fan.flag[i] != 0 && !sc.elems.has(fan.elem[i])) {
          sc.elems.put(fan.elem[i], 0);
          sc.elems_here.push_back(fan.elem[i]);
        }
      continue;
    }
    for (int i = 0; i < fan.k; ++i) {
      const uint32_t lid = uint32_t(sc.verts.get(fan.nb[i]));
      const uint32_t pos = uint32_t(std::lower_bound(first, last, fan.nb[i]) - first);
      const uint32_t flag = uint32_t(fan.flag[i]);
      w[i / 3] |= lid << (10 * (i % 3));
      if (slots == 7) {
        w[2] |= flag << (10 + 2 * i);
        w[3] |= pos << (3 * i);
      } else {
        w[5] |= flag << (2 * i);
        if (i < 8)
          w[6] |= pos << (4 * i);
        else
          w[7] |= pos << (4 * (i - 8));
      }
    }
    if (slots == 7) {
      w[0] |= (k & 3u) << 30;
      w[1] |= (k >> 2) << 30;
      w[2] |= dpos << 24;
    } else {
      w[0] |= (k & 3u) << 30;
      w[1] |= (k >> 2) << 30;
      w[2] |= (dpos & 3u) << 30;
      w[3] |= (dpos >> 2) << 30;
    }
    ar.rows.insert(ar.rows.end(), w, w + words);
    ar.rowstart.push_back(int32_t(rowptr[u]));
    for (int i = 0; i < fan.k; ++i)
      if (fan.flag[i] != 0) {
        sc.fan_elem[size_t(l) * size_t(slots) + size_t(i)] = fan.elem[i];
        sc.fan_loc[size_t(l) * size_t(slots) + size_t(i)] = fan.loc[i];
        if (!sc.elems.has(fan.elem[i])) {
          sc.elems.put(fan.elem[i], 0);
          sc.elems_here.push_back(fan.elem[i]);
        }
      }
  }
  // the tile's elements, ascending (coalesced source-value loads), and the slot codes
  std::vector<int32_t> &elems_here = sc.elems_here;
  std::sort(elems_here.begin(), elems_here.end());
  const int32_t n_elem = int32_t(elems_here.size());
  out.n_elem = n_elem;
  for (int32_t j = 0; j < n_elem; ++j) sc.elems.put(elems_here[size_t(j)], j);
  for (int32_t j = 0; j < n_elem; ++j) {
    const I *c = conn + 3 * int64_t(elems_here[size_t(j)]);
    ar.tile_tverts.push_back(uint32_t(sc.verts.get(int32_t(c[0]))) | uint32_t(sc.verts.get(int32_t(c[1]))) << 10 |
                              uint32_t(sc.verts.get(int32_t(c[2]))) << 20);
  }
  // Element numberings with locality: the ascending list is a few runs of consecutive ids.
  // Up to kRingElemRuns of them are stored as 16 ints -- first id of every run, then the
  // number of elements up to and including every run (n_elem for the unused ones) -- and the
  // kernel derives the ids (desc[18] = 1); otherwise the list itself (desc[18] = 0).
  if (elem_ranges && n_elem > 2 * kRingElemRuns) {
    int32_t start[kRingElemRuns], upto[kRingElemRuns];
    int runs = 0;
    for (int32_t j = 0; j < n_elem; ++j) {
      if (j == 0 || elems_here[size_t(j)] != elems_here[size_t(j) - 1] + 1) {
        if (++runs > kRingElemRuns) break;
        start[runs - 1] = elems_here[size_t(j)];
      }
      upto[runs - 1] = j + 1;
    }
    if (runs <= kRingElemRuns) {
      for (int r = runs; r < kRingElemRuns; ++r) {
        start[r] = 0;
        upto[r] = n_elem;
      }
      ar.tile_elems.insert(ar.tile_elems.end(), start, start + kRingElemRuns);
      ar.tile_elems.insert(ar.tile_elems.end(), upto, upto + kRingElemRuns);
      out.elem_mode = 1;
    }
  }
  if (!out.elem_mode) ar.tile_elems.insert(ar.tile_elems.end(), elems_here.begin(), elems_here.end());
  const int ewords = (12 * slots + 31) / 32;  // 12-bit codes, packed: 3 (7 slots) or 6 dwords
  for (int l = 0; l < n_own; ++l) {
    uint32_t ew[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < slots; ++i) {
      uint32_t code = 0xFFFu;
      if (sc.fan_elem[size_t(l) * size_t(slots) + size_t(i)] >= 0) {
        const int32_t le = sc.elems.get(sc.fan_elem[size_t(l) * size_t(slots) + size_t(i)]);
        code = le < 1023 ? uint32_t(le) | uint32_t(sc.fan_loc[size_t(l) * size_t(slots) + size_t(i)]) << 10 : 0xFFFu;
      }
      const int bit = 12 * i;
      ew[bit / 32] |= code << (bit % 32);
      if (bit % 32 > 20) ew[bit / 32 + 1] |= code >> (32 - bit % 32);
    }
    ar.row_ecodes.insert(ar.row_ecodes.end(), ew, ew + ewords);
  }
  for (int w = 0; w < 4; ++w)  // first vertex and first CSR entry of every wave's rows
    if (spec.wave_start[w] < spec.wave_start[w + 1]) {
      out.gid0[w] = owned[size_t(spec.wave_start[w])];
      out.rs0[w] = int32_t(rowptr[owned[size_t(spec.wave_start[w])]]);
    }
}

// All tiles, in parallel; then the plan arrays in tile order.  Returns the status of the first
// tile (in tile order) that failed.
template <typename I>
int emit_plan(const I *conn, const int64_t *adj_ptr, const int32_t *adj, const int64_t *rowptr,
              const int32_t *colind, bool elem_ranges, const std::vector<TileSpec> &specs, RingPlan &plan) {
  const int64_t n_tiles = int64_t(specs.size());
  std::vector<TileOut> outs(static_cast<size_t>(n_tiles));
  const int max_threads = host_threads();
  std::vector<TileArena> arenas(static_cast<size_t>(max_threads));
  std::vector<int64_t> first_tile(size_t(max_threads) + 1, n_tiles);  // tile range of every thread
  const int ewords = (12 * plan.slots + 31) / 32;
  int64_t total_rows = 0;
  for (const TileSpec &sp : specs) total_rows += int64_t(sp.owned.size());
  parallel_for(n_tiles, [&](int64_t b, int64_t e, int t) {
    first_tile[size_t(t)] = b;
    TileArena &ar = arenas[size_t(t)];
    const size_t rows_here = size_t(total_rows * (e - b) / std::max<int64_t>(n_tiles, 1)) + 1024;
    ar.vert_gid.reserve(rows_here * 3 / 2);
    ar.rowstart.reserve(rows_here);
    ar.rows.reserve(rows_here * size_t(plan.words));
    ar.row_ecodes.reserve(rows_here * size_t(ewords));
    ar.tile_elems.reserve(rows_here / 4);
    ar.tile_tverts.reserve(rows_here * 3);
    EmitScratch sc;
    for (int64_t k = b; k < e; ++k)
      emit_tile(conn, adj_ptr, adj, rowptr, colind, plan.slots, plan.words, elem_ranges, plan.long_mode,
                specs[size_t(k)], sc, ar, outs[size_t(k)]);
  }, 8);
  for (const TileOut &o : outs)
    if (o.status != TFEM_OK) {
      if (o.bad_len < 0)
        return fail(TFEM_ERR_UNSUPPORTED, "the triangles around vertex %d do not form fans", o.bad_vertex);
      return fail(TFEM_ERR_UNSUPPORTED, "row %d: %d entries for %d neighbours", o.bad_vertex, o.bad_len, o.bad_k);
    }
  // where every thread's arena starts inside the plan arrays
  std::vector<int64_t> base_vert(size_t(max_threads) + 1, 0), base_row(size_t(max_threads) + 1, 0),
      base_elem(size_t(max_threads) + 1, 0), base_tv(size_t(max_threads) + 1, 0);
  for (int t = 0; t < max_threads; ++t) {
    base_vert[size_t(t) + 1] = base_vert[size_t(t)] + int64_t(arenas[size_t(t)].vert_gid.size());
    base_row[size_t(t) + 1] = base_row[size_t(t)] + int64_t(arenas[size_t(t)].rowstart.size());
    base_elem[size_t(t) + 1] = base_elem[size_t(t)] + int64_t(arenas[size_t(t)].tile_elems.size());
    base_tv[size_t(t) + 1] = base_tv[size_t(t)] + int64_t(arenas[size_t(t)].tile_tverts.size());
  }
  for (const TileOut &o : outs) {
    plan.max_n_elem = std::max(plan.max_n_elem, o.n_elem);
    if (o.n_elem > kRingElemCap) plan.elems_staged = false;
    plan.max_n_halo = std::max(plan.max_n_halo, o.n_vert - o.n_own);
    plan.max_n_vert = std::max(plan.max_n_vert, o.n_vert);
    plan.max_n_own = std::max(plan.max_n_own, o.n_own);
  }
  if (base_vert[size_t(max_threads)] >= (int64_t(1) << 31) || base_elem[size_t(max_threads)] >= (int64_t(1) << 31) ||
      base_tv[size_t(max_threads)] >= (int64_t(1) << 31))
    return fail(TFEM_ERR_INDEX_RANGE, "ring plan too large for its int32 offsets");
  plan.desc.assign(size_t(n_tiles) * kRingDescStride, 0);
  plan.vert_gid.resize(size_t(base_vert[size_t(max_threads)]));
  plan.rowstart.resize(size_t(base_row[size_t(max_threads)]));
  plan.rows.resize(size_t(base_row[size_t(max_threads)]) * size_t(plan.words));
  plan.row_ecodes.resize(size_t(base_row[size_t(max_threads)]) * size_t(ewords));
  plan.tile_elems.resize(size_t(base_elem[size_t(max_threads)]));
  plan.tile_tverts.resize(size_t(base_tv[size_t(max_threads)]));
  plan.long_rows.clear();
  for (int t = 0; t < max_threads; ++t)
    plan.long_rows.insert(plan.long_rows.end(), arenas[size_t(t)].long_rows.begin(), arenas[size_t(t)].long_rows.end());
  {
    std::vector<std::thread> pool;
    for (int t = 0; t < max_threads; ++t)
      pool.emplace_back([&, t]() {
        const TileArena &ar = arenas[size_t(t)];
        std::copy(ar.vert_gid.begin(), ar.vert_gid.end(), plan.vert_gid.begin() + base_vert[size_t(t)]);
        std::copy(ar.rowstart.begin(), ar.rowstart.end(), plan.rowstart.begin() + base_row[size_t(t)]);
        std::copy(ar.rows.begin(), ar.rows.end(), plan.rows.begin() + base_row[size_t(t)] * plan.words);
        std::copy(ar.row_ecodes.begin(), ar.row_ecodes.end(), plan.row_ecodes.begin() + base_row[size_t(t)] * ewords);
        std::copy(ar.tile_elems.begin(), ar.tile_elems.end(), plan.tile_elems.begin() + base_elem[size_t(t)]);
        std::copy(ar.tile_tverts.begin(), ar.tile_tverts.end(), plan.tile_tverts.begin() + base_tv[size_t(t)]);
        // the thread's tiles: [first_tile[t], first tile of the next thread that got any)
        int64_t stop = n_tiles;
        for (int u = t + 1; u < max_threads; ++u)
          if (first_tile[size_t(u)] < n_tiles) {
            stop = first_tile[size_t(u)];
            break;
          }
        for (int64_t k = first_tile[size_t(t)]; k < stop; ++k) {
          const TileOut &o = outs[size_t(k)];
          int32_t *d = plan.desc.data() + size_t(k) * kRingDescStride;
          d[0] = int32_t(base_vert[size_t(t)] + o.off_vert);
          d[1] = o.n_vert;
          d[2] = int32_t(base_row[size_t(t)] + o.off_row);
          d[3] = 0;
          d[4] = o.wave_start[1];
          d[5] = o.wave_start[2];
          d[6] = o.wave_start[3];
          d[7] = o.n_own;
          for (int w = 0; w < 4; ++w) {
            d[8 + w] = o.gid0[w];
            d[12 + w] = o.rs0[w];
          }
          d[16] = int32_t(base_elem[size_t(t)] + o.off_elem);
          d[17] = o.n_elem;
          d[18] = o.elem_mode;
          d[19] = int32_t(base_tv[size_t(t)] + o.off_tv);
        }
      });
    for (std::thread &th : pool) th.join();
  }
  plan.n_tiles = n_tiles;
  return TFEM_OK;
}

// Blocks of the chain order one workgroup takes in a row.  Measured at S(2236) = 20,147 tiles on
// 1024 resident workgroups (profiles/r03_chain_sweep.log): the fused K + f launch of sin * sin takes
// 190 us with blocks of 1, 184 with 4, 172 with 8 and 16 (the doubly evaluated elements drop from
// 27 % to 9 / 6 / 5 %), 239 with 32 (too few blocks per workgroup).  Blocks of 8.
int pick_chain_len(int64_t n_tiles) {
  if (const char *v = std::getenv("TFEM_RING_CHAIN")) return std::max(1, std::min(std::atoi(v), 64));
  (void)n_tiles;
  return 8;
}

// Shares of the four quarters of the launch order (workgroups 0 .. W/4 - 1 are placed first, one
// per CU, and are the oldest waves of their SIMDs).  Measured at S(2236) with the priorities taking
// turns (profiles/r03_wave_loop_spread.log): equal shares 170.6 us (loops of the quarters end after
// 137 / 148 / 156 / 165 us), 1.2 : 1.05 : 0.92 : 0.83 gives 166.7 us (159 / 153 / 147 / 147),
// 1.3 : 1.07 : 0.88 : 0.75 gives 169.6 (172 / 157 / 143 / 137: over-corrected).
// TFEM_RING_SHARES="a,b,c,d" overrides.
void chain_shares(double (&share)[4]) {
  const double measured[4] = {1.2, 1.05, 0.92, 0.83};
  for (int i = 0; i < 4; ++i) share[i] = measured[i];
  if (const char *v = std::getenv("TFEM_RING_SHARES")) {
    double a, b, c, d;
    if (std::sscanf(v, "%lf,%lf,%lf,%lf", &a, &b, &c, &d) == 4 && a > 0 && b > 0 && c > 0 && d > 0) {
      share[0] = a;
      share[1] = b;
      share[2] = c;
      share[3] = d;
    }
  }
}

// first position of every run (+ n_tiles).  Whole tiles per run, so the lengths are the integers
// that MINIMISE the longest run measured in its own workgroup's time: a workgroup of quarter q needs
// 1 / share[q] per tile, the launch lasts max_w len_w / share_w.  T = the smallest such maximum with
// sum_w floor(T share_w) >= n_tiles; what is over is taken from the runs closest to T.  For long runs
// this is the proportional split; for the short runs of small meshes and of the shards of a strong
// split (5 tiles per workgroup at 2.5e6 elements) it avoids the one tile more that a rounded share
// gives some of the oldest workgroups: S(1118) 61.2 -> 56 us (profiles/r03_fused_launch_sizes.log).
std::vector<int32_t> chain_runs(int64_t n_tiles, int64_t wgs) {
  double share[4];
  chain_shares(share);
  auto quarter = [&](int64_t w) { return int(std::min<int64_t>(3, 4 * w / wgs)); };
  int64_t members[4] = {0, 0, 0, 0};
  for (int64_t w = 0; w < wgs; ++w) members[quarter(w)]++;
  // candidates for T: k / share[q]
  std::vector<double> cand;
  double sum_share = 0;
  for (int q = 0; q < 4; ++q) sum_share += share[q] * double(members[q]);
  const int64_t kmax = int64_t(double(n_tiles) / std::max(sum_share, 1e-300) * std::max({share[0], share[1], share[2], share[3]})) + 3;
  for (int q = 0; q < 4; ++q)
    for (int64_t k = 1; k <= kmax; ++k) cand.push_back(double(k) / share[q]);
  std::sort(cand.begin(), cand.end());
  int64_t len_q[4] = {0, 0, 0, 0};
  for (double t : cand) {
    int64_t total = 0;
    for (int q = 0; q < 4; ++q) total += members[q] * (len_q[q] = int64_t(std::floor(t * share[q] * (1.0 + 1e-12))));
    if (total >= n_tiles) break;
  }
  std::vector<int64_t> len(static_cast<size_t>(wgs));
  int64_t total = 0;
  for (int64_t w = 0; w < wgs; ++w) total += (len[size_t(w)] = len_q[quarter(w)]);
  // what is over: one tile less for runs of the quarter whose runs take longest, spread over the
  // quarter (every other, every third ... run), then the next quarter
  while (total > n_tiles) {
    int worst = -1;
    double worst_t = -1;
    for (int q = 0; q < 4; ++q)
      if (members[q] > 0 && len_q[q] > 0 && double(len_q[q]) / share[q] > worst_t) {
        worst_t = double(len_q[q]) / share[q];
        worst = q;
      }
    if (worst < 0) break;
    const int64_t take = std::min<int64_t>(total - n_tiles, members[worst]);
    int64_t first_w = 0;
    while (quarter(first_w) != worst) ++first_w;
    for (int64_t i = 0; i < take; ++i) {
      const int64_t w = first_w + (i * members[worst]) / take;  // spread over the quarter
      len[size_t(w)]--;
    }
    total -= take;
    len_q[worst]--;
  }
  std::vector<int32_t> first(size_t(wgs) + 1, 0);
  for (int64_t w = 0; w < wgs; ++w) first[size_t(w) + 1] = first[size_t(w)] + int32_t(len[size_t(w)]);
  first[size_t(wgs)] = int32_t(n_tiles);
  return first;
}

// first position of every block of hand-overs
std::vector<uint8_t> chain_block_starts(int64_t n_tiles, int len, const std::vector<int32_t> &runs, int64_t n_priority) {
  std::vector<uint8_t> starts(size_t(n_tiles), 0);
  if (runs.empty()) {
    for (int64_t u = 0; u < n_tiles; u += len) starts[size_t(u)] = 1;
    if (n_priority > 0 && n_priority < n_tiles) starts[size_t(n_priority)] = 1;  // two launches: no hand-over between them
    return starts;
  }
  for (size_t w = 0; w + 1 < runs.size(); ++w)
    if (runs[w] < runs[w + 1]) starts[size_t(runs[w])] = 1;
  return starts;
}

// Chain order, per-tile tables of the elements a tile evaluates itself, carry maps (RingPlan).
// Works on the assembled plan: a tile's full element list (tile_elems / its runs) and vertex table
// (tile_tverts, same order) are filtered, vert_gid tells which halo vertices the next tile owns.
void chain_pass(const std::vector<TileSpec> &specs, RingPlan &plan) {
  const int64_t n_tiles = plan.n_tiles;
  plan.chain_order.resize(size_t(n_tiles));
  std::iota(plan.chain_order.begin(), plan.chain_order.end(), 0);
  auto by_curve = [&](int32_t x, int32_t y) { return specs[size_t(x)].curve_rank < specs[size_t(y)].curve_rank; };
  std::sort(plan.chain_order.begin(), plan.chain_order.begin() + plan.n_priority, by_curve);
  std::sort(plan.chain_order.begin() + plan.n_priority, plan.chain_order.end(), by_curve);
  const int len = pick_chain_len(n_tiles);
  plan.chain_len = len;
  plan.chain_wgs = 1024;  // resident workgroups of a source-program launch: 4 per CU of an MI355X
  if (const char *v = std::getenv("TFEM_RING_WGS")) plan.chain_wgs = std::max(8, std::atoi(v));
  plan.runs.clear();
  bool use_runs = plan.n_priority == 0;
  if (const char *v = std::getenv("TFEM_RING_RUNS")) use_runs = use_runs && std::atoi(v) != 0;
  if (use_runs) plan.runs = chain_runs(n_tiles, plan.chain_wgs);
  const std::vector<uint8_t> starts = chain_block_starts(n_tiles, len, plan.runs, plan.n_priority);
  plan.hand_in.assign(plan.rowstart.size(), uint16_t(0xFFFF));
  // the blocks as ranges of positions
  std::vector<int64_t> block_first;
  for (int64_t u = 0; u < n_tiles; ++u)
    if (starts[size_t(u)]) block_first.push_back(u);
  block_first.push_back(n_tiles);
  const int64_t n_blocks = int64_t(block_first.size()) - 1;
  std::vector<std::vector<uint32_t>> kept(static_cast<size_t>(n_tiles));
  int32_t *desc = plan.desc.data();
  auto elem_id = [&](const int32_t *d, int32_t j) {  // j-th element of the tile's ascending list
    const int32_t *te = plan.tile_elems.data() + d[16];
    if (!d[18]) return te[j];
    int32_t id = te[0] + j;
    for (int r = 1; r < kRingElemRuns; ++r)
      if (j >= te[kRingElemRuns + r - 1]) id = te[r] + (j - te[kRingElemRuns + r - 1]);
    return id;
  };
  parallel_for(n_blocks, [&](int64_t b0, int64_t b1, int) {
    TileMap prev(16384), cur(16384), prev_local(2048);
    for (int64_t b = b0; b < b1; ++b) {
      bool have_prev = false;
      for (int64_t u = block_first[size_t(b)]; u < block_first[size_t(b) + 1]; ++u) {
        const int32_t t = plan.chain_order[size_t(u)];
        const int32_t *d = desc + size_t(t) * kRingDescStride;
        const int32_t n_elem = d[17];
        const uint32_t *codes = plan.tile_tverts.data() + d[19];
        std::vector<uint32_t> &mine = kept[size_t(t)];
        mine.reserve(size_t(n_elem));
        cur.clear();
        for (int32_t j = 0; j < n_elem; ++j) {
          const int32_t e = elem_id(d, j);
          if (have_prev && prev.has(e)) continue;  // the tile before evaluates it and hands the sums over
          cur.put(e, 0);
          mine.push_back(codes[j]);
        }
        if (have_prev) {  // where the tile before summed the shares of this tile's vertices
          const int32_t *dp = desc + size_t(plan.chain_order[size_t(u) - 1]) * kRingDescStride;
          prev_local.clear();
          for (int32_t l = dp[7]; l < dp[1]; ++l) prev_local.put(plan.vert_gid[size_t(dp[0]) + size_t(l)], l);
          for (int32_t r = 0; r < d[7]; ++r) {
            const int32_t g = plan.vert_gid[size_t(d[0]) + size_t(r)];
            if (prev_local.has(g)) plan.hand_in[size_t(d[2]) + size_t(r)] = uint16_t(prev_local.get(g));
          }
        }
        std::swap(prev, cur);
        have_prev = true;
      }
    }
  }, 16);
  // the tables, in tile order
  int64_t total = 0;
  plan.max_n_tv = 0;
  for (int64_t t = 0; t < n_tiles; ++t) {
    int32_t *d = desc + size_t(t) * kRingDescStride;
    const int32_t n_tv = int32_t(kept[size_t(t)].size());
    d[18] = (d[18] & 0xFF) | (n_tv << 8);
    d[19] = int32_t(total);
    total += n_tv;
    plan.max_n_tv = std::max(plan.max_n_tv, n_tv);
  }
  std::vector<uint32_t> tv(static_cast<size_t>(total));
  parallel_for(n_tiles, [&](int64_t b, int64_t e, int) {
    for (int64_t t = b; t < e; ++t)
      std::copy(kept[size_t(t)].begin(), kept[size_t(t)].end(), tv.begin() + desc[size_t(t) * kRingDescStride + 19]);
  }, 256);
  plan.tile_tverts.swap(tv);
}

template <typename I>
int build_rings(const I *conn, int64_t n_elems, int64_t n_verts, const double *coords,
                const int64_t *rowptr, const int32_t *colind, int own_cap, int vert_cap,
                bool chunk_mode, const uint8_t *priority, const int64_t *inc_ptr, int32_t *inc,
                RingPlan &plan) {
  // tiles that own a flagged vertex first, the order inside both groups unchanged (multi-GPU:
  // the rows shared with other ranks are then complete after the first launch over a tile range)
  auto priority_first = [&](std::vector<TileSpec> &tiles) {
    plan.n_priority = 0;
    if (!priority) return;
    auto flagged = [&](const TileSpec &t) {
      for (int32_t v : t.owned)
        if (priority[size_t(v)]) return true;
      return false;
    };
    const auto middle = std::stable_partition(tiles.begin(), tiles.end(), flagged);
    plan.n_priority = int64_t(middle - tiles.begin());
  };
  const bool timing = std::getenv("TFEM_PLAN_TIMING") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (!timing) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[ring plan] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };
  // TFEM_RING_ELEM_RANGES=0: every tile stores its element list (A/B of the range encoding)
  const char *ranges_env = std::getenv("TFEM_RING_ELEM_RANGES");
  const bool elem_ranges = !(ranges_env && ranges_env[0] == '0');
  int64_t longest = 0;
  for (int64_t v = 0; v < n_verts; ++v) longest = std::max(longest, rowptr[v + 1] - rowptr[v]);
  if (longest > 16)
    return fail(TFEM_ERR_UNSUPPORTED, "a row has %lld entries (> 16)", (long long)longest);
  plan.max_row_len = int32_t(longest);
  // Measured (profiles/r02_delaunay_long_rows.log, Delaunay mesh of 1e6 / 5e6 points): the tile
  // kernel gets 20 % faster with 4-dword records (21.8 against 27.1 us at 2e6 elements), but the
  // long-row launch -- 13 % of the rows, 96-byte records, coordinates by global ids, 72-byte
  // pieces of the value array -- costs more than that gains (31 us): 8-dword records stay the
  // default; TFEM_RING_LONG=1 builds the plan with long rows.
  const char *long_env = std::getenv("TFEM_RING_LONG");
  plan.long_mode = longest > 8 && long_env && long_env[0] == '1';
  plan.slots = (longest <= 8 || plan.long_mode) ? 7 : 15;
  plan.words = plan.slots == 7 ? 4 : 8;
  double lo[2] = {1e300, 1e300}, hi[2] = {-1e300, -1e300};
  for (int64_t v = 0; v < n_verts; ++v)
    for (int c = 0; c < 2; ++c) {
      lo[c] = std::min(lo[c], coords[2 * v + c]);
      hi[c] = std::max(hi[c], coords[2 * v + c]);
    }
  const double span = std::max(std::max(hi[0] - lo[0], hi[1] - lo[1]), 1e-300);
  const double scale = double(1u << 24) / span;
  lap("row lengths, bounding box");
  // ---- vertex -> incident elements (ascending per vertex) ----------------------------------
  // bucket sizes and filling with relaxed atomic counters, then every vertex's short list sorted:
  // the result does not depend on the thread count
  // (inc_ptr, inc): the incidence a CSR pattern handle already holds (lists in arrival order)
  std::vector<int64_t> adj_own;
  std::unique_ptr<int32_t[]> adj_store;
  const int64_t *adj_ptr_data = inc_ptr;
  int32_t *adj_data = inc;
  if (!inc_ptr) {
    adj_own.assign(size_t(n_verts) + 1, 0);
    {
      std::vector<int64_t> count(size_t(n_verts) + 1, 0);
      parallel_for(3 * n_elems, [&](int64_t b, int64_t e, int) {
        for (int64_t k = b; k < e; ++k) __atomic_fetch_add(&count[size_t(conn[k]) + 1], int64_t(1), __ATOMIC_RELAXED);
      }, 1 << 16);
      std::partial_sum(count.begin(), count.end(), adj_own.begin());
    }
    adj_store.reset(new int32_t[size_t(std::max<int64_t>(3 * n_elems, 1))]);
    adj_data = adj_store.get();
    adj_ptr_data = adj_own.data();
    std::vector<int64_t> cur(adj_own.begin(), adj_own.end() - 1);
    parallel_for(n_elems, [&](int64_t b, int64_t e, int) {
      for (int64_t el = b; el < e; ++el)
        for (int a = 0; a < 3; ++a)
          adj_data[size_t(__atomic_fetch_add(&cur[size_t(conn[3 * el + a])], int64_t(1), __ATOMIC_RELAXED))] = int32_t(el);
    }, 1 << 14);
  }
  parallel_for(n_verts, [&](int64_t b, int64_t e, int) {
    for (int64_t v = b; v < e; ++v) std::sort(adj_data + adj_ptr_data[size_t(v)], adj_data + adj_ptr_data[size_t(v) + 1]);
  }, 1 << 14);
  lap("vertex -> elements");
  auto reset_plan = [&]() {
    plan.desc.clear();
    plan.rows.clear();
    plan.rowstart.clear();
    plan.tile_elems.clear();
    plan.tile_tverts.clear();
    plan.long_rows.clear();
    plan.row_ecodes.clear();
    plan.vert_gid.clear();
    plan.max_n_elem = 0;
    plan.elems_staged = true;
    plan.max_n_vert = plan.max_n_own = plan.max_n_halo = 0;
    plan.n_tiles = 0;
    plan.chain_order.clear();
    plan.hand_in.clear();
    plan.chain_len = 1;
    plan.runs.clear();
    plan.max_n_tv = 0;
  };
  std::vector<int32_t> vert_stamp(size_t(n_verts), -1);
  std::vector<TileSpec> specs;

  // ---- mode 1: a wave's 64 rows are 64 CONSECUTIVE vertices (one contiguous piece of the CSR
  // array per wave: the kernel's fast output path); a tile = up to own_cap / 64 such chunks that
  // lie next to each other (chunks sorted along the Z-order curve of their centroids).  Pays
  // when the numbering has locality (structured generators, RCM / Morton renumbering): accepted
  // when the tiles hold at most twice as many local vertices as the mesh has vertices.
  const int chunks_per_tile = std::min(own_cap / 64, 4);
  bool chunked = false;
  if (chunk_mode && chunks_per_tile >= 1 && n_verts > 0) {
    // chunks = up to 64 consecutive vertices; a chunk ends early where the numbering jumps
    // (the end of a grid line): more than 32 mean vertex spacings to the next vertex
    const double spacing = std::sqrt(std::max((hi[0] - lo[0]) * (hi[1] - lo[1]), 1e-300) / double(n_verts));
    const double jump2 = 32.0 * 32.0 * spacing * spacing;
    // TFEM_RING_CHUNK (developer switch): widest chunk, <= 64 rows of a wave
    int chunk_width = 64;
    if (const char *v = std::getenv("TFEM_RING_CHUNK")) chunk_width = std::max(8, std::min(std::atoi(v), 64));
    std::vector<int32_t> chunk_first;  // first vertex of every chunk, + n_verts as sentinel
    for (int64_t seg0 = 0; seg0 < n_verts;) {
      int64_t seg1 = seg0 + 1;  // [seg0, seg1): a piece of the numbering without a jump
      for (; seg1 < n_verts; ++seg1) {
        const double dx = coords[2 * seg1] - coords[2 * seg1 - 2];
        const double dy = coords[2 * seg1 + 1] - coords[2 * seg1 - 1];
        if (dx * dx + dy * dy > jump2) break;
      }
      // ceil(length / width) chunks of equal size (+-1): every wave equally loaded
      const int64_t len = seg1 - seg0, parts = (len + chunk_width - 1) / chunk_width;
      for (int64_t c = 0; c < parts; ++c) chunk_first.push_back(int32_t(seg0 + c * len / parts));
      seg0 = seg1;
    }
    chunk_first.push_back(int32_t(n_verts));
    const int64_t n_chunks = int64_t(chunk_first.size()) - 1;
    std::vector<std::pair<uint64_t, int32_t>> corder(static_cast<size_t>(n_chunks));
    parallel_for(n_chunks, [&](int64_t b, int64_t e, int) {
      for (int64_t c = b; c < e; ++c) {
        const int64_t v0 = chunk_first[size_t(c)], v1 = chunk_first[size_t(c) + 1];
        double cx = 0, cy = 0;
        for (int64_t v = v0; v < v1; ++v) {
          cx += coords[2 * v];
          cy += coords[2 * v + 1];
        }
        cx /= double(v1 - v0);
        cy /= double(v1 - v0);
        const uint64_t qx = std::min<uint64_t>(uint64_t((cx - lo[0]) * scale), (1u << 24) - 1);
        const uint64_t qy = std::min<uint64_t>(uint64_t((cy - lo[1]) * scale), (1u << 24) - 1);
        corder[size_t(c)] = {ring_spread_bits(qx) | (ring_spread_bits(qy) << 1), int32_t(c)};
      }
    }, 256);
    std::sort(corder.begin(), corder.end());
    lap("chunks + their curve order");
    chunked = true;
    // pass 1: group the chunks into tiles (neighbouring chunks along the curve while the local
    // vertices fit); pass 2: emit the tiles in the order of their first vertex, so that the
    // tiles the resident workgroups work on at the same time cover long contiguous ranges of
    // the coordinate, source-value and CSR value arrays (TFEM_RING_ORDER=curve keeps curve order)
    std::vector<int32_t> tile_chunks, chunk_fresh;
    std::vector<std::vector<int32_t>> groups;  // last entry of every group: its rank along the curve
    int64_t ccursor = 0;
    int32_t probe = -2;  // stamp values of pass 1: negative, distinct from every tile id
    int64_t probe_local = 0, probe_rows = 0;
    while (ccursor < n_chunks && chunked) {
      tile_chunks.clear();
      int n_local = 0, n_owned_rows = 0;
      --probe;
      while (ccursor < n_chunks && int(tile_chunks.size()) < chunks_per_tile) {
        const int64_t c = corder[size_t(ccursor)].second;
        const int64_t v0 = chunk_first[size_t(c)], v1 = chunk_first[size_t(c) + 1];
        // stamp what the chunk would add to the tile's local vertices (rolled back if the
        // tile overflows)
        int added = 0;
        chunk_fresh.clear();
        for (int64_t v = v0; v < v1; ++v) {
          auto touch = [&](int32_t w) {
            if (vert_stamp[size_t(w)] != probe) {
              vert_stamp[size_t(w)] = probe;
              chunk_fresh.push_back(w);
              ++added;
            }
          };
          touch(int32_t(v));
          for (int64_t p = rowptr[v]; p < rowptr[v + 1]; ++p) touch(colind[p]);
        }
        const int owned_after = n_owned_rows + int(v1 - v0);
        if (n_local + added > vert_cap || n_local + added - owned_after > kRingHaloCapHost) {
          for (int32_t w : chunk_fresh) vert_stamp[size_t(w)] = -1;  // roll back
          if (tile_chunks.empty()) chunked = false;  // one chunk alone does not fit
          break;
        }
        n_local += added;
        n_owned_rows = owned_after;
        tile_chunks.push_back(int32_t(c));
        ++ccursor;
      }
      if (!chunked) break;
      std::sort(tile_chunks.begin(), tile_chunks.end());
      tile_chunks.push_back(int32_t(groups.size()));
      groups.push_back(tile_chunks);
      probe_local += n_local;
      probe_rows += n_owned_rows;
      // a numbering without locality shows early: stop building this tiling
      if ((groups.size() & 255) == 0 && probe_local > 2 * probe_rows + 4096) chunked = false;
    }
    if (chunked && probe_local > 2 * n_verts) chunked = false;
    lap("grouping chunks into tiles");
    if (chunked) {
      const char *ord = std::getenv("TFEM_RING_ORDER");
      if (!(ord && std::strcmp(ord, "curve") == 0))
        std::sort(groups.begin(), groups.end(),
                  [](const std::vector<int32_t> &x, const std::vector<int32_t> &y) { return x[0] < y[0]; });
      specs.resize(groups.size());
      parallel_for(int64_t(groups.size()), [&](int64_t b, int64_t e, int) {
        for (int64_t t = b; t < e; ++t) {
          const std::vector<int32_t> &g = groups[size_t(t)];
          TileSpec &spec = specs[size_t(t)];
          spec.curve_rank = g.back();
          for (int w = 0; w < 5; ++w) {
            spec.wave_start[w] = int32_t(spec.owned.size());  // one chunk per wave
            if (w + 1 < int(g.size()))
              for (int32_t v = chunk_first[size_t(g[size_t(w)])]; v < chunk_first[size_t(g[size_t(w)]) + 1]; ++v)
                spec.owned.push_back(v);
          }
        }
      }, 64);
      priority_first(specs);
      const int st = emit_plan(conn, adj_ptr_data, adj_data, rowptr, colind, elem_ranges, specs, plan);
      if (st != TFEM_OK) return st;
      if (int64_t(plan.vert_gid.size()) > 2 * n_verts) chunked = false;
      if (!chunked) reset_plan();
      lap("emitting tiles");
      if (chunked) chain_pass(specs, plan);
      lap("chain order, element tables");
    }
  }
  plan.chunked = chunked;
  if (chunked) return TFEM_OK;

  // ---- mode 2: greedy tiling of the VERTICES along the Z-order curve (any numbering) ------------
  std::vector<std::pair<uint64_t, int32_t>> order(static_cast<size_t>(n_verts));
  parallel_for(n_verts, [&](int64_t b, int64_t e, int) {
    for (int64_t v = b; v < e; ++v) {
      const uint64_t qx = std::min<uint64_t>(uint64_t((coords[2 * v] - lo[0]) * scale), (1u << 24) - 1);
      const uint64_t qy = std::min<uint64_t>(uint64_t((coords[2 * v + 1] - lo[1]) * scale), (1u << 24) - 1);
      order[size_t(v)] = {ring_spread_bits(qx) | (ring_spread_bits(qy) << 1), int32_t(v)};
    }
  }, 4096);
  std::sort(order.begin(), order.end());
  lap("z-order sort of vertices");
  std::fill(vert_stamp.begin(), vert_stamp.end(), -1);
  specs.clear();
  std::vector<int32_t> fresh;
  int64_t cursor = 0;
  int32_t tile = 0;
  while (cursor < n_verts) {
    TileSpec spec;
    int n_local = 0;
    while (cursor < n_verts && int(spec.owned.size()) < own_cap) {
      const int32_t u = order[size_t(cursor)].second;
      // vertices `u` would add to the local set of the tile under construction
      fresh.clear();
      if (vert_stamp[size_t(u)] != tile) fresh.push_back(u);
      for (int64_t p = rowptr[u]; p < rowptr[u + 1]; ++p) {
        const int32_t w = colind[p];
        if (w != u && vert_stamp[size_t(w)] != tile) fresh.push_back(w);
      }
      if (n_local + int(fresh.size()) > vert_cap ||
          n_local + int(fresh.size()) - int(spec.owned.size()) - 1 > kRingHaloCapHost) {
        if (spec.owned.empty())
          return fail(TFEM_ERR_UNSUPPORTED, "vertex %d alone exceeds the tile capacity", u);
        break;
      }
      for (int32_t w : fresh) vert_stamp[size_t(w)] = tile;
      n_local += int(fresh.size());
      spec.owned.push_back(u);
      ++cursor;
    }
    // owned rows ascending (contiguous output runs), 64 per wave
    std::sort(spec.owned.begin(), spec.owned.end());
    for (int w = 0; w < 5; ++w) spec.wave_start[w] = std::min<int32_t>(64 * w, int32_t(spec.owned.size()));
    spec.curve_rank = tile;
    specs.push_back(std::move(spec));
    ++tile;
  }
  lap("greedy tiling along the curve");
  priority_first(specs);
  const int st = emit_plan(conn, adj_ptr_data, adj_data, rowptr, colind, elem_ranges, specs, plan);
  lap("emitting tiles");
  if (st == TFEM_OK) chain_pass(specs, plan);
  lap("chain order, element tables");
  return st;
}

void ring_layout(const RingPlan &p, int64_t layout[kRingLayoutLen]) {
  std::memset(layout, 0, sizeof(int64_t) * kRingLayoutLen);
  layout[0] = p.n_tiles;
  layout[1] = int64_t(p.rowstart.size());
  layout[2] = int64_t(p.vert_gid.size());
  layout[3] = p.max_n_vert;
  layout[4] = p.max_n_own;
  layout[5] = p.max_row_len;
  layout[6] = p.slots;
  layout[7] = p.words;
  const int64_t bytes[11] = {int64_t(p.desc.size()) * 4, int64_t(p.rows.size()) * 4,
                             int64_t(p.rowstart.size()) * 4, int64_t(p.vert_gid.size()) * 4,
                             int64_t(p.row_ecodes.size()) * 4, int64_t(p.tile_elems.size()) * 4,
                             int64_t(p.tile_tverts.size()) * 4, int64_t(p.long_rows.size()) * 4,
                             int64_t(p.chain_order.size()) * 4, int64_t(p.hand_in.size()) * 2,
                             int64_t(p.runs.size()) * 4};
  const int slot_of[11] = {8, 9, 10, 11, 15, 16, 20, 22, 24, 26, 30};
  int64_t off = 0;
  for (int i = 0; i < 11; ++i) {
    layout[slot_of[i]] = off;
    off += (bytes[i] + 15) & ~int64_t(15);
  }
  layout[12] = off + 64;
  layout[17] = p.max_n_elem;
  layout[18] = p.elems_staged ? 1 : 0;
  layout[19] = int64_t(p.tile_elems.size());
  layout[21] = int64_t(p.tile_tverts.size());
  layout[23] = int64_t(p.long_rows.size() / 24);
  layout[13] = p.chunked ? 1 : 0;
  layout[14] = p.max_n_halo;
  layout[25] = p.chain_len;
  layout[27] = p.max_n_tv;
  layout[28] = p.runs.empty() ? -1 : p.chain_wgs;  // number of runs (-1: blocks of chain_len positions)
}

}  // namespace
}  // namespace tfem

namespace tfem {
namespace {
int ring_plan_build(const void *conn_host, int idx_bytes, int64_t n_elems, int64_t n_verts,
                    const double *coords_host, const int64_t *rowptr_host, const int32_t *colind_host,
                    int own_cap, int vert_cap, const uint8_t *vertex_priority_host, const int64_t *inc_ptr,
                    int32_t *inc, void **plan_out, int64_t *n_priority_tiles) {
  if (!plan_out) return fail(TFEM_ERR_INVALID_ARGUMENT, "plan_out is NULL");
  *plan_out = nullptr;
  if (idx_bytes != 4 && idx_bytes != 8)
    return fail(TFEM_ERR_INVALID_ARGUMENT, "idx_bytes must be 4 or 8");
  if (n_elems < 0 || n_verts < 0 || (n_elems > 0 && !conn_host) || (n_verts > 0 && !coords_host) ||
      !rowptr_host || (rowptr_host[n_verts] > 0 && !colind_host))
    return fail(TFEM_ERR_INVALID_ARGUMENT, "bad arguments");
  if (own_cap < 1 || own_cap > 256 || vert_cap < 17 || vert_cap > 1024 || own_cap > vert_cap)
    return fail(TFEM_ERR_INVALID_ARGUMENT, "bad tile capacities");
  if (n_elems >= (int64_t(1) << 30) - 1 || rowptr_host[n_verts] >= (int64_t(1) << 31))
    return fail(TFEM_ERR_INDEX_RANGE, "mesh too large for the int32 ring plan");
  auto *plan = new RingPlan();
  // developer switch: TFEM_RING_TILES=zorder disables the consecutive-vertex tiles
  const char *mode = std::getenv("TFEM_RING_TILES");
  const bool chunk_mode = !(mode && std::strcmp(mode, "zorder") == 0);
  const int st =
      idx_bytes == 4
          ? build_rings(static_cast<const int32_t *>(conn_host), n_elems, n_verts, coords_host, rowptr_host,
                        colind_host, own_cap, vert_cap, chunk_mode, vertex_priority_host, inc_ptr, inc, *plan)
          : build_rings(static_cast<const int64_t *>(conn_host), n_elems, n_verts, coords_host, rowptr_host,
                        colind_host, own_cap, vert_cap, chunk_mode, vertex_priority_host, inc_ptr, inc, *plan);
  if (st != TFEM_OK) {
    delete plan;
    return st;
  }
  *plan_out = plan;
  if (n_priority_tiles) *n_priority_tiles = plan->n_priority;
  return TFEM_OK;
}
}  // namespace
}  // namespace tfem

extern "C" {


int tfem_ring_plan_create_priority(const void *conn_host, int idx_bytes, int64_t n_elems, int64_t n_verts,
                                   const double *coords_host, const int64_t *rowptr_host,
                                   const int32_t *colind_host, int own_cap, int vert_cap,
                                   const uint8_t *vertex_priority_host, void **plan_out,
                                   int64_t *n_priority_tiles) {
  return tfem::ring_plan_build(conn_host, idx_bytes, n_elems, n_verts, coords_host, rowptr_host, colind_host,
                               own_cap, vert_cap, vertex_priority_host, nullptr, nullptr, plan_out,
                               n_priority_tiles);
}

int tfem_ring_plan_create_from_pattern(void *pattern, const double *coords_host, const int32_t *colind_host,
                                       int own_cap, int vert_cap, const uint8_t *vertex_priority_host,
                                       void **plan_out, int64_t *n_priority_tiles) {
  tfem::PatternView v;
  if (!tfem::pattern_view(pattern, &v)) return tfem::fail(TFEM_ERR_INVALID_ARGUMENT, "pattern is NULL");
  if (v.n_local != 3)
    return tfem::fail(TFEM_ERR_INVALID_ARGUMENT, "the ring plan is built from the pattern of a P1 connectivity (3 DoFs per element), got %d", v.n_local);
  return tfem::ring_plan_build(v.conn, v.idx_bytes, v.n_elems, v.n_dofs, coords_host, v.rowptr, colind_host,
                               own_cap, vert_cap, vertex_priority_host, v.inc_ptr, v.inc, plan_out,
                               n_priority_tiles);
}

int tfem_ring_plan_create(const void *conn_host, int idx_bytes, int64_t n_elems, int64_t n_verts,
                          const double *coords_host, const int64_t *rowptr_host,
                          const int32_t *colind_host, int own_cap, int vert_cap, void **plan_out) {
  return tfem_ring_plan_create_priority(conn_host, idx_bytes, n_elems, n_verts, coords_host, rowptr_host,
                                        colind_host, own_cap, vert_cap, nullptr, plan_out, nullptr);
}

int tfem_ring_plan_sizes(const void *plan_handle, int64_t layout[32]) {
  using namespace tfem;
  if (!plan_handle || !layout) return fail(TFEM_ERR_INVALID_ARGUMENT, "NULL pointer");
  ring_layout(*static_cast<const RingPlan *>(plan_handle), layout);
  return TFEM_OK;
}

int tfem_ring_plan_pack(const void *plan_handle, void *blob_host) {
  using namespace tfem;
  if (!plan_handle || !blob_host) return fail(TFEM_ERR_INVALID_ARGUMENT, "NULL pointer");
  const auto *p = static_cast<const RingPlan *>(plan_handle);
  int64_t layout[kRingLayoutLen];
  ring_layout(*p, layout);
  auto *out = static_cast<unsigned char *>(blob_host);
  // the arrays, each cut into pieces for the threads; the padding between them and the 64 spare
  // bytes at the end are zero
  struct Part { int64_t off; const void *src; int64_t bytes; };
  const Part parts[11] = {
      {layout[8], p->desc.data(), int64_t(p->desc.size()) * 4},
      {layout[9], p->rows.data(), int64_t(p->rows.size()) * 4},
      {layout[10], p->rowstart.data(), int64_t(p->rowstart.size()) * 4},
      {layout[11], p->vert_gid.data(), int64_t(p->vert_gid.size()) * 4},
      {layout[15], p->row_ecodes.data(), int64_t(p->row_ecodes.size()) * 4},
      {layout[16], p->tile_elems.data(), int64_t(p->tile_elems.size()) * 4},
      {layout[20], p->tile_tverts.data(), int64_t(p->tile_tverts.size()) * 4},
      {layout[22], p->long_rows.data(), int64_t(p->long_rows.size()) * 4},
      {layout[24], p->chain_order.data(), int64_t(p->chain_order.size()) * 4},
      {layout[26], p->hand_in.data(), int64_t(p->hand_in.size()) * 2},
      {layout[30], p->runs.data(), int64_t(p->runs.size()) * 4},
  };
  for (const Part &part : parts) {
    const int64_t padded = (part.bytes + 15) & ~int64_t(15);
    std::memset(out + part.off + part.bytes, 0, size_t(padded - part.bytes));
    if (part.bytes == 0) continue;
    parallel_for(part.bytes, [&](int64_t b, int64_t e, int) {
      std::memcpy(out + part.off + b, static_cast<const unsigned char *>(part.src) + b, size_t(e - b));
    }, 1 << 20);
  }
  std::memset(out + layout[12] - 64, 0, 64);
  return TFEM_OK;
}

void tfem_ring_plan_destroy(void *plan_handle) { delete static_cast<tfem::RingPlan *>(plan_handle); }

}  // extern "C"
