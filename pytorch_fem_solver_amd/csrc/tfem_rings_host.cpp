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
// Supported fans: one closed cycle (interior vertex) or any number of open chains (boundary
// vertex, several fans meeting in a vertex).  An edge with three or more triangles, a
// duplicated or degenerate triangle, or a closed cycle beside another fan is reported as
// TFEM_ERR_UNSUPPORTED; the caller then uses the element-record tile plan instead.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>

#include "tfem_common.hpp"

namespace tfem {

constexpr int kRingDescStride = 20;
constexpr int kRingLayoutLen = 24;
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
  int32_t max_n_elem = 0;
  bool elems_staged = true;  // false: some tile has more than kRingElemCap elements
  std::vector<int32_t> vert_gid;  // global id of every tile-local vertex, owned rows first
  int32_t max_n_vert = 0, max_n_own = 0, max_row_len = 0, max_n_halo = 0;
  int64_t n_tiles = 0;
  bool chunked = false;           // every wave's 64 rows are 64 consecutive vertices
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

template <typename I>
int build_rings(const I *conn, int64_t n_elems, int64_t n_verts, const double *coords,
                const int64_t *rowptr, const int32_t *colind, int own_cap, int vert_cap,
                bool chunk_mode, RingPlan &plan) {
  // TFEM_RING_ELEM_RANGES=0: every tile stores its element list (A/B of the range encoding)
  const char *ranges_env = std::getenv("TFEM_RING_ELEM_RANGES");
  const bool elem_ranges = !(ranges_env && ranges_env[0] == '0');
  int64_t longest = 0;
  for (int64_t v = 0; v < n_verts; ++v) longest = std::max(longest, rowptr[v + 1] - rowptr[v]);
  if (longest > 16)
    return fail(TFEM_ERR_UNSUPPORTED, "a row has %lld entries (> 16)", (long long)longest);
  plan.max_row_len = int32_t(longest);
  plan.slots = longest <= 8 ? 7 : 15;
  plan.words = longest <= 8 ? 4 : 8;
  // ---- Z-order of the vertices ---------------------------------------------------------
  double lo[2] = {1e300, 1e300}, hi[2] = {-1e300, -1e300};
  for (int64_t v = 0; v < n_verts; ++v)
    for (int c = 0; c < 2; ++c) {
      lo[c] = std::min(lo[c], coords[2 * v + c]);
      hi[c] = std::max(hi[c], coords[2 * v + c]);
    }
  const double span = std::max(std::max(hi[0] - lo[0], hi[1] - lo[1]), 1e-300);
  std::vector<std::pair<uint64_t, int32_t>> order(static_cast<size_t>(n_verts));
  const double scale = double(1u << 24) / span;
  for (int64_t v = 0; v < n_verts; ++v) {
    const uint64_t qx = std::min<uint64_t>(uint64_t((coords[2 * v] - lo[0]) * scale), (1u << 24) - 1);
    const uint64_t qy = std::min<uint64_t>(uint64_t((coords[2 * v + 1] - lo[1]) * scale), (1u << 24) - 1);
    order[size_t(v)] = {ring_spread_bits(qx) | (ring_spread_bits(qy) << 1), int32_t(v)};
  }
  std::sort(order.begin(), order.end());
  // ---- vertex -> incident elements ------------------------------------------------------
  std::vector<int64_t> adj_ptr(size_t(n_verts) + 1, 0);
  for (int64_t k = 0; k < 3 * n_elems; ++k) adj_ptr[size_t(conn[k]) + 1]++;
  std::partial_sum(adj_ptr.begin(), adj_ptr.end(), adj_ptr.begin());
  std::vector<int32_t> adj(size_t(3 * n_elems));
  {
    std::vector<int64_t> cur(adj_ptr.begin(), adj_ptr.end() - 1);
    for (int64_t e = 0; e < n_elems; ++e)
      for (int a = 0; a < 3; ++a) adj[size_t(cur[size_t(conn[3 * e + a])]++)] = int32_t(e);
  }
  // ---- tiles -----------------------------------------------------------------------------
  std::vector<int32_t> vert_stamp(size_t(n_verts), -1), vert_local(size_t(n_verts), 0);
  std::vector<int32_t> owned, fresh;
  std::vector<int32_t> elem_stamp(size_t(n_elems), -1), elem_local(size_t(n_elems), 0);
  std::vector<int32_t> fan_elem, fan_loc, elems_here;
  int32_t wave_start[5] = {0, 0, 0, 0, 0};  // of the tile under construction
  int32_t tile = 0;
  Fan fan;
  int status = TFEM_OK;
  // vertices `u` would add to the local set of the tile under construction
  auto collect_fresh = [&](int32_t u) {
    if (vert_stamp[size_t(u)] != tile) fresh.push_back(u);
    for (int64_t p = rowptr[u]; p < rowptr[u + 1]; ++p) {
      const int32_t w = colind[p];
      if (w != u && vert_stamp[size_t(w)] != tile) fresh.push_back(w);
    }
  };
  // numbers the tile's vertices (owned rows first, in the order given, then the halo in order of
  // first reference) and writes its row records
  auto emit_tile = [&]() {
    const int n_own = int(owned.size());
    const int32_t vert_off = int32_t(plan.vert_gid.size());
    const int32_t row_off = int32_t(plan.rowstart.size());
    for (int l = 0; l < n_own; ++l) {
      vert_local[size_t(owned[size_t(l)])] = l;
      vert_stamp[size_t(owned[size_t(l)])] = -2 - tile;  // numbered
    }
    plan.vert_gid.insert(plan.vert_gid.end(), owned.begin(), owned.end());
    int next_local = n_own;
    for (int l = 0; l < n_own; ++l) {
      const int32_t u = owned[size_t(l)];
      for (int64_t p = rowptr[u]; p < rowptr[u + 1]; ++p) {
        const int32_t w = colind[p];
        if (vert_stamp[size_t(w)] == tile) {
          vert_stamp[size_t(w)] = -2 - tile;
          vert_local[size_t(w)] = next_local++;
          plan.vert_gid.push_back(w);
        }
      }
    }
    fan_elem.assign(size_t(n_own) * size_t(plan.slots), -1);
    fan_loc.assign(size_t(n_own) * size_t(plan.slots), 0);
    elems_here.clear();
    for (int l = 0; l < n_own; ++l) {
      const int32_t u = owned[size_t(l)];
      const int len = int(rowptr[u + 1] - rowptr[u]);
      if (!build_fan(conn, u, adj.data() + adj_ptr[size_t(u)], adj.data() + adj_ptr[size_t(u) + 1], fan)) {
        status = fail(TFEM_ERR_UNSUPPORTED, "the triangles around vertex %d do not form fans", u);
        return;
      }
      if (len != (fan.k ? fan.k + 1 : 0) || fan.k > plan.slots) {
        status = fail(TFEM_ERR_UNSUPPORTED, "row %d: %d entries for %d neighbours", u, len, fan.k);
        return;
      }
      uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      const int32_t *first = colind + rowptr[u];
      const int32_t *last = colind + rowptr[u + 1];
      const uint32_t k = uint32_t(fan.k);
      const uint32_t dpos = len ? uint32_t(std::lower_bound(first, last, u) - first) : 0u;
      for (int i = 0; i < fan.k; ++i) {
        const uint32_t lid = uint32_t(vert_local[size_t(fan.nb[i])]);
        const uint32_t pos = uint32_t(std::lower_bound(first, last, fan.nb[i]) - first);
        const uint32_t flag = uint32_t(fan.flag[i]);
        w[i / 3] |= lid << (10 * (i % 3));
        if (plan.slots == 7) {
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
      if (plan.slots == 7) {
        w[0] |= (k & 3u) << 30;
        w[1] |= (k >> 2) << 30;
        w[2] |= dpos << 24;
      } else {
        w[0] |= (k & 3u) << 30;
        w[1] |= (k >> 2) << 30;
        w[2] |= (dpos & 3u) << 30;
        w[3] |= (dpos >> 2) << 30;
      }
      plan.rows.insert(plan.rows.end(), w, w + plan.words);
      plan.rowstart.push_back(int32_t(rowptr[u]));
      for (int i = 0; i < fan.k; ++i)
        if (fan.flag[i] != 0) {
          fan_elem[size_t(l) * size_t(plan.slots) + size_t(i)] = fan.elem[i];
          fan_loc[size_t(l) * size_t(plan.slots) + size_t(i)] = fan.loc[i];
          if (elem_stamp[size_t(fan.elem[i])] != tile) {
            elem_stamp[size_t(fan.elem[i])] = tile;
            elems_here.push_back(fan.elem[i]);
          }
        }
    }
    // the tile's elements, ascending (coalesced source-value loads), and the slot codes
    std::sort(elems_here.begin(), elems_here.end());
    const int32_t elem_off = int32_t(plan.tile_elems.size());
    const int32_t n_elem = int32_t(elems_here.size());
    const int32_t tvert_off = int32_t(plan.tile_tverts.size());
    for (int32_t j = 0; j < n_elem; ++j) {
      const I *c = conn + 3 * int64_t(elems_here[size_t(j)]);
      plan.tile_tverts.push_back(uint32_t(vert_local[size_t(c[0])]) | uint32_t(vert_local[size_t(c[1])]) << 10 |
                                 uint32_t(vert_local[size_t(c[2])]) << 20);
    }
    if (n_elem > kRingElemCap) plan.elems_staged = false;
    for (int32_t j = 0; j < n_elem; ++j) elem_local[size_t(elems_here[size_t(j)])] = j;
    // Element numberings with locality: the ascending list is a few runs of consecutive ids.
    // Up to kRingElemRuns of them are stored as 16 ints -- first id of every run, then the
    // number of elements up to and including every run (n_elem for the unused ones) -- and the
    // kernel derives the ids (desc[18] = 1); otherwise the list itself (desc[18] = 0).
    int32_t elem_mode = 0;
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
        plan.tile_elems.insert(plan.tile_elems.end(), start, start + kRingElemRuns);
        plan.tile_elems.insert(plan.tile_elems.end(), upto, upto + kRingElemRuns);
        elem_mode = 1;
      }
    }
    if (!elem_mode) plan.tile_elems.insert(plan.tile_elems.end(), elems_here.begin(), elems_here.end());
    plan.max_n_elem = std::max(plan.max_n_elem, n_elem);
    const int ewords = (12 * plan.slots + 31) / 32;  // 12-bit codes, packed: 3 (7 slots) or 6 dwords
    for (int l = 0; l < n_own; ++l) {
      uint32_t ew[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int i = 0; i < plan.slots; ++i) {
        uint32_t code = 0xFFFu;
        if (fan_elem[size_t(l) * size_t(plan.slots) + size_t(i)] >= 0) {
          const int32_t le = elem_local[size_t(fan_elem[size_t(l) * size_t(plan.slots) + size_t(i)])];
          code = le < 1023 ? uint32_t(le) | uint32_t(fan_loc[size_t(l) * size_t(plan.slots) + size_t(i)]) << 10 : 0xFFFu;
        }
        const int bit = 12 * i;
        ew[bit / 32] |= code << (bit % 32);
        if (bit % 32 > 20) ew[bit / 32 + 1] |= code >> (32 - bit % 32);
      }
      plan.row_ecodes.insert(plan.row_ecodes.end(), ew, ew + ewords);
    }
    int32_t d[kRingDescStride] = {vert_off, next_local, row_off, 0,
                                  wave_start[1], wave_start[2], wave_start[3], n_own,
                                  0, 0, 0, 0, 0, 0, 0, 0, elem_off, n_elem, elem_mode, tvert_off};
    plan.max_n_halo = std::max(plan.max_n_halo, next_local - n_own);
    for (int w = 0; w < 4; ++w)  // first vertex and first CSR entry of every wave's rows
      if (wave_start[w] < wave_start[w + 1]) {
        d[8 + w] = owned[size_t(wave_start[w])];
        d[12 + w] = int32_t(rowptr[owned[size_t(wave_start[w])]]);
      }
    plan.desc.insert(plan.desc.end(), d, d + kRingDescStride);
    plan.max_n_vert = std::max(plan.max_n_vert, next_local);
    plan.max_n_own = std::max(plan.max_n_own, int32_t(n_own));
    ++tile;
  };
  auto reset_plan = [&]() {
    plan.desc.clear();
    plan.rows.clear();
    plan.rowstart.clear();
    plan.tile_elems.clear();
    plan.tile_tverts.clear();
    plan.row_ecodes.clear();
    plan.max_n_elem = 0;
    plan.elems_staged = true;
    std::fill(elem_stamp.begin(), elem_stamp.end(), -1);
    plan.vert_gid.clear();
    plan.max_n_vert = plan.max_n_own = plan.max_n_halo = 0;
    std::fill(vert_stamp.begin(), vert_stamp.end(), -1);
    tile = 0;
  };
  plan.rows.reserve(size_t(n_verts) * size_t(plan.words));
  plan.rowstart.reserve(size_t(n_verts));

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
    std::vector<int32_t> chunk_first;  // first vertex of every chunk, + n_verts as sentinel
    for (int64_t seg0 = 0; seg0 < n_verts;) {
      int64_t seg1 = seg0 + 1;  // [seg0, seg1): a piece of the numbering without a jump
      for (; seg1 < n_verts; ++seg1) {
        const double dx = coords[2 * seg1] - coords[2 * seg1 - 2];
        const double dy = coords[2 * seg1 + 1] - coords[2 * seg1 - 1];
        if (dx * dx + dy * dy > jump2) break;
      }
      // ceil(length / 64) chunks of equal size (+-1): every wave equally loaded
      const int64_t len = seg1 - seg0, parts = (len + 63) / 64;
      for (int64_t c = 0; c < parts; ++c) chunk_first.push_back(int32_t(seg0 + c * len / parts));
      seg0 = seg1;
    }
    chunk_first.push_back(int32_t(n_verts));
    const int64_t n_chunks = int64_t(chunk_first.size()) - 1;
    std::vector<std::pair<uint64_t, int32_t>> corder(static_cast<size_t>(n_chunks));
    for (int64_t c = 0; c < n_chunks; ++c) {
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
    std::sort(corder.begin(), corder.end());
    chunked = true;
    // pass 1: group the chunks into tiles (neighbouring chunks along the curve while the local
    // vertices fit); pass 2: emit the tiles in the order of their first vertex, so that the
    // tiles the resident workgroups work on at the same time cover long contiguous ranges of
    // the coordinate, source-value and CSR value arrays (TFEM_RING_ORDER=curve keeps curve order)
    std::vector<int32_t> tile_chunks, chunk_fresh;
    std::vector<std::vector<int32_t>> groups;
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
      groups.push_back(tile_chunks);
      probe_local += n_local;
      probe_rows += n_owned_rows;
      // a numbering without locality shows early: stop building this tiling
      if ((groups.size() & 255) == 0 && probe_local > 2 * probe_rows + 4096) chunked = false;
    }
    if (chunked && probe_local > 2 * n_verts) chunked = false;
    if (chunked) {
      const char *ord = std::getenv("TFEM_RING_ORDER");
      if (!(ord && std::strcmp(ord, "curve") == 0))
        std::sort(groups.begin(), groups.end(),
                  [](const std::vector<int32_t> &x, const std::vector<int32_t> &y) { return x[0] < y[0]; });
      std::fill(vert_stamp.begin(), vert_stamp.end(), -1);
      for (const std::vector<int32_t> &g : groups) {
        owned.clear();
        for (int w = 0; w < 5; ++w) {
          wave_start[w] = int32_t(owned.size());  // one chunk per wave
          if (w < int(g.size()))
            for (int32_t v = chunk_first[size_t(g[size_t(w)])]; v < chunk_first[size_t(g[size_t(w)]) + 1]; ++v)
              owned.push_back(v);
        }
        // the tile's local vertices: stamped with the tile id, as emit_tile expects
        for (int32_t u : owned) {
          fresh.clear();
          collect_fresh(u);
          for (int32_t w : fresh) vert_stamp[size_t(w)] = tile;
        }
        emit_tile();
        if (status != TFEM_OK) return status;
      }
    } else {
      std::fill(vert_stamp.begin(), vert_stamp.end(), -1);
    }
    if (status != TFEM_OK) return status;
    if (chunked && int64_t(plan.vert_gid.size()) > 2 * n_verts) chunked = false;
    if (!chunked) reset_plan();
  }
  plan.chunked = chunked;

  // ---- mode 2: greedy tiling of the VERTICES along the Z-order curve (any numbering) ------------
  int64_t cursor = 0;
  while (!chunked && cursor < n_verts) {
    owned.clear();
    int n_local = 0;
    while (cursor < n_verts && int(owned.size()) < own_cap) {
      const int32_t u = order[size_t(cursor)].second;
      fresh.clear();
      collect_fresh(u);
      if (n_local + int(fresh.size()) > vert_cap ||
          n_local + int(fresh.size()) - int(owned.size()) - 1 > kRingHaloCapHost) {
        if (owned.empty())
          return fail(TFEM_ERR_UNSUPPORTED, "vertex %d alone exceeds the tile capacity", u);
        break;
      }
      for (int32_t w : fresh) vert_stamp[size_t(w)] = tile;
      n_local += int(fresh.size());
      owned.push_back(u);
      ++cursor;
    }
    // owned rows ascending (contiguous output runs), 64 per wave
    std::sort(owned.begin(), owned.end());
    for (int w = 0; w < 5; ++w) wave_start[w] = std::min<int32_t>(64 * w, int32_t(owned.size()));
    emit_tile();
    if (status != TFEM_OK) return status;
  }
  plan.n_tiles = tile;
  return TFEM_OK;
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
  const int64_t bytes[7] = {int64_t(p.desc.size()) * 4, int64_t(p.rows.size()) * 4,
                            int64_t(p.rowstart.size()) * 4, int64_t(p.vert_gid.size()) * 4,
                            int64_t(p.row_ecodes.size()) * 4, int64_t(p.tile_elems.size()) * 4,
                            int64_t(p.tile_tverts.size()) * 4};
  const int slot_of[7] = {8, 9, 10, 11, 15, 16, 20};
  int64_t off = 0;
  for (int i = 0; i < 7; ++i) {
    layout[slot_of[i]] = off;
    off += (bytes[i] + 15) & ~int64_t(15);
  }
  layout[12] = off + 64;
  layout[17] = p.max_n_elem;
  layout[18] = p.elems_staged ? 1 : 0;
  layout[19] = int64_t(p.tile_elems.size());
  layout[21] = int64_t(p.tile_tverts.size());
  layout[13] = p.chunked ? 1 : 0;
  layout[14] = p.max_n_halo;
}

}  // namespace
}  // namespace tfem

extern "C" {

int tfem_ring_plan_create(const void *conn_host, int idx_bytes, int64_t n_elems, int64_t n_verts,
                          const double *coords_host, const int64_t *rowptr_host,
                          const int32_t *colind_host, int own_cap, int vert_cap, void **plan_out) {
  using namespace tfem;
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
          ? build_rings(static_cast<const int32_t *>(conn_host), n_elems, n_verts, coords_host,
                        rowptr_host, colind_host, own_cap, vert_cap, chunk_mode, *plan)
          : build_rings(static_cast<const int64_t *>(conn_host), n_elems, n_verts, coords_host,
                        rowptr_host, colind_host, own_cap, vert_cap, chunk_mode, *plan);
  if (st != TFEM_OK) {
    delete plan;
    return st;
  }
  *plan_out = plan;
  return TFEM_OK;
}

int tfem_ring_plan_sizes(const void *plan_handle, int64_t layout[24]) {
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
  std::memset(out, 0, size_t(layout[12]));
  std::memcpy(out + layout[8], p->desc.data(), p->desc.size() * 4);
  std::memcpy(out + layout[9], p->rows.data(), p->rows.size() * 4);
  std::memcpy(out + layout[10], p->rowstart.data(), p->rowstart.size() * 4);
  std::memcpy(out + layout[11], p->vert_gid.data(), p->vert_gid.size() * 4);
  std::memcpy(out + layout[15], p->row_ecodes.data(), p->row_ecodes.size() * 4);
  std::memcpy(out + layout[16], p->tile_elems.data(), p->tile_elems.size() * 4);
  std::memcpy(out + layout[20], p->tile_tverts.data(), p->tile_tverts.size() * 4);
  return TFEM_OK;
}

void tfem_ring_plan_destroy(void *plan_handle) { delete static_cast<tfem::RingPlan *>(plan_handle); }

}  // extern "C"
