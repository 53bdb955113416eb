// Row plan for the P2 (quadratic) stiffness/mass kernels (host, once per mesh).
//
// Owner-computes row form, like the P1 ring plan (tfem_rings_host.cpp): one lane owns one CSR
// row.  P2 DoFs are the vertices (DoF id = vertex id) followed by the edges (local order
// (v0,v1), (v1,v2), (v2,v0), element_tri.py:50-52); the two kinds of row get their own tiles,
// records and kernel instantiation:
//   vertex row v: the fan of v as in the P1 plan (neighbours n_0..n_{k-1} in fan order, a
//     triangle flag per slot); its columns are v, the k neighbours, the k edges (v, n_i) and
//     the opposite edge (n_i, n_next) of every triangle: 1 + 2k + t entries.
//   edge row (a, b): its one or two triangles, each in its own stored cyclic frame
//     (a, b, c) / (a2, b2, d); columns a, b, c, d, itself and the four other edges: 6 or 9.
// Tiles are made of chunks of <= 64 CONSECUTIVE DoFs (one chunk per wave: a wave writes one
// contiguous piece of the CSR array, its offset is in the descriptor), four chunks per tile.
// Vertices with 8 to 15 neighbours (every Delaunay mesh has some) keep their place in the tiles
// as rows of length zero and are listed as LONG ROWS: 32-dword records with global ids, handled
// by a launch of their own (one lane per long row, k_p2_long_rows).
// Meshes whose numbering has no locality, vertices with more than 15 neighbours, fans without
// ring form and DoF layouts other than "vertices, then edges" are reported as
// TFEM_ERR_UNSUPPORTED: the caller then assembles element blocks and gathers them
// (tfem_tri_bilinear_csr in local-block mode + tfem_csr_gather).
//
// Vertex-row record (8 dwords):
//   w0, w1, w2[0:10)  : local ids of the neighbours, 10 bits each (3 per dword)
//   w2[10:24)         : triangle flag of every slot, 2 bits (as in the P1 ring plan)
//   w2[24:27)         : k;  w2[27:32): position of the diagonal
//   w3..w6            : 21 positions of 5 bits, 6 per dword: field i = column n_i, field 7 + i =
//                       edge (v, n_i), field 14 + i = opposite edge of slot i's triangle
//   a LONG row has k = 0 and w3 = 0x80000000 | number of entries of the row (the rows behind it in
//   the wave start that many entries later in the CSR array)
// Long-row record (32 dwords): vertex id, CSR offset of the row, k | position of the diagonal << 8,
//   triangle flags (2 bits per slot), 15 neighbour vertex ids (global), then 45 positions of 6 bits,
//   5 per dword, from dword 19: field i = column n_i, 15 + i = edge (v, n_i), 30 + i = opposite edge
// Edge-row record (4 dwords):
//   w0 : local ids a | b << 10 | c << 20 (frame of triangle 1)
//   w1 : local id d | has2 << 10 | rev << 11 (triangle 2's frame starts at b: (b, a, d))
//   w2 : positions, 4 bits each: a, b, c, self, edge (b,c), edge (c,a), d, edge (b2,d)
//   w3 : position of edge (d,a2)
// Element codes (load vector in row form, tfem_p2load.hip), in the order of the records: per fan
//   slot of a vertex row (8 dwords per row; 16 per long row) and per triangle of an edge row (2
//   dwords) element * 4 + the local index of the row's DoF among the three DoFs of its kind in that
//   element (0xFFFFFFFF: no triangle).
// Descriptor (16 ints per tile): vert_off, n_vert, row_off, 0, first row of wave 1, 2, 3,
//   n_own, vertex id of the first row of wave 0..3 (vertex tiles), CSR offset of the first
//   row of wave 0..3.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>
#include <vector>

#include "tfem_common.hpp"

namespace tfem {

constexpr int kP2LayoutLen = 24;
constexpr int kP2VertCap = 1000;  // local vertices per tile (10-bit ids)
constexpr int kP2HaloCap = 256;   // halo vertices of a vertex tile: one per lane

struct P2Plan {
  // [0] vertex-row tiles, [1] edge-row tiles
  std::vector<int32_t> desc[2];
  std::vector<uint32_t> rows[2];
  std::vector<int32_t> vert_gid[2];
  std::vector<uint32_t> long_rows;  // 32 dwords per vertex with 8 .. 15 neighbours
  // load vector in row form (tfem_p2_load_rows): per fan slot / per triangle of an edge the element
  // and the local index the row's DoF has in it, element * 4 + index (0xFFFFFFFF: no triangle);
  // 8 dwords per vertex row, 2 per edge row, 16 per long row, in the order of the row records
  std::vector<uint32_t> codes[2], long_codes;
  int32_t max_n_vert[2] = {0, 0}, max_n_halo = 0;
  int64_t n_tiles[2] = {0, 0};
  int64_t n_verts = 0, n_edges = 0;
};

namespace {

constexpr int kP2FanMax = 15;  // neighbours of a vertex the plan can express (7 in the tile records)

struct P2Fan {
  int k = 0;
  int32_t nb[kP2FanMax + 1];
  int flag[kP2FanMax + 1];
  int32_t edge_v[kP2FanMax + 1];    // DoF of edge (v, n_i)
  int32_t edge_op[kP2FanMax + 1];   // DoF of the edge opposite to v in slot i's triangle (-1: none)
  uint32_t code[kP2FanMax + 1];     // slot i's triangle: element * 4 + local index of v (0xFFFFFFFF: none)
};

// Fan of vertex v from its incident elements (conn6: three vertices, three edge DoFs).
bool build_p2_fan(const int32_t *conn6, int32_t v, const int32_t *adj_first, const int32_t *adj_last,
                  P2Fan &fan) {
  const int nt = int(adj_last - adj_first);
  fan.k = 0;
  if (nt == 0) return true;
  if (nt > kP2FanMax) return false;
  int32_t ta[kP2FanMax + 1], tb[kP2FanMax + 1], tea[kP2FanMax + 1], teo[kP2FanMax + 1], teb[kP2FanMax + 1];
  int tj[kP2FanMax + 1];
  int32_t nb[kP2FanMax + 2], nb_edge[kP2FanMax + 2];
  int cnt[kP2FanMax + 2], tri[kP2FanMax + 2][2];
  int n_nb = 0;
  for (int t = 0; t < nt; ++t) {
    const int32_t *c = conn6 + 6 * int64_t(adj_first[t]);
    int j = -1, hits = 0;
    for (int a = 0; a < 3; ++a)
      if (c[a] == v) {
        j = a;
        ++hits;
      }
    if (hits != 1) return false;
    tj[t] = j;
    ta[t] = c[(j + 1) % 3];
    tb[t] = c[(j + 2) % 3];
    tea[t] = c[3 + j];            // edge (v, a)
    teo[t] = c[3 + (j + 1) % 3];  // edge (a, b)
    teb[t] = c[3 + (j + 2) % 3];  // edge (b, v)
    if (ta[t] == tb[t]) return false;
    for (int side = 0; side < 2; ++side) {
      const int32_t w = side ? tb[t] : ta[t];
      const int32_t ew = side ? teb[t] : tea[t];
      int i = 0;
      while (i < n_nb && nb[i] != w) ++i;
      if (i == n_nb) {
        if (n_nb == kP2FanMax + 1) return false;
        nb[n_nb] = w;
        nb_edge[n_nb] = ew;
        cnt[n_nb] = 0;
        ++n_nb;
      }
      if (nb_edge[i] != ew) return false;  // two DoFs for one edge
      if (cnt[i] == 2) return false;
      tri[i][cnt[i]++] = t;
    }
  }
  if (n_nb > kP2FanMax) return false;
  for (int t = 0; t < nt; ++t)
    for (int u = t + 1; u < nt; ++u)
      if ((ta[t] == ta[u] && tb[t] == tb[u]) || (ta[t] == tb[u] && tb[t] == ta[u])) return false;
  bool used[kP2FanMax + 1] = {false};
  bool seen[kP2FanMax + 2] = {false};
  auto index_of = [&](int32_t w) {
    int i = 0;
    while (nb[i] != w) ++i;
    return i;
  };
  auto walk = [&](int c) {
    for (;;) {
      seen[c] = true;
      fan.nb[fan.k] = nb[c];
      fan.edge_v[fan.k] = nb_edge[c];
      fan.flag[fan.k] = 0;
      fan.edge_op[fan.k] = -1;
      fan.code[fan.k] = 0xFFFFFFFFu;
      int t = -1;
      for (int s = 0; s < cnt[c]; ++s)
        if (!used[tri[c][s]]) {
          t = tri[c][s];
          break;
        }
      if (t < 0) {
        ++fan.k;
        return c;
      }
      used[t] = true;
      const bool forward = ta[t] == nb[c];
      fan.flag[fan.k] = forward ? 1 : 2;
      fan.edge_op[fan.k] = teo[t];
      fan.code[fan.k] = uint32_t(adj_first[t]) * 4u + uint32_t(tj[t]);
      ++fan.k;
      const int o = index_of(forward ? tb[t] : ta[t]);
      if (seen[o]) return o;
      c = o;
    }
  };
  bool any_end = false;
  for (int i = 0; i < n_nb; ++i) any_end = any_end || cnt[i] == 1;
  if (!any_end) {
    int start = 0;
    for (int i = 1; i < n_nb; ++i)
      if (nb[i] < nb[start]) start = i;
    const int stop = walk(start);
    if (stop != start || fan.k != n_nb) return false;
  } else {
    for (;;) {
      int start = -1;
      for (int i = 0; i < n_nb; ++i)
        if (!seen[i] && cnt[i] == 1 && (start < 0 || nb[i] < nb[start])) start = i;
      if (start < 0) break;
      walk(start);
      if (fan.flag[fan.k - 1] != 0) return false;
    }
    if (fan.k != n_nb) return false;
  }
  for (int t = 0; t < nt; ++t)
    if (!used[t]) return false;
  return true;
}

inline uint32_t pos_in_row(const int64_t *rowptr, const int32_t *colind, int64_t row, int32_t col,
                           bool &ok) {
  const int32_t *first = colind + rowptr[row];
  const int32_t *last = colind + rowptr[row + 1];
  const int32_t *it = std::lower_bound(first, last, col);
  if (it == last || *it != col) ok = false;
  return uint32_t(it - first);
}

// Chunks of <= 64 consecutive ids in [0, n): cut where `jump(i)` says the numbering jumps
// between i - 1 and i; every piece is split into equal parts.
template <typename Jump>
std::vector<int32_t> make_chunks(int64_t n, Jump jump) {
  std::vector<int32_t> first;
  for (int64_t seg0 = 0; seg0 < n;) {
    int64_t seg1 = seg0 + 1;
    while (seg1 < n && !jump(seg1)) ++seg1;
    const int64_t len = seg1 - seg0, parts = (len + 63) / 64;
    for (int64_t c = 0; c < parts; ++c) first.push_back(int32_t(seg0 + c * len / parts));
    seg0 = seg1;
  }
  first.push_back(int32_t(n));
  return first;
}

int build_p2(const int32_t *conn6, int64_t n_elems, int64_t n_verts, int64_t n_dofs,
             const double *coords, const int64_t *rowptr, const int32_t *colind, P2Plan &plan) {
  const int64_t n_edges = n_dofs - n_verts;
  plan.n_verts = n_verts;
  plan.n_edges = n_edges;
  for (int64_t t = 0; t < n_elems; ++t)
    for (int a = 0; a < 6; ++a) {
      const int32_t g = conn6[6 * t + a];
      if (a < 3 ? (g < 0 || g >= n_verts) : (g < n_verts || g >= n_dofs))
        return fail(TFEM_ERR_UNSUPPORTED, "P2 DoFs are not numbered vertices first, then edges");
    }
  double lo[2] = {1e300, 1e300}, hi[2] = {-1e300, -1e300};
  for (int64_t v = 0; v < n_verts; ++v)
    for (int c = 0; c < 2; ++c) {
      lo[c] = std::min(lo[c], coords[2 * v + c]);
      hi[c] = std::max(hi[c], coords[2 * v + c]);
    }
  const double spacing = std::sqrt(std::max((hi[0] - lo[0]) * (hi[1] - lo[1]), 1e-300) / double(std::max<int64_t>(n_verts, 1)));
  const double jump2 = 32.0 * 32.0 * spacing * spacing;
  // vertex -> incident elements; edge -> its (at most two) elements and local edge index
  std::vector<int64_t> adj_ptr(size_t(n_verts) + 1, 0);
  for (int64_t t = 0; t < n_elems; ++t)
    for (int a = 0; a < 3; ++a) adj_ptr[size_t(conn6[6 * t + a]) + 1]++;
  std::partial_sum(adj_ptr.begin(), adj_ptr.end(), adj_ptr.begin());
  std::vector<int32_t> adj(size_t(3 * n_elems));
  {
    std::vector<int64_t> cur(adj_ptr.begin(), adj_ptr.end() - 1);
    for (int64_t t = 0; t < n_elems; ++t)
      for (int a = 0; a < 3; ++a) adj[size_t(cur[size_t(conn6[6 * t + a])]++)] = int32_t(t);
  }
  std::vector<int32_t> edge_elem(size_t(2 * n_edges), -1);  // element * 4 + local edge
  std::vector<double> edge_mid(size_t(2 * n_edges), 0.0);
  for (int64_t t = 0; t < n_elems; ++t)
    for (int m = 0; m < 3; ++m) {
      const int64_t e = conn6[6 * t + 3 + m] - n_verts;
      int32_t *slot = &edge_elem[size_t(2 * e)];
      if (slot[0] < 0)
        slot[0] = int32_t(4 * t + m);
      else if (slot[1] < 0)
        slot[1] = int32_t(4 * t + m);
      else
        return fail(TFEM_ERR_UNSUPPORTED, "edge DoF %lld belongs to three elements", (long long)(e + n_verts));
      const int32_t a = conn6[6 * t + m], b = conn6[6 * t + (m + 1) % 3];
      edge_mid[size_t(2 * e)] = 0.5 * (coords[2 * a] + coords[2 * b]);
      edge_mid[size_t(2 * e) + 1] = 0.5 * (coords[2 * a + 1] + coords[2 * b + 1]);
    }
  for (int64_t e = 0; e < n_edges; ++e)
    if (edge_elem[size_t(2 * e)] < 0)
      return fail(TFEM_ERR_UNSUPPORTED, "edge DoF %lld belongs to no element", (long long)(e + n_verts));

  std::vector<int32_t> stamp(size_t(n_verts), -1), local_id(size_t(n_verts), 0);
  int32_t serial = 0;  // one stamp value per tile, both kinds

  // ---------------------------------------------------------------- vertex-row tiles
  {
    const std::vector<int32_t> chunk_first = make_chunks(n_verts, [&](int64_t v) {
      const double dx = coords[2 * v] - coords[2 * v - 2], dy = coords[2 * v + 1] - coords[2 * v - 1];
      return dx * dx + dy * dy > jump2;
    });
    const int64_t n_chunks = int64_t(chunk_first.size()) - 1;
    // chunks sorted along a Z-order curve of their centroids: neighbouring chunks share a tile
    auto spread = [](uint64_t x) {
      x &= 0xFFFFFFFFull;
      x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
      x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
      x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
      x = (x | (x << 2)) & 0x3333333333333333ull;
      x = (x | (x << 1)) & 0x5555555555555555ull;
      return x;
    };
    const double span = std::max(std::max(hi[0] - lo[0], hi[1] - lo[1]), 1e-300);
    const double scale = double(1u << 24) / span;
    std::vector<std::pair<uint64_t, int32_t>> corder(static_cast<size_t>(n_chunks));
    for (int64_t c = 0; c < n_chunks; ++c) {
      double cx = 0, cy = 0;
      for (int64_t v = chunk_first[size_t(c)]; v < chunk_first[size_t(c) + 1]; ++v) {
        cx += coords[2 * v];
        cy += coords[2 * v + 1];
      }
      const double cnt = double(chunk_first[size_t(c) + 1] - chunk_first[size_t(c)]);
      const uint64_t qx = std::min<uint64_t>(uint64_t((cx / cnt - lo[0]) * scale), (1u << 24) - 1);
      const uint64_t qy = std::min<uint64_t>(uint64_t((cy / cnt - lo[1]) * scale), (1u << 24) - 1);
      corder[size_t(c)] = {spread(qx) | (spread(qy) << 1), int32_t(c)};
    }
    std::sort(corder.begin(), corder.end());
    std::vector<int32_t> tile_chunks, fresh, owned;
    std::vector<std::vector<int32_t>> groups;
    P2Fan fan;
    int64_t cursor = 0;
    while (cursor < n_chunks) {
      tile_chunks.clear();
      int n_local = 0, n_rows = 0;
      const int32_t tag = serial++;
      while (cursor < n_chunks && tile_chunks.size() < 4) {
        const int64_t c = corder[size_t(cursor)].second;
        fresh.clear();
        for (int64_t v = chunk_first[size_t(c)]; v < chunk_first[size_t(c) + 1]; ++v) {
          // local vertices of a vertex row: itself and its vertex neighbours (columns < n_verts)
          if (stamp[size_t(v)] != tag) {
            stamp[size_t(v)] = tag;
            fresh.push_back(int32_t(v));
          }
          for (int64_t p = rowptr[v]; p < rowptr[v + 1] && colind[p] < n_verts; ++p)
            if (stamp[size_t(colind[p])] != tag) {
              stamp[size_t(colind[p])] = tag;
              fresh.push_back(colind[p]);
            }
        }
        const int rows_after = n_rows + int(chunk_first[size_t(c) + 1] - chunk_first[size_t(c)]);
        if (n_local + int(fresh.size()) > kP2VertCap ||
            n_local + int(fresh.size()) - rows_after > kP2HaloCap) {
          for (int32_t w : fresh) stamp[size_t(w)] = -1;
          if (tile_chunks.empty())
            return fail(TFEM_ERR_UNSUPPORTED, "a chunk of the vertex numbering alone exceeds the tile capacity");
          break;
        }
        n_local += int(fresh.size());
        n_rows = rows_after;
        tile_chunks.push_back(int32_t(c));
        ++cursor;
      }
      std::sort(tile_chunks.begin(), tile_chunks.end());
      groups.push_back(tile_chunks);
    }
    // the tiles in the order of their first vertex: the tiles the workgroups work on at the same
    // time cover long contiguous ranges of the coordinate and CSR value arrays
    std::sort(groups.begin(), groups.end(),
              [](const std::vector<int32_t> &x, const std::vector<int32_t> &y) { return x[0] < y[0]; });
    for (const std::vector<int32_t> &group : groups) {
      tile_chunks = group;
      const int32_t tag = serial++;
      for (int32_t c : tile_chunks)
        for (int64_t v = chunk_first[size_t(c)]; v < chunk_first[size_t(c) + 1]; ++v) {
          stamp[size_t(v)] = tag;
          for (int64_t p = rowptr[v]; p < rowptr[v + 1] && colind[p] < n_verts; ++p) stamp[size_t(colind[p])] = tag;
        }
      owned.clear();
      int32_t ws[5];
      for (int w = 0; w < 5; ++w) {
        ws[w] = int32_t(owned.size());
        if (w < int(tile_chunks.size()))
          for (int32_t v = chunk_first[size_t(tile_chunks[size_t(w)])];
               v < chunk_first[size_t(tile_chunks[size_t(w)]) + 1]; ++v)
            owned.push_back(v);
      }
      const int n_own = int(owned.size());
      const int32_t vert_off = int32_t(plan.vert_gid[0].size());
      const int32_t row_off = int32_t(plan.rows[0].size() / 8);
      for (int l = 0; l < n_own; ++l) {
        local_id[size_t(owned[size_t(l)])] = l;
        stamp[size_t(owned[size_t(l)])] = -2 - tag;
      }
      plan.vert_gid[0].insert(plan.vert_gid[0].end(), owned.begin(), owned.end());
      int next_local = n_own;
      for (int l = 0; l < n_own; ++l) {
        const int32_t u = owned[size_t(l)];
        for (int64_t p = rowptr[u]; p < rowptr[u + 1] && colind[p] < n_verts; ++p) {
          const int32_t w = colind[p];
          if (stamp[size_t(w)] == tag) {
            stamp[size_t(w)] = -2 - tag;
            local_id[size_t(w)] = next_local++;
            plan.vert_gid[0].push_back(w);
          }
        }
      }
      for (int l = 0; l < n_own; ++l) {
        const int32_t u = owned[size_t(l)];
        if (!build_p2_fan(conn6, u, adj.data() + adj_ptr[size_t(u)], adj.data() + adj_ptr[size_t(u) + 1], fan))
          return fail(TFEM_ERR_UNSUPPORTED, "vertex %d: more than 15 neighbours or no ring form", u);
        int n_tri = 0;
        for (int i = 0; i < fan.k; ++i) n_tri += fan.flag[i] != 0;
        const int len = int(rowptr[u + 1] - rowptr[u]);
        if (len != (fan.k ? 1 + 2 * fan.k + n_tri : 0))
          return fail(TFEM_ERR_UNSUPPORTED, "row %d: %d entries for %d neighbours", u, len, fan.k);
        uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        bool ok = true;
        if (fan.k > 7) {
          // a long row: nothing for the tile's lane to do, the row goes to the long-row launch
          w[3] = 0x80000000u | uint32_t(len);
          uint32_t r[32] = {0};
          r[0] = uint32_t(u);
          r[1] = uint32_t(rowptr[u]);
          r[2] = uint32_t(fan.k) | pos_in_row(rowptr, colind, u, u, ok) << 8;
          uint32_t field[45] = {0};
          for (int i = 0; i < fan.k; ++i) {
            r[3] |= uint32_t(fan.flag[i]) << (2 * i);
            r[4 + i] = uint32_t(fan.nb[i]);
            field[i] = pos_in_row(rowptr, colind, u, fan.nb[i], ok);
            field[15 + i] = pos_in_row(rowptr, colind, u, fan.edge_v[i], ok);
            if (fan.flag[i]) field[30 + i] = pos_in_row(rowptr, colind, u, fan.edge_op[i], ok);
          }
          for (int f = 0; f < 45; ++f) r[19 + f / 5] |= field[f] << (6 * (f % 5));
          if (!ok) return fail(TFEM_ERR_INVALID_ARGUMENT, "row %d: a column is missing from the CSR pattern", u);
          plan.long_rows.insert(plan.long_rows.end(), r, r + 32);
          plan.rows[0].insert(plan.rows[0].end(), w, w + 8);
          uint32_t lc[16];
          for (int i = 0; i < 16; ++i) lc[i] = i < fan.k ? fan.code[i] : 0xFFFFFFFFu;
          plan.long_codes.insert(plan.long_codes.end(), lc, lc + 16);
          plan.codes[0].insert(plan.codes[0].end(), 8, 0xFFFFFFFFu);
          continue;
        }
        uint32_t field[21] = {0};
        for (int i = 0; i < fan.k; ++i) {
          w[i / 3] |= uint32_t(local_id[size_t(fan.nb[i])]) << (10 * (i % 3));
          w[2] |= uint32_t(fan.flag[i]) << (10 + 2 * i);
          field[i] = pos_in_row(rowptr, colind, u, fan.nb[i], ok);
          field[7 + i] = pos_in_row(rowptr, colind, u, fan.edge_v[i], ok);
          if (fan.flag[i]) field[14 + i] = pos_in_row(rowptr, colind, u, fan.edge_op[i], ok);
        }
        w[2] |= uint32_t(fan.k) << 24;
        if (len) w[2] |= pos_in_row(rowptr, colind, u, u, ok) << 27;
        if (!ok) return fail(TFEM_ERR_INVALID_ARGUMENT, "row %d: a column is missing from the CSR pattern", u);
        for (int f = 0; f < 21; ++f) w[3 + f / 6] |= field[f] << (5 * (f % 6));
        plan.rows[0].insert(plan.rows[0].end(), w, w + 8);
        uint32_t vc[8];
        for (int i = 0; i < 8; ++i) vc[i] = i < fan.k ? fan.code[i] : 0xFFFFFFFFu;
        plan.codes[0].insert(plan.codes[0].end(), vc, vc + 8);
      }
      int32_t d[16] = {vert_off, next_local, row_off, 0, ws[1], ws[2], ws[3], n_own, 0, 0, 0, 0, 0, 0, 0, 0};
      for (int w = 0; w < 4; ++w)
        if (ws[w] < ws[w + 1]) {
          d[8 + w] = owned[size_t(ws[w])];
          d[12 + w] = int32_t(rowptr[owned[size_t(ws[w])]]);
        }
      plan.desc[0].insert(plan.desc[0].end(), d, d + 16);
      plan.max_n_vert[0] = std::max(plan.max_n_vert[0], next_local);
      plan.max_n_halo = std::max(plan.max_n_halo, next_local - n_own);
      plan.n_tiles[0]++;
    }
    if (int64_t(plan.vert_gid[0].size()) > 2 * n_verts && n_verts > 4096)
      return fail(TFEM_ERR_UNSUPPORTED, "the vertex numbering has no locality");
  }

  // ---------------------------------------------------------------- edge-row tiles
  {
    const std::vector<int32_t> chunk_first = make_chunks(n_edges, [&](int64_t e) {
      const double dx = edge_mid[size_t(2 * e)] - edge_mid[size_t(2 * e) - 2];
      const double dy = edge_mid[size_t(2 * e) + 1] - edge_mid[size_t(2 * e) - 1];
      return dx * dx + dy * dy > jump2;
    });
    const int64_t n_chunks = int64_t(chunk_first.size()) - 1;
    std::vector<int32_t> fresh;
    int64_t cursor = 0;
    while (cursor < n_chunks) {
      // consecutive chunks of the edge numbering share a tile while their vertices fit
      const int32_t tag = serial++;
      int32_t ws[5] = {0, 0, 0, 0, 0};
      int n_chunk = 0, n_local = 0;
      const int32_t vert_off = int32_t(plan.vert_gid[1].size());
      while (cursor < n_chunks && n_chunk < 4) {
        fresh.clear();
        for (int64_t e = chunk_first[size_t(cursor)]; e < chunk_first[size_t(cursor) + 1]; ++e)
          for (int s = 0; s < 2; ++s) {
            const int32_t te = edge_elem[size_t(2 * e + s)];
            if (te < 0) continue;
            for (int a = 0; a < 3; ++a) {
              const int32_t w = conn6[6 * int64_t(te >> 2) + a];
              if (stamp[size_t(w)] != tag) {
                stamp[size_t(w)] = tag;
                fresh.push_back(w);
              }
            }
          }
        if (n_local + int(fresh.size()) > kP2VertCap) {
          for (int32_t w : fresh) stamp[size_t(w)] = -1;
          if (n_chunk == 0)
            return fail(TFEM_ERR_UNSUPPORTED, "a chunk of the edge numbering alone exceeds the tile capacity");
          break;
        }
        for (int32_t w : fresh) {
          local_id[size_t(w)] = n_local++;
          plan.vert_gid[1].push_back(w);
        }
        ++n_chunk;
        ws[n_chunk] = ws[n_chunk - 1] + (chunk_first[size_t(cursor) + 1] - chunk_first[size_t(cursor)]);
        ++cursor;
      }
      for (int w = n_chunk + 1; w < 5; ++w) ws[w] = ws[n_chunk];
      const int64_t e0 = chunk_first[size_t(cursor - n_chunk)];
      const int n_own = ws[4];
      const int32_t row_off = int32_t(plan.rows[1].size() / 4);
      for (int64_t e = e0; e < e0 + n_own; ++e) {
        const int64_t row = n_verts + e;
        uint32_t w[4] = {0, 0, 0, 0};
        bool ok = true;
        const int32_t te1 = edge_elem[size_t(2 * e)], te2 = edge_elem[size_t(2 * e) + 1];
        const int32_t *c1 = conn6 + 6 * int64_t(te1 >> 2);
        const int m1 = te1 & 3;
        const int32_t a = c1[m1], b = c1[(m1 + 1) % 3], c = c1[(m1 + 2) % 3];
        w[0] = uint32_t(local_id[size_t(a)]) | uint32_t(local_id[size_t(b)]) << 10 | uint32_t(local_id[size_t(c)]) << 20;
        uint32_t pos[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        pos[0] = pos_in_row(rowptr, colind, row, a, ok);
        pos[1] = pos_in_row(rowptr, colind, row, b, ok);
        pos[2] = pos_in_row(rowptr, colind, row, c, ok);
        pos[3] = pos_in_row(rowptr, colind, row, int32_t(row), ok);
        pos[4] = pos_in_row(rowptr, colind, row, c1[3 + (m1 + 1) % 3], ok);
        pos[5] = pos_in_row(rowptr, colind, row, c1[3 + (m1 + 2) % 3], ok);
        int len = 6;
        if (te2 >= 0) {
          const int32_t *c2 = conn6 + 6 * int64_t(te2 >> 2);
          const int m2 = te2 & 3;
          const int32_t a2 = c2[m2], b2 = c2[(m2 + 1) % 3], dd = c2[(m2 + 2) % 3];
          if (!((a2 == b && b2 == a) || (a2 == a && b2 == b)))
            return fail(TFEM_ERR_UNSUPPORTED, "edge DoF %lld joins different vertex pairs", (long long)row);
          w[1] = uint32_t(local_id[size_t(dd)]) | 1u << 10 | (a2 == b ? 1u << 11 : 0u);
          pos[6] = pos_in_row(rowptr, colind, row, dd, ok);
          pos[7] = pos_in_row(rowptr, colind, row, c2[3 + (m2 + 1) % 3], ok);
          pos[8] = pos_in_row(rowptr, colind, row, c2[3 + (m2 + 2) % 3], ok);
          len = 9;
        }
        if (!ok || rowptr[row + 1] - rowptr[row] != len)
          return fail(TFEM_ERR_UNSUPPORTED, "edge row %lld does not have %d entries", (long long)row, len);
        for (int f = 0; f < 8; ++f) w[2] |= pos[f] << (4 * f);
        w[3] = pos[8];
        plan.rows[1].insert(plan.rows[1].end(), w, w + 4);
        plan.codes[1].push_back(uint32_t(te1));
        plan.codes[1].push_back(te2 >= 0 ? uint32_t(te2) : 0xFFFFFFFFu);
      }
      int32_t d[16] = {vert_off, n_local, row_off, 0, ws[1], ws[2], ws[3], n_own, 0, 0, 0, 0, 0, 0, 0, 0};
      for (int w = 0; w < 4; ++w)
        if (ws[w] < ws[w + 1]) {
          d[8 + w] = int32_t(n_verts + e0 + ws[w]);
          d[12 + w] = int32_t(rowptr[n_verts + e0 + ws[w]]);
        }
      plan.desc[1].insert(plan.desc[1].end(), d, d + 16);
      plan.max_n_vert[1] = std::max(plan.max_n_vert[1], int32_t(n_local));
      plan.n_tiles[1]++;
    }
    if (int64_t(plan.vert_gid[1].size()) > 3 * n_edges && n_edges > 4096)
      return fail(TFEM_ERR_UNSUPPORTED, "the edge numbering has no locality");
  }
  return TFEM_OK;
}

void p2_layout(const P2Plan &p, int64_t layout[kP2LayoutLen]) {
  std::memset(layout, 0, sizeof(int64_t) * kP2LayoutLen);
  layout[0] = p.n_tiles[0];
  layout[1] = p.n_tiles[1];
  layout[2] = p.n_verts;
  layout[3] = p.n_edges;
  layout[4] = p.max_n_vert[0];
  layout[5] = p.max_n_vert[1];
  layout[6] = p.max_n_halo;
  layout[7] = int64_t(p.vert_gid[0].size());
  layout[8] = int64_t(p.vert_gid[1].size());
  const int64_t bytes[10] = {int64_t(p.desc[0].size()) * 4, int64_t(p.rows[0].size()) * 4,
                             int64_t(p.vert_gid[0].size()) * 4, int64_t(p.desc[1].size()) * 4,
                             int64_t(p.rows[1].size()) * 4, int64_t(p.vert_gid[1].size()) * 4,
                             int64_t(p.long_rows.size()) * 4, int64_t(p.codes[0].size()) * 4,
                             int64_t(p.codes[1].size()) * 4, int64_t(p.long_codes.size()) * 4};
  const int slot_of[10] = {10, 11, 12, 13, 14, 15, 17, 19, 20, 21};
  int64_t off = 0;
  for (int i = 0; i < 10; ++i) {
    layout[slot_of[i]] = off;
    off += (bytes[i] + 15) & ~int64_t(15);
  }
  layout[16] = off + 64;
  layout[18] = int64_t(p.long_rows.size() / 32);
}

}  // namespace
}  // namespace tfem

extern "C" {

int tfem_p2_plan_create(const int32_t *conn_dof_host, int64_t n_elems, int64_t n_verts,
                        int64_t n_dofs, const double *coords_host, const int64_t *rowptr_host,
                        const int32_t *colind_host, void **plan_out) {
  using namespace tfem;
  if (!plan_out) return fail(TFEM_ERR_INVALID_ARGUMENT, "plan_out is NULL");
  *plan_out = nullptr;
  if (n_elems < 0 || n_verts < 0 || n_dofs < n_verts || (n_elems > 0 && !conn_dof_host) ||
      (n_verts > 0 && !coords_host) || !rowptr_host || (rowptr_host[n_dofs] > 0 && !colind_host))
    return fail(TFEM_ERR_INVALID_ARGUMENT, "bad arguments");
  if (6 * n_elems >= (int64_t(1) << 31) || rowptr_host[n_dofs] >= (int64_t(1) << 31))
    return fail(TFEM_ERR_INDEX_RANGE, "mesh too large for the int32 P2 row plan");
  auto *plan = new P2Plan();
  const int st = build_p2(conn_dof_host, n_elems, n_verts, n_dofs, coords_host, rowptr_host, colind_host, *plan);
  if (st != TFEM_OK) {
    delete plan;
    return st;
  }
  *plan_out = plan;
  return TFEM_OK;
}

int tfem_p2_plan_sizes(const void *plan_handle, int64_t layout[24]) {
  using namespace tfem;
  if (!plan_handle || !layout) return fail(TFEM_ERR_INVALID_ARGUMENT, "NULL pointer");
  p2_layout(*static_cast<const P2Plan *>(plan_handle), layout);
  return TFEM_OK;
}

int tfem_p2_plan_pack(const void *plan_handle, void *blob_host) {
  using namespace tfem;
  if (!plan_handle || !blob_host) return fail(TFEM_ERR_INVALID_ARGUMENT, "NULL pointer");
  const auto *p = static_cast<const P2Plan *>(plan_handle);
  int64_t layout[kP2LayoutLen];
  p2_layout(*p, layout);
  auto *out = static_cast<unsigned char *>(blob_host);
  std::memset(out, 0, size_t(layout[16]));
  std::memcpy(out + layout[10], p->desc[0].data(), p->desc[0].size() * 4);
  std::memcpy(out + layout[11], p->rows[0].data(), p->rows[0].size() * 4);
  std::memcpy(out + layout[12], p->vert_gid[0].data(), p->vert_gid[0].size() * 4);
  std::memcpy(out + layout[13], p->desc[1].data(), p->desc[1].size() * 4);
  std::memcpy(out + layout[14], p->rows[1].data(), p->rows[1].size() * 4);
  std::memcpy(out + layout[15], p->vert_gid[1].data(), p->vert_gid[1].size() * 4);
  std::memcpy(out + layout[17], p->long_rows.data(), p->long_rows.size() * 4);
  std::memcpy(out + layout[19], p->codes[0].data(), p->codes[0].size() * 4);
  std::memcpy(out + layout[20], p->codes[1].data(), p->codes[1].size() * 4);
  std::memcpy(out + layout[21], p->long_codes.data(), p->long_codes.size() * 4);
  return TFEM_OK;
}

void tfem_p2_plan_destroy(void *plan_handle) { delete static_cast<tfem::P2Plan *>(plan_handle); }

}  // extern "C"
