// Tile plan for the P1 headline path (host, once per mesh).
//
// The CSR rows (= vertices) are cut into spatially compact tiles along a Z-order curve.
// A tile OWNS its rows: it processes every element incident to an owned vertex (elements
// on a tile border are processed by up to three tiles), accumulates the owned rows in LDS
// and writes each CSR value exactly once with plain stores -- no global atomics, no
// zero-fill of the value array, no second pass.  Everything the kernel needs per element
// is one 12-byte record: the three tile-local vertex ids (pre-scaled to LDS byte offsets)
// and, for each of the three rows, the positions of the three columns inside that row.
#include <algorithm>
#include <cstring>
#include <numeric>
#include <vector>

#include "tfem_common.hpp"

namespace tfem {

constexpr int kDescStride = 12;

struct TilePlan {
  // limits the kernel was compiled for
  int elem_cap = 0, vert_cap = 0, acc_cap = 0;
  // per tile (kDescStride ints): elem_off, n_elem, vert_off, n_vert, n_own, acc_size,
  // loff_off, run_off, n_runs, lrun_off, 0, 0
  std::vector<int32_t> desc;
  // Element records.  12-byte form (rec_words = 3): word j = 16 * local id of vertex j |
  // 4-bit positions << 16.  8-byte form (rec_words = 2), used when no row has more than 8
  // entries and tiles hold at most 1022 local vertices: word 0 = three 10-bit local ids,
  // word 1 = nine 3-bit positions (row j, column i at bit 3 * (3 j + i)).
  int rec_words = 3;
  std::vector<uint32_t> records;
  std::vector<int32_t> elem_id;     // original element id of every record (load vector path)
  std::vector<int32_t> vert_gid;    // global vertex id of every tile-local vertex
  std::vector<uint16_t> row_loff;   // accumulator offset of every owned row
  // Output runs: maximal groups of owned rows that are contiguous in the CSR value array.
  // Accumulator entry s of run r goes to vals[s + run_delta[r]]; run r covers entries
  // [run_lstart[r], run_lstart[r+1]) (one sentinel per tile).
  std::vector<int32_t> run_delta;
  std::vector<uint16_t> run_lstart;
  int32_t max_n_elem = 0, max_n_vert = 0, max_n_own = 0, max_acc = 0, max_row_len = 0;
  int32_t max_n_runs = 0;
  int64_t n_tiles = 0;
};

namespace {

inline uint64_t spread_bits(uint64_t x) {
  x &= 0xFFFFFFFFull;
  x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
  x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
  x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
  x = (x | (x << 2)) & 0x3333333333333333ull;
  x = (x | (x << 1)) & 0x5555555555555555ull;
  return x;
}

template <typename I>
int build(const I *conn, int64_t n_elems, int64_t n_verts, const double *coords,
          const int64_t *rowptr, const int32_t *colind, int elem_cap, int vert_cap, int acc_cap,
          int own_cap, TilePlan &plan) {
  {
    int64_t longest = 0;
    for (int64_t v = 0; v < n_verts; ++v) longest = std::max(longest, rowptr[v + 1] - rowptr[v]);
    if (longest <= 8) {
      plan.rec_words = 2;
      vert_cap = std::min(vert_cap, 1022);
    }
  }
  plan.elem_cap = elem_cap;
  plan.vert_cap = vert_cap;
  plan.acc_cap = acc_cap;
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
    order[size_t(v)] = {spread_bits(qx) | (spread_bits(qy) << 1), int32_t(v)};
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
  // ---- greedy tiling along the curve ------------------------------------------------------
  std::vector<int32_t> elem_stamp(size_t(n_elems), -1), vert_stamp(size_t(n_verts), -1);
  std::vector<int32_t> vert_local(size_t(n_verts), 0);
  std::vector<int32_t> owned, tile_elems, halo, new_elems, new_verts;
  int64_t cursor = 0;
  int32_t tile = 0;
  while (cursor < n_verts) {
    owned.clear();
    tile_elems.clear();
    halo.clear();
    int acc = 0;
    int n_local = 0;  // vertices referenced so far (owned or not)
    // phase A: choose the owned vertices of this tile
    while (cursor < n_verts) {
      const int32_t u = order[size_t(cursor)].second;
      const int len = int(rowptr[u + 1] - rowptr[u]);
      if (len > plan.max_row_len) plan.max_row_len = len;
      if (len > 16)
        return fail(TFEM_ERR_UNSUPPORTED, "row of vertex %d has %d entries (> 16)", u, len);
      new_elems.clear();
      new_verts.clear();
      for (int64_t k = adj_ptr[u]; k < adj_ptr[u + 1]; ++k) {
        const int32_t e = adj[size_t(k)];
        if (elem_stamp[size_t(e)] == tile) continue;
        elem_stamp[size_t(e)] = tile;
        new_elems.push_back(e);
        for (int a = 0; a < 3; ++a) {
          const int32_t w = int32_t(conn[3 * int64_t(e) + a]);
          if (vert_stamp[size_t(w)] != tile) {
            vert_stamp[size_t(w)] = tile;
            new_verts.push_back(w);
          }
        }
      }
      int extra_local = int(new_verts.size());
      if (vert_stamp[size_t(u)] != tile) {  // isolated vertex: still owns its (empty) row
        vert_stamp[size_t(u)] = tile;
        new_verts.push_back(u);
        extra_local++;
      }
      const bool fits = int(tile_elems.size() + new_elems.size()) <= elem_cap &&
                        n_local + extra_local <= vert_cap && acc + len <= acc_cap &&
                        int(owned.size()) + 1 <= own_cap;
      if (!fits && !owned.empty()) {  // roll back and close the tile
        for (int32_t e : new_elems) elem_stamp[size_t(e)] = -1;
        for (int32_t w : new_verts) vert_stamp[size_t(w)] = -1;
        break;
      }
      if (!fits)
        return fail(TFEM_ERR_UNSUPPORTED, "vertex %d alone exceeds the tile capacity", u);
      owned.push_back(u);
      tile_elems.insert(tile_elems.end(), new_elems.begin(), new_elems.end());
      n_local += extra_local;
      acc += len;
      ++cursor;
    }
    // phase B: local numbering -- owned rows first, ascending global id (contiguous output runs)
    std::sort(owned.begin(), owned.end());
    const int n_own = int(owned.size());
    for (int l = 0; l < n_own; ++l) {
      vert_local[size_t(owned[size_t(l)])] = l;
      vert_stamp[size_t(owned[size_t(l)])] = -2 - tile;  // marks "owned by this tile"
    }
    int next_local = n_own;
    const int32_t vert_off = int32_t(plan.vert_gid.size());
    plan.vert_gid.insert(plan.vert_gid.end(), owned.begin(), owned.end());
    std::sort(tile_elems.begin(), tile_elems.end());
    for (int32_t e : tile_elems)
      for (int a = 0; a < 3; ++a) {
        const int32_t w = int32_t(conn[3 * int64_t(e) + a]);
        if (vert_stamp[size_t(w)] == tile) {  // referenced, not owned, not numbered yet
          vert_stamp[size_t(w)] = -1;         // numbered halo (stamp reset; vert_local valid)
          vert_local[size_t(w)] = next_local++;
          halo.push_back(w);
          plan.vert_gid.push_back(w);
        }
      }
    // phase C: rows and output runs
    const int32_t loff_off = int32_t(plan.row_loff.size());
    const int32_t run_off = int32_t(plan.run_delta.size());
    const int32_t lrun_off = int32_t(plan.run_lstart.size());
    int run = 0;
    int64_t next_global = -1;
    for (int l = 0; l < n_own; ++l) {
      const int32_t g = owned[size_t(l)];
      const int len = int(rowptr[g + 1] - rowptr[g]);
      plan.row_loff.push_back(uint16_t(run));
      if (len > 0 && rowptr[g] != next_global) {  // a new run starts at this row
        plan.run_lstart.push_back(uint16_t(run));
        plan.run_delta.push_back(int32_t(rowptr[g] - run));
      }
      if (len > 0) next_global = rowptr[g + 1];
      run += len;
    }
    plan.run_lstart.push_back(uint16_t(run));
    const int32_t n_runs = int32_t(plan.run_delta.size()) - run_off;
    // phase D: element records
    const int32_t elem_off = int32_t(plan.records.size() / size_t(plan.rec_words));
    for (int32_t e : tile_elems) {
      const I *c = conn + 3 * int64_t(e);
      uint32_t pos[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};  // [row j][column i], owned rows only
      for (int j = 0; j < 3; ++j) {
        const int32_t row = int32_t(c[j]);
        if (vert_stamp[size_t(row)] != -2 - tile) continue;  // row not owned here
        const int32_t *first = colind + rowptr[row];
        const int32_t *last = colind + rowptr[row + 1];
        for (int i = 0; i < 3; ++i)
          pos[j][i] = uint32_t(std::lower_bound(first, last, int32_t(c[i])) - first);
      }
      uint32_t word[3] = {0, 0, 0};
      if (plan.rec_words == 3) {
        for (int j = 0; j < 3; ++j) {
          word[j] = uint32_t(vert_local[size_t(c[j])]) << 4;
          for (int i = 0; i < 3; ++i) word[j] |= pos[j][i] << (16 + 4 * i);
        }
      } else {
        for (int j = 0; j < 3; ++j) {
          word[0] |= uint32_t(vert_local[size_t(c[j])]) << (10 * j);
          for (int i = 0; i < 3; ++i) word[1] |= pos[j][i] << (3 * (3 * j + i));
        }
      }
      plan.records.insert(plan.records.end(), word, word + plan.rec_words);
      plan.elem_id.push_back(e);
    }
    // un-own (so a later tile that references these vertices as halo numbers them afresh)
    for (int32_t g : owned) vert_stamp[size_t(g)] = -1;
    const int32_t d[kDescStride] = {elem_off, int32_t(tile_elems.size()), vert_off, next_local,
                                    n_own, run, loff_off, run_off, n_runs, lrun_off, 0, 0};
    plan.desc.insert(plan.desc.end(), d, d + kDescStride);
    plan.max_n_elem = std::max(plan.max_n_elem, d[1]);
    plan.max_n_vert = std::max(plan.max_n_vert, d[3]);
    plan.max_n_own = std::max(plan.max_n_own, d[4]);
    plan.max_acc = std::max(plan.max_acc, d[5]);
    plan.max_n_runs = std::max(plan.max_n_runs, n_runs);
    ++tile;
    // stamps of this tile's elements must not collide with the next tile id: they hold `tile-1`
  }
  plan.n_tiles = tile;
  return TFEM_OK;
}

}  // namespace
}  // namespace tfem

extern "C" {

int tfem_tile_plan_create(const void *conn_host, int idx_bytes, int64_t n_elems, int64_t n_verts,
                          const double *coords_host, const int64_t *rowptr_host,
                          const int32_t *colind_host, int elem_cap, int vert_cap, int acc_cap,
                          int own_cap, void **plan_out) {
  using namespace tfem;
  if (!plan_out) return fail(TFEM_ERR_INVALID_ARGUMENT, "plan_out is NULL");
  *plan_out = nullptr;
  if (idx_bytes != 4 && idx_bytes != 8)
    return fail(TFEM_ERR_INVALID_ARGUMENT, "idx_bytes must be 4 or 8");
  if (n_elems < 0 || n_verts < 0 || (n_elems > 0 && !conn_host) || (n_verts > 0 && !coords_host) ||
      !rowptr_host)
    return fail(TFEM_ERR_INVALID_ARGUMENT, "bad arguments");
  if (elem_cap < 1 || vert_cap < 3 || vert_cap > 4096 || acc_cap < 16 || acc_cap > 65535 ||
      own_cap < 1)
    return fail(TFEM_ERR_INVALID_ARGUMENT, "bad tile capacities");
  if (3 * n_elems >= (int64_t(1) << 31) || rowptr_host[n_verts] >= (int64_t(1) << 31))
    return fail(TFEM_ERR_INDEX_RANGE, "mesh too large for the int32 tile plan");
  auto *plan = new TilePlan();
  int st;
  if (idx_bytes == 4)
    st = build(static_cast<const int32_t *>(conn_host), n_elems, n_verts, coords_host, rowptr_host,
               colind_host, elem_cap, vert_cap, acc_cap, own_cap, *plan);
  else
    st = build(static_cast<const int64_t *>(conn_host), n_elems, n_verts, coords_host, rowptr_host,
               colind_host, elem_cap, vert_cap, acc_cap, own_cap, *plan);
  if (st != TFEM_OK) {
    delete plan;
    return st;
  }
  *plan_out = plan;
  return TFEM_OK;
}

// Packed form: ONE blob holding desc | records | vert_gid | row_loff | run_delta | run_lstart |
// elem_id, each 16-byte aligned, with slack behind the blob so that lanes of the last tile
// that run past an array stay inside it.  layout[0..11] = sizes, [12..18] = byte offsets of
// the seven arrays in that order, [19] = total bytes, [20] = words per element record.
static void plan_layout(const tfem::TilePlan &p, int64_t layout[24]) {
  const int64_t sizes[12] = {p.n_tiles, int64_t(p.records.size() / size_t(p.rec_words)),
                             int64_t(p.vert_gid.size()),
                             int64_t(p.row_loff.size()), int64_t(p.run_delta.size()),
                             p.max_n_elem, p.max_n_vert, p.max_n_own, p.max_acc, p.max_row_len,
                             p.max_n_runs, int64_t(p.run_lstart.size())};
  for (int i = 0; i < 12; ++i) layout[i] = sizes[i];
  const int64_t bytes[7] = {int64_t(p.desc.size()) * 4, int64_t(p.records.size()) * 4,
                            int64_t(p.vert_gid.size()) * 4, int64_t(p.row_loff.size()) * 2,
                            int64_t(p.run_delta.size()) * 4, int64_t(p.run_lstart.size()) * 2,
                            int64_t(p.elem_id.size()) * 4};
  int64_t off = 0;
  for (int i = 0; i < 7; ++i) {
    layout[12 + i] = off;
    off += (bytes[i] + 15) & ~int64_t(15);
  }
  layout[19] = off + 64;
  layout[20] = p.rec_words;
  layout[21] = layout[22] = layout[23] = 0;
}

int tfem_tile_plan_sizes(const void *plan_handle, int64_t layout[24]) {
  using namespace tfem;
  if (!plan_handle || !layout) return fail(TFEM_ERR_INVALID_ARGUMENT, "NULL pointer");
  plan_layout(*static_cast<const TilePlan *>(plan_handle), layout);
  return TFEM_OK;
}

int tfem_tile_plan_pack(const void *plan_handle, void *blob_host) {
  using namespace tfem;
  if (!plan_handle || !blob_host) return fail(TFEM_ERR_INVALID_ARGUMENT, "NULL pointer");
  const auto *p = static_cast<const TilePlan *>(plan_handle);
  int64_t layout[24];
  plan_layout(*p, layout);
  auto *out = static_cast<unsigned char *>(blob_host);
  std::memset(out, 0, size_t(layout[19]));
  std::memcpy(out + layout[12], p->desc.data(), p->desc.size() * 4);
  std::memcpy(out + layout[13], p->records.data(), p->records.size() * 4);
  std::memcpy(out + layout[14], p->vert_gid.data(), p->vert_gid.size() * 4);
  std::memcpy(out + layout[15], p->row_loff.data(), p->row_loff.size() * 2);
  std::memcpy(out + layout[16], p->run_delta.data(), p->run_delta.size() * 4);
  std::memcpy(out + layout[17], p->run_lstart.data(), p->run_lstart.size() * 2);
  std::memcpy(out + layout[18], p->elem_id.data(), p->elem_id.size() * 4);
  return TFEM_OK;
}

void tfem_tile_plan_destroy(void *plan_handle) { delete static_cast<tfem::TilePlan *>(plan_handle); }

}  // extern "C"
