// Host side of libtfem_hip: status/error plumbing, reference-element tables and the
// symbolic (CSR pattern + slot map) phase.  No device code in this file.
#include <algorithm>
#include <cstring>
#include <numeric>
#include <vector>

#include <chrono>
#include <memory>
#include <cstdio>
#include <cstdlib>

#include "tfem_common.hpp"
#include "tfem_threads.hpp"

namespace tfem {

static thread_local char g_last_error[512] = "";

int fail(int status, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
  va_end(ap);
  return status;
}

// Literals of the reference, element_tri.py:77-130 (15-digit truncations included).
bool triangle_rule(int quad_order, int *nq, double nodes[kMaxQuad][2], double weights[kMaxQuad]) {
  switch (quad_order) {
    case 1:
      *nq = 1;
      nodes[0][0] = 1.0 / 3;
      nodes[0][1] = 1.0 / 3;
      weights[0] = 1.0;
      return true;
    case 2: {
      *nq = 3;
      const double n[3][2] = {{1.0 / 6, 1.0 / 6}, {2.0 / 3, 1.0 / 6}, {1.0 / 6, 2.0 / 3}};
      for (int q = 0; q < 3; ++q) {
        nodes[q][0] = n[q][0];
        nodes[q][1] = n[q][1];
        weights[q] = 1.0 / 3;
      }
      return true;
    }
    case 3: {
      *nq = 4;
      const double n[4][2] = {{1.0 / 3, 1.0 / 3}, {0.6, 0.2}, {0.2, 0.6}, {0.2, 0.2}};
      const double w[4] = {-9.0 / 16, 25.0 / 48, 25.0 / 48, 25.0 / 48};
      for (int q = 0; q < 4; ++q) {
        nodes[q][0] = n[q][0];
        nodes[q][1] = n[q][1];
        weights[q] = w[q];
      }
      return true;
    }
    case 4: {
      *nq = 6;
      const double n[6][2] = {{0.816847572980459, 0.091576213509771},
                              {0.091576213509771, 0.816847572980459},
                              {0.091576213509771, 0.091576213509771},
                              {0.108103018168070, 0.445948490915965},
                              {0.445948490915965, 0.108103018168070},
                              {0.445948490915965, 0.445948490915965}};
      for (int q = 0; q < 6; ++q) {
        nodes[q][0] = n[q][0];
        nodes[q][1] = n[q][1];
        weights[q] = q < 3 ? 0.109951743655322 : 0.223381589678011;
      }
      return true;
    }
    default:
      return false;
  }
}

template <typename T>
static void fill_tables(int nq, const double nodes[kMaxQuad][2], const double weights[kMaxQuad],
                        TriTables *t) {
  // barycentric_grad, element_tri.py:10-12
  const T g[3][2] = {{T(-1), T(-1)}, {T(1), T(0)}, {T(0), T(1)}};
  t->nq = nq;
  for (int q = 0; q < nq; ++q) {
    const T xi = T(nodes[q][0]), eta = T(nodes[q][1]);
    const T w = T(weights[q]);
    t->hw[q] = double(T(0.5) * w);  // reference_element_area * gaussian_weights
    // element_tri.py:23-26
    const T l[3] = {T(T(1.0) - xi) - eta, xi, eta};
    for (int i = 0; i < 3; ++i) t->lam[q][i] = double(l[i]);
    // element_tri.py:45-55
    t->phi2[q][0] = double(l[0] * (T(2) * l[0] - T(1)));
    t->phi2[q][1] = double(l[1] * (T(2) * l[1] - T(1)));
    t->phi2[q][2] = double(l[2] * (T(2) * l[2] - T(1)));
    t->phi2[q][3] = double(T(4) * l[0] * l[1]);
    t->phi2[q][4] = double(T(4) * l[1] * l[2]);
    t->phi2[q][5] = double(T(4) * l[2] * l[0]);
    // element_tri.py:57-68
    for (int c = 0; c < 2; ++c) {
      t->rgrad2[q][0][c] = double((T(4) * l[0] - T(1)) * g[0][c]);
      t->rgrad2[q][1][c] = double((T(4) * l[1] - T(1)) * g[1][c]);
      t->rgrad2[q][2][c] = double((T(4) * l[2] - T(1)) * g[2][c]);
      t->rgrad2[q][3][c] = double(T(4) * (l[1] * g[0][c] + l[0] * g[1][c]));
      t->rgrad2[q][4][c] = double(T(4) * (l[2] * g[1][c] + l[1] * g[2][c]));
      t->rgrad2[q][5][c] = double(T(4) * (l[0] * g[2][c] + l[2] * g[0][c]));
    }
  }
}

bool build_tri_tables(int quad_order, int real_bytes, TriTables *out) {
  int nq = 0;
  double nodes[kMaxQuad][2], weights[kMaxQuad];
  if (!triangle_rule(quad_order, &nq, nodes, weights)) return false;
  std::memset(out, 0, sizeof(*out));
  if (real_bytes == 4)
    fill_tables<float>(nq, nodes, weights, out);
  else
    fill_tables<double>(nq, nodes, weights, out);
  return true;
}

// ---------------------------------------------------------------------------------
// symbolic phase
// ---------------------------------------------------------------------------------
namespace {

template <typename I>
int check_conn(const I *conn, int64_t count, int64_t n_dofs) {
  std::vector<int64_t> bad(size_t(host_threads()) + 1, -1);
  parallel_for(count, [&](int64_t b, int64_t e, int t) {
    for (int64_t k = b; k < e; ++k)
      if (conn[k] < 0 || int64_t(conn[k]) >= n_dofs) {
        bad[size_t(t)] = k;
        return;
      }
  }, 1 << 16);
  for (int64_t k : bad)  // pieces are ascending: the first hit is the smallest index
    if (k >= 0)
      return fail(TFEM_ERR_INDEX_RANGE, "connectivity entry %lld = %lld outside [0, %lld)",
                  (long long)k, (long long)conn[k], (long long)n_dofs);
  return TFEM_OK;
}

// DoF -> incident elements (CSR form; the order inside a DoF's list is the order the threads
// arrived in -- callers that need it ascending sort the short lists).  Bucket sizes and bucket
// filling with relaxed atomic counters.
template <typename I>
void build_incidence(const I *conn, int64_t n_elems, int n, int64_t n_dofs, std::vector<int64_t> &ptr,
                     std::unique_ptr<int32_t[]> &elems) {
  std::vector<int64_t> count(size_t(n_dofs) + 1, 0);
  parallel_for(n_elems * n, [&](int64_t b, int64_t e, int) {
    for (int64_t k = b; k < e; ++k) __atomic_fetch_add(&count[size_t(conn[k]) + 1], int64_t(1), __ATOMIC_RELAXED);
  }, 1 << 16);
  ptr.assign(size_t(n_dofs) + 1, 0);
  std::partial_sum(count.begin(), count.end(), ptr.begin());
  elems.reset(new int32_t[size_t(std::max<int64_t>(ptr[size_t(n_dofs)], 1))]);  // first touched by the threads below
  std::vector<int64_t> &cursor = count;
  std::copy(ptr.begin(), ptr.end() - 1, cursor.begin());
  parallel_for(n_elems, [&](int64_t b, int64_t e, int) {
    for (int64_t el = b; el < e; ++el)
      for (int a = 0; a < n; ++a)
        elems[size_t(__atomic_fetch_add(&cursor[size_t(conn[el * n + a])], int64_t(1), __ATOMIC_RELAXED))] = int32_t(el);
  }, 1 << 14);
}

// Sorted unique columns of row r: the DoFs of every element that contains DoF r (the pattern
// is symmetric, so the transposed scatter convention does not change it).  `scratch` holds at
// least n * (incident elements) entries; returns the row length.
template <typename I>
inline int row_columns(const I *conn, int n, const int32_t *first_elem, const int32_t *last_elem,
                       std::vector<int32_t> &scratch) {
  scratch.clear();
  for (const int32_t *e = first_elem; e != last_elem; ++e) {
    const I *c = conn + int64_t(*e) * n;
    for (int k = 0; k < n; ++k) scratch.push_back(int32_t(c[k]));
  }
  std::sort(scratch.begin(), scratch.end());
  return int(std::unique(scratch.begin(), scratch.end()) - scratch.begin());
}

// The CSR pattern between the calls of the handle interface (tfem_csr_pattern_*): incidence and
// row pointers; the columns are written straight into the caller's array by export.
struct CsrPattern {
  std::vector<int64_t> inc_ptr, rowptr;
  std::unique_ptr<int32_t[]> inc;
  int64_t n_dofs = 0, nnz = 0, n_elems = 0;
  int n = 0;
};

template <typename I>
void pattern_rows(const I *conn, int64_t n_elems, int n, int64_t n_dofs, CsrPattern &p) {
  const bool timing = std::getenv("TFEM_PLAN_TIMING") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (!timing) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[pattern]   %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };
  p.n_dofs = n_dofs;
  p.n_elems = n_elems;
  p.n = n;
  build_incidence(conn, n_elems, n, n_dofs, p.inc_ptr, p.inc);
  lap("DoF -> elements");
  p.rowptr.assign(size_t(n_dofs) + 1, 0);
  parallel_for(n_dofs, [&](int64_t b, int64_t e, int) {
    std::vector<int32_t> scratch;
    for (int64_t r = b; r < e; ++r)
      p.rowptr[size_t(r) + 1] = row_columns(conn, n, p.inc.get() + p.inc_ptr[size_t(r)], p.inc.get() + p.inc_ptr[size_t(r) + 1], scratch);
  }, 1 << 12);
  for (int64_t r = 0; r < n_dofs; ++r) p.rowptr[size_t(r) + 1] += p.rowptr[size_t(r)];
  p.nnz = p.rowptr[size_t(n_dofs)];
  lap("row lengths");
}

template <typename I>
void pattern_columns(const I *conn, const CsrPattern &p, const int64_t *rowptr, int32_t *colind) {
  parallel_for(p.n_dofs, [&](int64_t b, int64_t e, int) {
    std::vector<int32_t> scratch;
    for (int64_t r = b; r < e; ++r) {
      const int len = row_columns(conn, p.n, p.inc.get() + p.inc_ptr[size_t(r)], p.inc.get() + p.inc_ptr[size_t(r) + 1], scratch);
      std::copy_n(scratch.data(), len, colind + rowptr[r]);
    }
  }, 1 << 12);
}

// slots[e][i][j] -> position of (row conn[j], col conn[i]): basis.py:73-76 builds rows_idx
// by tiling conn and cols_idx by repeating it, the value is local.reshape(-1).
template <typename I>
void fill_slots(const I *conn, int64_t n_elems, int n, const int64_t *rowptr, const int32_t *colind,
                int32_t *slots) {
  parallel_for(n_elems, [&](int64_t b, int64_t e, int) {
    for (int64_t el = b; el < e; ++el) {
      const I *c = conn + el * n;
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
          const int64_t row = int64_t(c[j]);
          const int32_t col = int32_t(c[i]);
          const int32_t *first = colind + rowptr[row];
          const int32_t *last = colind + rowptr[row + 1];
          slots[(el * n + i) * n + j] = int32_t(std::lower_bound(first, last, col) - colind);
        }
    }
  }, 1 << 12);
}

template <typename I>
int symbolic_count(const I *conn, int64_t n_elems, int n, int64_t n_dofs, int64_t *rowptr,
                   int64_t *nnz) {
  if (int st = check_conn(conn, n_elems * n, n_dofs)) return st;
  CsrPattern p;
  pattern_rows(conn, n_elems, n, n_dofs, p);
  std::copy(p.rowptr.begin(), p.rowptr.end(), rowptr);
  *nnz = p.nnz;
  if (*nnz >= (int64_t(1) << 31))
    return fail(TFEM_ERR_INDEX_RANGE, "nnz = %lld does not fit the int32 slot map",
                (long long)*nnz);
  return TFEM_OK;
}

template <typename I>
int symbolic_fill(const I *conn, int64_t n_elems, int n, int64_t n_dofs, const int64_t *rowptr,
                  int32_t *colind, int32_t *slots) {
  if (int st = check_conn(conn, n_elems * n, n_dofs)) return st;
  CsrPattern p;
  pattern_rows(conn, n_elems, n, n_dofs, p);
  for (int64_t r = 0; r <= n_dofs; ++r)
    if (rowptr[r] != p.rowptr[size_t(r)])
      return fail(TFEM_ERR_INVALID_ARGUMENT, "rowptr does not belong to this connectivity");
  pattern_columns(conn, p, rowptr, colind);
  fill_slots(conn, n_elems, n, rowptr, colind, slots);
  return TFEM_OK;
}

// The handle keeps a pointer to the caller's connectivity: it must stay alive until export.
struct PatternHandle {
  CsrPattern p;
  const void *conn = nullptr;
  int idx_bytes = 4;
};

template <typename I>
int pattern_create(const I *conn, int64_t n_elems, int n, int64_t n_dofs, PatternHandle **out) {
  if (int st = check_conn(conn, n_elems * n, n_dofs)) return st;
  auto *h = new PatternHandle();
  h->conn = conn;
  h->idx_bytes = int(sizeof(I));
  pattern_rows(conn, n_elems, n, n_dofs, h->p);
  if (h->p.nnz >= (int64_t(1) << 31)) {
    const long long nnz = h->p.nnz;
    delete h;
    return fail(TFEM_ERR_INDEX_RANGE, "nnz = %lld does not fit the int32 slot map", nnz);
  }
  *out = h;
  return TFEM_OK;
}

int check_symbolic_args(const void *conn, int idx_bytes, int64_t n_elems, int n_local,
                        int64_t n_dofs) {
  if (!conn && n_elems > 0) return fail(TFEM_ERR_INVALID_ARGUMENT, "conn is NULL");
  if (idx_bytes != 4 && idx_bytes != 8)
    return fail(TFEM_ERR_INVALID_ARGUMENT, "idx_bytes must be 4 or 8, got %d", idx_bytes);
  if (n_elems < 0 || n_dofs < 0 || n_local < 1 || n_local > 16)
    return fail(TFEM_ERR_INVALID_ARGUMENT, "bad sizes n_elems=%lld n_local=%d n_dofs=%lld",
                (long long)n_elems, n_local, (long long)n_dofs);
  if (n_dofs >= (int64_t(1) << 31))
    return fail(TFEM_ERR_INDEX_RANGE, "n_dofs = %lld does not fit int32 column indices",
                (long long)n_dofs);
  return TFEM_OK;
}

}  // namespace

bool pattern_view(void *pattern_handle, PatternView *out) {
  if (!pattern_handle || !out) return false;
  auto *h = static_cast<PatternHandle *>(pattern_handle);
  *out = PatternView{h->conn, h->idx_bytes, h->p.n, h->p.n_elems, h->p.n_dofs, h->p.nnz,
                     h->p.inc_ptr.data(), h->p.inc.get(), h->p.rowptr.data()};
  return true;
}

}  // namespace tfem

extern "C" {

int tfem_abi_version(void) { return TFEM_ABI_VERSION; }

const char *tfem_status_string(int status) {
  switch (status) {
    case TFEM_OK: return "ok";
    case TFEM_ERR_INVALID_ARGUMENT: return "invalid argument";
    case TFEM_ERR_UNSUPPORTED: return "not implemented";
    case TFEM_ERR_HIP: return "HIP runtime error";
    case TFEM_ERR_INDEX_RANGE: return "index out of range";
    case TFEM_ERR_NO_DEVICE: return "no HIP device";
    default: return "unknown status";
  }
}

const char *tfem_last_error(void) { return tfem::g_last_error; }

int tfem_quadrature_size(int quad_order) {
  int nq = 0;
  double nodes[tfem::kMaxQuad][2], weights[tfem::kMaxQuad];
  return tfem::triangle_rule(quad_order, &nq, nodes, weights) ? nq : 0;
}

int tfem_quadrature_rule(int quad_order, double *nodes_host, double *weights_host) {
  int nq = 0;
  double nodes[tfem::kMaxQuad][2], weights[tfem::kMaxQuad];
  if (!tfem::triangle_rule(quad_order, &nq, nodes, weights))
    return tfem::fail(TFEM_ERR_UNSUPPORTED, "Integration order not implemented");
  if (!nodes_host || !weights_host)
    return tfem::fail(TFEM_ERR_INVALID_ARGUMENT, "NULL output");
  for (int q = 0; q < nq; ++q) {
    nodes_host[2 * q] = nodes[q][0];
    nodes_host[2 * q + 1] = nodes[q][1];
    weights_host[q] = weights[q];
  }
  return TFEM_OK;
}

int tfem_csr_symbolic_count(const void *conn_host, int idx_bytes, int64_t n_elems, int n_local,
                            int64_t n_dofs, int64_t *rowptr_host, int64_t *nnz_host) {
  if (int st = tfem::check_symbolic_args(conn_host, idx_bytes, n_elems, n_local, n_dofs))
    return st;
  if (!rowptr_host || !nnz_host) return tfem::fail(TFEM_ERR_INVALID_ARGUMENT, "NULL output");
  if (idx_bytes == 4)
    return tfem::symbolic_count(static_cast<const int32_t *>(conn_host), n_elems, n_local,
                                n_dofs, rowptr_host, nnz_host);
  return tfem::symbolic_count(static_cast<const int64_t *>(conn_host), n_elems, n_local, n_dofs,
                              rowptr_host, nnz_host);
}

int tfem_csr_symbolic_fill(const void *conn_host, int idx_bytes, int64_t n_elems, int n_local,
                           int64_t n_dofs, const int64_t *rowptr_host, int32_t *colind_host,
                           int32_t *slots_host) {
  if (int st = tfem::check_symbolic_args(conn_host, idx_bytes, n_elems, n_local, n_dofs))
    return st;
  if (!rowptr_host || (!colind_host && rowptr_host[n_dofs] > 0) || (!slots_host && n_elems > 0))
    return tfem::fail(TFEM_ERR_INVALID_ARGUMENT, "NULL pointer");
  if (idx_bytes == 4)
    return tfem::symbolic_fill(static_cast<const int32_t *>(conn_host), n_elems, n_local, n_dofs,
                               rowptr_host, colind_host, slots_host);
  return tfem::symbolic_fill(static_cast<const int64_t *>(conn_host), n_elems, n_local, n_dofs,
                             rowptr_host, colind_host, slots_host);
}

int tfem_csr_pattern_create(const void *conn_host, int idx_bytes, int64_t n_elems, int n_local,
                            int64_t n_dofs, void **pattern_out, int64_t *nnz_host) {
  if (int st = tfem::check_symbolic_args(conn_host, idx_bytes, n_elems, n_local, n_dofs)) return st;
  if (!pattern_out || !nnz_host) return tfem::fail(TFEM_ERR_INVALID_ARGUMENT, "NULL output");
  *pattern_out = nullptr;
  tfem::PatternHandle *h = nullptr;
  const int st = idx_bytes == 4
                     ? tfem::pattern_create(static_cast<const int32_t *>(conn_host), n_elems, n_local, n_dofs, &h)
                     : tfem::pattern_create(static_cast<const int64_t *>(conn_host), n_elems, n_local, n_dofs, &h);
  if (st != TFEM_OK) return st;
  *pattern_out = h;
  *nnz_host = h->p.nnz;
  return TFEM_OK;
}

int tfem_csr_pattern_export(const void *pattern, int64_t *rowptr_host, int32_t *colind_host) {
  if (!pattern || !rowptr_host) return tfem::fail(TFEM_ERR_INVALID_ARGUMENT, "NULL pointer");
  const auto *h = static_cast<const tfem::PatternHandle *>(pattern);
  if (h->p.nnz > 0 && !colind_host) return tfem::fail(TFEM_ERR_INVALID_ARGUMENT, "NULL pointer");
  std::copy(h->p.rowptr.begin(), h->p.rowptr.end(), rowptr_host);
  if (h->idx_bytes == 4)
    tfem::pattern_columns(static_cast<const int32_t *>(h->conn), h->p, rowptr_host, colind_host);
  else
    tfem::pattern_columns(static_cast<const int64_t *>(h->conn), h->p, rowptr_host, colind_host);
  return TFEM_OK;
}

void tfem_csr_pattern_destroy(void *pattern) { delete static_cast<tfem::PatternHandle *>(pattern); }

int tfem_csr_symbolic_slots(const void *conn_host, int idx_bytes, int64_t n_elems, int n_local,
                            int64_t n_dofs, const int64_t *rowptr_host, const int32_t *colind_host,
                            int32_t *slots_host) {
  if (int st = tfem::check_symbolic_args(conn_host, idx_bytes, n_elems, n_local, n_dofs)) return st;
  if (!rowptr_host || (!colind_host && rowptr_host[n_dofs] > 0) || (!slots_host && n_elems > 0))
    return tfem::fail(TFEM_ERR_INVALID_ARGUMENT, "NULL pointer");
  if (idx_bytes == 4) {
    if (int st = tfem::check_conn(static_cast<const int32_t *>(conn_host), n_elems * n_local, n_dofs)) return st;
    tfem::fill_slots(static_cast<const int32_t *>(conn_host), n_elems, n_local, rowptr_host, colind_host, slots_host);
  } else {
    if (int st = tfem::check_conn(static_cast<const int64_t *>(conn_host), n_elems * n_local, n_dofs)) return st;
    tfem::fill_slots(static_cast<const int64_t *>(conn_host), n_elems, n_local, rowptr_host, colind_host, slots_host);
  }
  return TFEM_OK;
}

int tfem_csr_gather_map(const int32_t *slots_host, int64_t n_elems, int nn, int64_t nnz,
                        int64_t *gptr_host, int32_t *gsrc_host) {
  if (n_elems < 0 || nn < 1 || nnz < 0 || !gptr_host || (n_elems > 0 && (!slots_host || !gsrc_host)))
    return tfem::fail(TFEM_ERR_INVALID_ARGUMENT, "bad arguments");
  const int64_t n_entries = n_elems * nn;
  if (n_entries >= (int64_t(1) << 31))
    return tfem::fail(TFEM_ERR_INDEX_RANGE, "%lld local entries do not fit the int32 gather map",
                      (long long)n_entries);
  for (int64_t p = 0; p <= nnz; ++p) gptr_host[p] = 0;
  for (int64_t t = 0; t < n_entries; ++t) {
    const int32_t p = slots_host[t];
    if (p < 0 || p >= nnz) return tfem::fail(TFEM_ERR_INDEX_RANGE, "slot %d outside [0, nnz)", p);
    gptr_host[p + 1]++;
  }
  for (int64_t p = 0; p < nnz; ++p) gptr_host[p + 1] += gptr_host[p];
  // element-major walk: every CSR entry lists its contributions in ascending (element, local
  // entry) order, the order of a sequential index_put_(accumulate=True)
  for (int64_t e = 0; e < n_elems; ++e)
    for (int k = 0; k < nn; ++k) {
      const int32_t p = slots_host[e * nn + k];
      gsrc_host[gptr_host[p]++] = int32_t(int64_t(k) * n_elems + e);  // entry-major local index
    }
  for (int64_t p = nnz; p > 0; --p) gptr_host[p] = gptr_host[p - 1];
  gptr_host[0] = 0;
  return TFEM_OK;
}

}  // extern "C"
