// Shared host-side helpers of libtfem_hip: error reporting and the reference-element tables.
#pragma once

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "tfem_assembly.h"

namespace tfem {

int fail(int status, const char *fmt, ...);

constexpr int kMaxQuad = 6;

// Reference-element data evaluated on the host IN THE ARITHMETIC OF THE REAL
// TYPE (the reference builds these with torch at the default dtype, so a float32
// run rounds them to float32 first; element_tri.py:23-26,43-70,77-130), then
// widened to double for transport.  hw = 0.5 * w_q (basis.py:93-96).
struct TriTables {
  int nq;
  double hw[kMaxQuad];
  double lam[kMaxQuad][3];       // barycentric coordinates = P1 shape functions
  double phi2[kMaxQuad][6];      // P2 shape functions
  double rgrad2[kMaxQuad][6][2];  // P2 reference gradients (before @ inv_jacobian)
};

// Returns false for an order the reference raises NotImplementedError on.
bool triangle_rule(int quad_order, int *nq, double nodes[kMaxQuad][2], double weights[kMaxQuad]);
bool build_tri_tables(int quad_order, int real_bytes, TriTables *out);

// What a CSR pattern handle (tfem_csr_pattern_create) holds, for the plan builders that start
// from it: the DoF -> elements incidence (lists in arrival order; a builder may sort them in
// place), the row pointers and the caller's connectivity.
struct PatternView {
  const void *conn;
  int idx_bytes;
  int n_local;
  int64_t n_elems, n_dofs, nnz;
  const int64_t *inc_ptr;
  int32_t *inc;
  const int64_t *rowptr;
};
bool pattern_view(void *pattern_handle, PatternView *out);

}  // namespace tfem
