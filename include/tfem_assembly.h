/*
 * tfem_assembly.h -- C ABI of the MI355X (gfx950) element-wise FEM assembly path.
 *
 * Drop-in boundary for the hot path of Nicolas-Zamorano/pytorch_fem_solver
 * (package torch_fem).  The reference has no FFI: the path is a sequence of
 * torch tensor expressions inside AbstractBasis (torch_fem/basis/
 * abstract_basis.py:42-112).  Each entry point below names the reference
 * lines it replaces.  All signatures are plain pointers and sizes: no torch
 * types cross this boundary.
 *
 * Conventions
 *   - "device" pointers are HIP device pointers valid on the current device;
 *     "host" pointers are ordinary process memory.  The library never
 *     allocates or frees caller-visible memory and keeps no global state (the
 *     only library-owned object is the opaque tile-plan handle, below).
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).
 *     Device entry points only enqueue work; they never synchronise.
 *   - every function returns a tfem_status (0 = success).  tfem_last_error()
 *     returns a thread-local human-readable message for the last failure.
 *   - connectivity is row-major (n_elems, n_local) with 4- or 8-byte signed
 *     indices (`idx_bytes`), as the reference keeps it (int32 from MeshTri,
 *     abstract_mesh.py:51-54; int64 from FractureBasis, fracture_basis.py:80).
 *   - real type: `real_bytes` 8 = double (the graded path), 4 = float.
 *   - scatter convention (reference basis.py:73-76 + abstract_basis.py:166-167):
 *     local[i][j] is ADDED to A[conn[e][j]][conn[e][i]].
 *   - quadrature: `quad_order` 1..4 selects the 1/3/4/6-point triangle rule
 *     with the literals of element_tri.py:77-130.
 */
#ifndef TFEM_ASSEMBLY_H
#define TFEM_ASSEMBLY_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TFEM_ABI_VERSION 2

typedef enum tfem_status {
  TFEM_OK = 0,
  TFEM_ERR_INVALID_ARGUMENT = 1, /* NULL pointer, negative size, bad idx_bytes ... */
  TFEM_ERR_UNSUPPORTED = 2,      /* quadrature / polynomial order the reference raises on */
  TFEM_ERR_HIP = 3,              /* a HIP runtime call failed */
  TFEM_ERR_INDEX_RANGE = 4,      /* connectivity entry outside [0, n_dofs) or nnz >= 2^31 */
  TFEM_ERR_NO_DEVICE = 5         /* no gfx950 device visible */
} tfem_status;

int tfem_abi_version(void);
const char *tfem_status_string(int status);
const char *tfem_last_error(void);
/* Number of visible HIP devices (0 when none); never fails. */
int tfem_device_count(void);
/* Number of quadrature points of `quad_order` (1,3,4,6), or 0 if unsupported.
 * Replaces the shape of ElementTri._compute_gauss_values (element_tri.py:77-130). */
int tfem_quadrature_size(int quad_order);
/* Copy the rule: nodes (Q,2) and weights (Q) as doubles (host pointers). */
int tfem_quadrature_rule(int quad_order, double *nodes_host, double *weights_host);

/* ------------------------------------------------------------------------- *
 * Symbolic phase (HOST memory, once per mesh).
 * Replaces Basis._compute_basis_parameters (basis.py:64-85): instead of the
 * 2 x (n^2 N_T) index tensors for a dense index_put_, it produces the CSR
 * pattern of the operator and, per element entry, the CSR position it adds to.
 * ------------------------------------------------------------------------- */

/* Pass 1: rowptr_host[n_dofs+1] (int64) and *nnz_host. */
int tfem_csr_symbolic_count(const void *conn_host, int idx_bytes, int64_t n_elems,
                            int n_local, int64_t n_dofs, int64_t *rowptr_host,
                            int64_t *nnz_host);
/* Pass 2: colind_host[nnz] (int32, ascending inside each row) and
 * slots_host[n_elems*n_local*n_local] (int32): slots[e][i][j] = CSR position of
 * (row conn[e][j], col conn[e][i]).  `rowptr_host` is the output of pass 1. */
int tfem_csr_symbolic_fill(const void *conn_host, int idx_bytes, int64_t n_elems,
                           int n_local, int64_t n_dofs, const int64_t *rowptr_host,
                           int32_t *colind_host, int32_t *slots_host);

/* The same in one pass (the pattern is built once, HOST, multi-threaded: TFEM_HOST_THREADS):
 *   create : builds the pattern, returns a handle (library memory; release with _destroy) and nnz;
 *            conn_host must stay valid until export
 *   export : rowptr_host[n_dofs+1], colind_host[nnz] (caller-owned)
 *   slots  : slots_host[n_elems*n_local*n_local] from the exported pattern -- only the scatter /
 *            gather paths need it, the row-form plans do not */
int tfem_csr_pattern_create(const void *conn_host, int idx_bytes, int64_t n_elems, int n_local,
                            int64_t n_dofs, void **pattern_out, int64_t *nnz_host);
int tfem_csr_pattern_export(const void *pattern, int64_t *rowptr_host, int32_t *colind_host);
void tfem_csr_pattern_destroy(void *pattern);
int tfem_csr_symbolic_slots(const void *conn_host, int idx_bytes, int64_t n_elems, int n_local,
                            int64_t n_dofs, const int64_t *rowptr_host, const int32_t *colind_host,
                            int32_t *slots_host);

/* ------------------------------------------------------------------------- *
 * Geometry cache (DEVICE).  Replaces AbstractBasis._compute_integral_values
 * (abstract_basis.py:42-63) with Basis._compute_jacobian_map /
 * _compute_integration_points / _compute_integral_weights (basis.py:87-96) and
 * ElementTri.compute_det_and_inv_map / compute_shape_functions
 * (element_tri.py:28-75,132-145), fused with the X[conn] gather
 * (abstract_mesh.py:257-262).  Any output pointer may be NULL (skipped).
 *   coords   (n_verts, 2)            conn (n_elems, 3) vertex ids
 *   v_grad   P1: (n_elems, 3, 2)     P2: (n_elems, Q, 6, 2)
 *   dx       (n_elems, Q)            = 0.5 * w_q * det   (signed det)
 *   points   (n_elems, Q, 2)         inv_jac (n_elems, 2, 2)
 * ------------------------------------------------------------------------- */
int tfem_tri_geometry(const void *coords, int real_bytes, const void *conn, int idx_bytes,
                      int64_t n_elems, int64_t n_verts, int poly_order, int quad_order,
                      void *v_grad, void *dx, void *points, void *inv_jac, void *stream);

/* ------------------------------------------------------------------------- *
 * Fused bilinear forms (DEVICE): alpha * grad(u).grad(v) + beta * u v,
 * i.e. the integrands `v_grad @ v_grad.mT`, `v @ v.mT` and their sum
 * (tests/test_assembly.py:68-73, examples/example_fractures_fem.py:112-116),
 * integrated as abstract_basis.py:83 and scattered as :87-91, gather fused.
 *   conn_geo  (n_elems, 3) vertex ids into coords (geometry)
 *   slots     (n_elems, n, n) from tfem_csr_symbolic_fill, n = 3 (P1) or 6 (P2)
 *   vals      (nnz) CSR values; OVERWRITTEN (zero-filled by this call, then summed)
 *   slots == NULL: LOCAL-BLOCK mode -- vals (n*n, n_elems) receives the element blocks
 *             entry-major (vals[(i*n+j) * n_elems + e] = local[e][i][j]), nnz = n*n*n_elems;
 *             tfem_csr_gather then forms the CSR values without atomics.
 * Fracture variant (fracture_basis.py:20-26,189-197): if `frac_pinv` is not NULL
 * the elements are `n_fractures` consecutive groups of n_elems/n_fractures; the
 * reference gradients are post-multiplied by frac_pinv[f] (2x3) and dx by
 * frac_det[f].
 * ------------------------------------------------------------------------- */
int tfem_tri_bilinear_csr(const void *coords, int real_bytes, const void *conn_geo,
                          int idx_bytes, int64_t n_elems, int64_t n_verts, int poly_order,
                          int quad_order, double alpha, double beta, const int32_t *slots,
                          void *vals, int64_t nnz, const void *frac_pinv,
                          const void *frac_det, int n_fractures, int64_t coords_per_fracture,
                          void *stream);

/* Fused linear form f(x_q) * v (tests/test_assembly.py:79-84): `fq` (n_elems, Q) are
 * the user's source values at the integration points.  out (n_dofs) is overwritten.
 * conn_dof (n_elems, n) are the global DoF ids (P1: n = 3, may equal conn_geo).
 * conn_dof == NULL: LOCAL-VECTOR mode -- out (n, n_elems) receives the element vectors
 * entry-major (out[i * n_elems + e]), n_dofs = n * n_elems; tfem_csr_gather with the gather
 * map of conn_dof (tfem_csr_gather_map(conn_dof, n_elems, n, n_dofs, ...)) then forms the
 * vector without atomics, in the reference's accumulation order. */
int tfem_tri_load_vector(const void *coords, int real_bytes, const void *conn_geo,
                         const void *conn_dof, int idx_bytes, int64_t n_elems,
                         int64_t n_verts, int poly_order, int quad_order, const void *fq,
                         void *out, int64_t n_dofs, const void *frac_det, int n_fractures,
                         int64_t coords_per_fracture, void *stream);

/* ------------------------------------------------------------------------- *
 * Generic quadrature-reduce + scatter (DEVICE) for integrands torch evaluated
 * from an arbitrary user callable.  integrand is (n_elems, Q, n, m) addressed by
 * element strides `es`,`qs` (in elements of the real type; 0 = broadcast) with
 * the inner (n, m) block contiguous.  Computes sum_q integrand * dx exactly as
 * abstract_basis.py:83 / :104 / :72 and
 *   bilinear  (m = n): adds into CSR vals through `slots`      (:87-91)
 *   linear    (m = 1): adds into out[conn_dof]                 (:106-110)
 *   functional(m = 1): writes out[e] = sum_k sum_q (n_inner = n, usually 1)  (:65-72)
 * `vals` / `out` are overwritten.  tfem_reduce_scatter_bilinear with slots == NULL works in
 * LOCAL-BLOCK mode like tfem_tri_bilinear_csr, tfem_reduce_scatter_linear with
 * conn_dof == NULL in LOCAL-VECTOR mode like tfem_tri_load_vector.
 * ------------------------------------------------------------------------- */
int tfem_reduce_scatter_bilinear(const void *integrand, int real_bytes, int64_t es, int64_t qs,
                                 const void *dx, int64_t n_elems, int n_quad, int n_local,
                                 const int32_t *slots, void *vals, int64_t nnz, void *stream);
int tfem_reduce_scatter_linear(const void *integrand, int real_bytes, int64_t es, int64_t qs,
                               const void *dx, int64_t n_elems, int n_quad, int n_local,
                               const void *conn_dof, int idx_bytes, void *out, int64_t n_dofs,
                               void *stream);
int tfem_reduce_functional(const void *integrand, int real_bytes, int64_t es, int64_t qs,
                           const void *dx, int64_t n_elems, int n_quad, int n_inner, void *out,
                           void *stream);

/* Gather map (HOST, once per mesh) and gather (DEVICE): the deterministic, atomic-free form
 * of the scatter abstract_basis.py:87-91.  map: from the slots of tfem_csr_symbolic_fill,
 * gptr_host (nnz+1) int64 and gsrc_host (n_elems*nn) int32 = for every CSR entry the
 * entry-major indices of the local-block entries that add to it, in ascending (element,
 * local entry) order -- the order a sequential index_put_(accumulate=True) adds them in.
 * gather: vals[p] = sum_t local[gsrc[t]], t in [gptr[p], gptr[p+1]); local = the
 * LOCAL-BLOCK output of tfem_tri_bilinear_csr / tfem_reduce_scatter_bilinear. */
int tfem_csr_gather_map(const int32_t *slots_host, int64_t n_elems, int nn, int64_t nnz,
                        int64_t *gptr_host, int32_t *gsrc_host);
int tfem_csr_gather(const void *local, int real_bytes, const int64_t *gptr, const int32_t *gsrc,
                    int64_t nnz, void *vals, void *stream);

/* y = A x for the assembled CSR operator (DEVICE; x, y of n_rows entries).  The consumer of
 * the assembled values: Krylov solves where the reference's dense reduce + torch.linalg.solve
 * (abstract_basis.py:114-117,177-195) cannot go. */
int tfem_csr_spmv(const int64_t *rowptr, const int32_t *colind, const void *vals, int real_bytes,
                  int64_t n_rows, const void *x, void *y, void *stream);

/* A P1 DoF vector u (n_verts entries) on both sides of every interior edge: replaces the
 * tensor branch of Basis.interpolate(InteriorEdgesBasis, u) (reference basis.py:98-177;
 * SURVEY 8(f) f-2).  conn: (n_elems, 3) int32 cell vertices; edge_cells: (n_edges, 2) int64
 * = mesh["interior_edges", "cells"]; points: (n_edges, n_points, 2) = the edge basis's
 * integration points.  value: (n_edges, 2, n_points); grad: (n_edges, 2, 2) (constant along
 * the edge for P1).  All DEVICE pointers; ids are trusted (they come from the mesh topology). */
int tfem_edge_interpolate_p1(const void *coords, int real_bytes, const int32_t *conn,
                             const int64_t *edge_cells, const void *points, int64_t n_edges,
                             int n_points, const void *u, void *value, void *grad, void *stream);

/* Adjoint of tfem_edge_interpolate_p1 in u: grad_u (n_verts entries, overwritten) =
 * A^T g_value + B^T g_grad for value = A u, grad = B u; g_value (n_edges, 2, n_points) and
 * g_grad (n_edges, 2, 2) are the cotangents of the two outputs.  What autograd derives from the
 * reference's expressions (basis.py:150-158) when u carries history; floating-point atomics, the
 * summation order is not fixed. */
int tfem_edge_interpolate_p1_backward(const void *coords, int real_bytes, const int32_t *conn,
                                      const int64_t *edge_cells, const void *points, int64_t n_edges,
                                      int n_points, const void *g_value, const void *g_grad,
                                      void *grad_u, int64_t n_verts, void *stream);

/* The same adjoint WITHOUT atomics (fixed summation order, bitwise reproducible): one lane per
 * vertex walks the (edge, side) pairs around it.  inc_ptr (n_verts + 1) / inc_side: for every
 * vertex the entries 4 * (2 * edge + side) + local index of the vertex in that side's cell,
 * ascending (DEVICE int64 arrays, built once per edge table by the caller). */
int tfem_edge_interpolate_p1_backward_rows(const void *coords, int real_bytes, const int32_t *conn,
                                           const int64_t *edge_cells, const void *points,
                                           int64_t n_edges, int n_points, const void *g_value,
                                           const void *g_grad, const int64_t *inc_ptr,
                                           const int64_t *inc_side, void *grad_u, int64_t n_verts,
                                           void *stream);

/* FractureBasis.interpolate(InteriorEdgesFractureBasis, u) (fracture_basis.py:225-272): a P1
 * DoF vector on both sides of every interior edge of every fracture.  All DEVICE:
 *   coords2d (F, n_verts, 2), coords3d (F, n_verts, 3) = mesh["vertices", "coordinates_3d"],
 *   conn (F, n_cells, 3) int32 per-fracture vertex ids, edge_cells (F, n_edges, 2) int64 per-
 *   fracture cell ids, points (F, n_edges, n_points, 3) the edge basis's 3-D integration points,
 *   pinv (F, 2, 3) = mesh["inv_jacobian_fracture_map"], u (n_u).
 *   value (F, n_edges, 2, n_points), grad (F, n_edges, 2, 3).
 * u is indexed with the PER-FRACTURE vertex ids of the cells, exactly as the reference does
 * (fracture_basis.py:229-231) -- reproduced, not repaired. */
int tfem_edge_interpolate_p1_fracture(const void *coords2d, const void *coords3d, int real_bytes,
                                      const int32_t *conn, const int64_t *edge_cells, const void *points,
                                      const void *pinv, int64_t n_fractures, int64_t n_verts,
                                      int64_t n_cells, int64_t n_edges, int n_points, const void *u,
                                      int64_t n_u, void *value, void *grad, void *stream);

/* CSR -> dense (n_dofs, n_dofs) row-major, the layout integrate_bilinear_form
 * returns in the reference (abstract_basis.py:81).  dense is overwritten. */
int tfem_csr_to_dense(const int64_t *rowptr, const int32_t *colind, const void *vals,
                      int real_bytes, int64_t n_dofs, void *dense, void *stream);

/* ------------------------------------------------------------------------- *
 * Tile plan (HOST, once per mesh) + tile kernel (DEVICE): the P1 headline path.
 * The CSR rows (= vertices) are cut into spatially compact tiles along a Z-order
 * curve; a tile owns its rows, processes every element incident to them,
 * accumulates in LDS and writes each CSR value once (no global atomics, no
 * zero-fill).  Replaces, like tfem_tri_bilinear_csr, abstract_basis.py:74-93
 * fused with abstract_mesh.py:257-262 and basis.py:64-96; requires P1 with DoFs =
 * vertices (conn indexes coords) and rows of at most 16 entries.
 *   create : plan handle from connectivity, coordinates (for the curve) and the CSR
 *            pattern of tfem_csr_symbolic_*; capacities bound a tile's elements,
 *            local vertices, accumulator entries and owned rows.
 *   sizes  : fills layout[24]:
 *            [0] n_tiles [1] n_records [2] n_local_verts [3] n_owned_rows [4] n_runs
 *            [5] max elems/tile [6] max verts/tile [7] max owned/tile
 *            [8] max accumulator entries/tile [9] max row length [10] max runs/tile
 *            [11] n_run_starts (= n_runs + n_tiles)
 *            [12..18] byte offsets of desc, records, vert_gid, row_loff, run_delta,
 *            run_lstart, elem_id inside the packed plan; [19] bytes of the packed plan;
 *            [20] 32-bit words per element record: 3 (12-byte form) or 2 (8-byte form, used
 *            when no row has more than 8 entries); [21..23] reserved
 *   pack   : write the packed plan (layout[19] bytes, caller-owned HOST memory):
 *            desc int32 (12 per tile) | records (12-byte form: per vertex 16 * local id |
 *            4-bit column positions << 16; 8-byte form: three 10-bit local ids, nine 3-bit
 *            positions) | vert_gid int32 | row_loff uint16 |
 *            run_delta int32 | run_lstart uint16 | elem_id int32 (original element of every
 *            record).  An output run is a maximal group of owned rows that is contiguous in
 *            the CSR value array.  The caller copies the blob to the device once.
 * The handle is internal library memory and must be released with _destroy.
 * ------------------------------------------------------------------------- */
int tfem_tile_plan_create(const void *conn_host, int idx_bytes, int64_t n_elems,
                          int64_t n_verts, const double *coords_host,
                          const int64_t *rowptr_host, const int32_t *colind_host,
                          int elem_cap, int vert_cap, int acc_cap, int own_cap,
                          void **plan_out);
int tfem_tile_plan_sizes(const void *plan, int64_t layout[24]);
int tfem_tile_plan_pack(const void *plan, void *blob_host);
void tfem_tile_plan_destroy(void *plan);
/* Largest capacity the compiled kernel accepts: what = 0 elements, 1 local vertices,
 * 2 owned rows, 3 accumulator entries per tile. */
int tfem_tile_capacity(int what);
/* One launch over the tile plan (`plan_device` = DEVICE copy of the packed plan,
 * `plan_layout_host` = the HOST layout[24] of tfem_tile_plan_sizes):
 *   vals != NULL : CSR values of alpha * stiffness + beta * mass (every entry written
 *                  once; vals need not be initialised)        [abstract_basis.py:74-93]
 *   fq   != NULL : load vector fout[n_verts] = sum_e sum_q fq[e][q] phi_i(x_q) dx_q from
 *                  the user's source values fq (n_elems, Q) in ORIGINAL element order
 *                  (every entry written once)                 [abstract_basis.py:95-112]
 * Either or both.  The kernel addresses every array through range-checked buffer
 * resources (each array must stay below 4 GiB). */
int tfem_p1_assemble_tiles(const void *coords, int real_bytes, int64_t n_verts, int quad_order,
                           double alpha, double beta, const void *plan_device,
                           const int64_t *plan_layout_host, void *vals, int64_t nnz,
                           const void *fq, int64_t n_elems, void *fout, void *stream);

/* ------------------------------------------------------------------------- *
 * Ring plan (HOST, once per mesh) + ring kernel (DEVICE): P1 alpha * stiffness +
 * beta * mass in owner-computes ROW form.  Same operation as tfem_p1_assemble_tiles
 * with vals != NULL (abstract_basis.py:74-93 fused with abstract_mesh.py:257-262,
 * basis.py:64-96, element_tri.py:132-145), same requirements (P1, DoFs = vertices,
 * rows of at most 16 entries), no atomics at all: one lane owns one CSR row, walks
 * the fan of triangles around its vertex (neighbours listed in fan order as 10-bit
 * tile-local ids, the triangle of every slot flagged with its stored orientation)
 * and writes the row once.
 *   create : TFEM_ERR_UNSUPPORTED when the triangles around some vertex do not form
 *            one closed fan or open fans (an edge with three triangles, duplicated or
 *            degenerate elements): use the tile plan for such a mesh.
 *   sizes  : fills layout[32]: [0] n_tiles [1] n_rows [2] n_local_verts
 *            [3] max local verts/tile [4] max owned rows/tile [5] max row length
 *            [6] neighbour slots per row record (7 | 15) [7] dwords per row record (4 | 8)
 *            [8..11] byte offsets of desc, rows, rowstart, vert_gid in the packed plan
 *            [12] bytes of the packed plan [13] 1 when every wave's rows are consecutive
 *            vertices (tiles made of chunks of the numbering) [14] max halo vertices/tile
 *            [15] byte offset of row_ecodes, [16] of tile_elems in the packed plan [17] max
 *            elements per tile [18] 1 when every tile's elements fit the kernel's LDS stage
 *            (the fused load vector needs it) [19] entries of tile_elems [20] byte offset of
 *            tile_tverts (uint32 per tile element, in the order of the tile's ascending element
 *            list: the three tile-local vertex ids of the element in its own local order,
 *            10 bits each; desc[19] of a tile = its offset into this array), 0 when absent
 *            [21] entries of tile_tverts [22] byte offset and [23] number of the long rows (24-dword
 *            records of the vertices with 8 .. 15 neighbours in a plan with 4-dword records: their
 *            rows are written by a second launch; csrc/tfem_rings_host.cpp)
 *            [24] byte offset of chain_order (int32 per tile: the order in which the launches that
 *            evaluate a source program walk the tiles -- along the space-filling curve, the tiles
 *            owning flagged vertices first) [25] chain length L: a workgroup takes L consecutive
 *            positions of that order; inside such a block an element in the fans of two
 *            consecutive tiles is evaluated once, by the earlier one [26] byte offset of hand_in
 *            (uint16 per owned row, parallel to rowstart: the local id the row's vertex has in the
 *            PREVIOUS tile of its block, 0xFFFF: none -- the earlier tile sums the shares of all its
 *            local vertices, the row adds that sum to its own)
 *            [27] most elements a tile evaluates itself (desc[18] >> 8 of a tile = its count,
 *            tile_tverts holds exactly those; desc[18] & 0xFF = element-list mode)
 *   pack   : desc int32 (20 per tile: vert_off, n_vert, row_off, first row of wave 0..3 of
 *            the 256-lane workgroup (the first is 0), n_own, vertex id of the first row of
 *            wave 0..3, CSR offset of the first row of wave 0..3, offset into tile_elems,
 *            number of elements of the tile, 0, 0) | row records
 *            (bit layout: csrc/tfem_rings_host.cpp) | rowstart int32 (CSR offset of every
 *            owned row) | vert_gid int32 (owned rows first, ascending, then the halo)
 *   capacity: what = 0 owned rows per tile, 1 local vertices per tile
 * ------------------------------------------------------------------------- */
int tfem_ring_plan_create(const void *conn_host, int idx_bytes, int64_t n_elems,
                          int64_t n_verts, const double *coords_host,
                          const int64_t *rowptr_host, const int32_t *colind_host,
                          int own_cap, int vert_cap, void **plan_out);
int tfem_ring_plan_sizes(const void *plan, int64_t layout[32]);
int tfem_ring_plan_pack(const void *plan, void *blob_host);
void tfem_ring_plan_destroy(void *plan);
int tfem_ring_capacity(int what);
/* vals (nnz) = CSR values of alpha * stiffness + beta * mass; every entry of a row with
 * at least one element is written exactly once (vals need not be initialised).
 * fq != NULL: the same launch also writes the load vector fout[n_verts] = sum_e sum_q
 * fq[e][q] phi_i(x_q) dx_q from the user's source values fq (n_elems, Q) in ORIGINAL element
 * order (abstract_basis.py:95-112; every entry written once, 0 for a vertex without
 * elements).  vals == NULL with fq != NULL: the load vector alone. */
int tfem_p1_assemble_rings(const void *coords, int real_bytes, int64_t n_verts, int quad_order,
                           double alpha, double beta, const void *plan_device,
                           const int64_t *plan_layout_host, void *vals, int64_t nnz,
                           const void *fq, int64_t n_elems, void *fout, void *stream);

/* ------------------------------------------------------------------------- *
 * Source programs: the coefficient f of the linear form f(x_q) * v
 * (abstract_basis.py:95-112 with the callers' closed vocabulary, SURVEY 8 a-7:
 * tests/test_assembly.py:75-84, examples/example_weak.py:59-61 ...) as a postfix
 * program over the coordinates (x, y) of an integration point, evaluated inside the
 * kernels at x_q = bar(q)^T X (basis.py:90-91).  The reference evaluates the user's
 * torch expressions on the cached (n_elems, Q, 1, 2) tensor and multiplies by v on
 * every call; here the Python tracer (pytorch_fem_solver_amd/basis/forms.py) records
 * those expressions once and the assembly launch re-evaluates them, so the 8 Q bytes
 * per element of pre-evaluated source values never exist in HBM.
 *   A stack machine of TFEM_SOURCE_STACK entries.  ops[i] with constant consts[i]:
 *   PUSH_X / PUSH_Y / PUSH_C   push c * x, c * y, or the constant c
 *   ADD SUB MUL DIV            pop hi, pop lo, push lo (op) hi;  SUB_R, DIV_R: hi (op) lo
 *   ADD_C MUL_C RSUB_C RDIV_C  top = top + c, top * c, c - top, c / top
 *   NEG ABS                    top = -top, |top|
 *   SIN COS EXP SQRT LOG TANH  top = c * fn(top)
 *   POW_I                      top = top^n, n = (int)c in 2..8, by multiplications
 *                              from the left (x*x, x*x*x as torch evaluates them)
 * A valid program never pops an empty stack, never exceeds the stack and leaves
 * exactly one entry: the value of f.
 * ------------------------------------------------------------------------- */
#define TFEM_SOURCE_MAX_OPS 32
#define TFEM_SOURCE_STACK 4
enum tfem_source_op {
  TFEM_SRC_END = 0,
  TFEM_SRC_PUSH_X = 1, TFEM_SRC_PUSH_Y = 2, TFEM_SRC_PUSH_C = 3,
  TFEM_SRC_ADD = 4, TFEM_SRC_SUB = 5, TFEM_SRC_SUB_R = 6, TFEM_SRC_MUL = 7,
  TFEM_SRC_DIV = 8, TFEM_SRC_DIV_R = 9,
  TFEM_SRC_ADD_C = 10, TFEM_SRC_MUL_C = 11, TFEM_SRC_RSUB_C = 12, TFEM_SRC_RDIV_C = 13,
  TFEM_SRC_NEG = 14, TFEM_SRC_ABS = 15, TFEM_SRC_POW_I = 16,
  TFEM_SRC_SIN = 17, TFEM_SRC_COS = 18, TFEM_SRC_EXP = 19, TFEM_SRC_SQRT = 20,
  TFEM_SRC_LOG = 21, TFEM_SRC_TANH = 22,
  TFEM_SRC_OP_COUNT = 23
};
typedef struct tfem_source_program {
  int32_t n_ops;
  int32_t reserved;
  uint8_t ops[TFEM_SOURCE_MAX_OPS];
  double consts[TFEM_SOURCE_MAX_OPS];
} tfem_source_program;

/* TFEM_OK when `program` (HOST) is a valid source program, TFEM_ERR_INVALID_ARGUMENT
 * (tfem_last_error says why) otherwise. */
int tfem_source_validate(const tfem_source_program *program);

/* fq (n_elems, Q) = f at the integration points of every element (DEVICE; conn (n_elems, 3)
 * vertex ids, `program` HOST): what the reference's callers compute with torch from
 * basis.integration_points before multiplying by v (tests/test_assembly.py:79-84).  Feeds the
 * entry points that take pre-evaluated source values (tfem_tri_load_vector,
 * tfem_p1_assemble_tiles, P2) when the ring kernel does not apply. */
int tfem_source_eval(const void *coords, int real_bytes, const void *conn, int idx_bytes,
                     int64_t n_elems, int64_t n_verts, int quad_order,
                     const tfem_source_program *program, void *fq, void *stream);

/* tfem_p1_assemble_rings with the source values computed in the launch: the load vector
 * fout[n_verts] = sum_e sum_q f(x_q) phi_i(x_q) dx_q of the program `source` (HOST), and, with
 * vals != NULL, the CSR values of alpha * stiffness + beta * mass in the same launch.  The
 * plan must carry the per-tile element vertex table (layout[20] != 0). */
int tfem_p1_assemble_rings_source(const void *coords, int real_bytes, int64_t n_verts,
                                  int quad_order, double alpha, double beta,
                                  const void *plan_device, const int64_t *plan_layout_host,
                                  void *vals, int64_t nnz, const tfem_source_program *source,
                                  int64_t n_elems, void *fout, void *stream);

/* Multi-GPU: the rows shared with other ranks first.  create_priority = create with
 * vertex_priority_host (n_verts bytes, non-zero = flagged; NULL = none): the tiles that own a flagged
 * vertex come first in the plan's tile list (*n_priority_tiles of them), the order inside both groups
 * unchanged.  _range = tfem_p1_assemble_rings / _source (fq or source or neither) over the tiles
 * [tile_first, tile_first + tile_count) only (tile_count < 0: to the end): the launch over the
 * priority tiles completes every shared row, so their exchange can run beside the launch over the
 * remaining tiles.  (In a plan with long rows those are written with the range that ends the list.) */
int tfem_ring_plan_create_priority(const void *conn_host, int idx_bytes, int64_t n_elems, int64_t n_verts,
                                   const double *coords_host, const int64_t *rowptr_host,
                                   const int32_t *colind_host, int own_cap, int vert_cap,
                                   const uint8_t *vertex_priority_host, void **plan_out,
                                   int64_t *n_priority_tiles);
/* The same plan from a CSR pattern handle (tfem_csr_pattern_create of the P1 connectivity, still
 * alive, with the colind tfem_csr_pattern_export wrote): reuses the handle's connectivity, row
 * pointers and vertex -> elements incidence instead of building the incidence again (the lists of
 * the handle are sorted in place).  Same plan bytes as tfem_ring_plan_create_priority. */
int tfem_ring_plan_create_from_pattern(void *pattern, const double *coords_host, const int32_t *colind_host,
                                       int own_cap, int vert_cap, const uint8_t *vertex_priority_host,
                                       void **plan_out, int64_t *n_priority_tiles);
int tfem_p1_assemble_rings_range(const void *coords, int real_bytes, int64_t n_verts, int quad_order,
                                 double alpha, double beta, const void *plan_device,
                                 const int64_t *plan_layout_host, void *vals, int64_t nnz, const void *fq,
                                 const tfem_source_program *source, int64_t n_elems, void *fout,
                                 int64_t tile_first, int64_t tile_count, void *stream);


/* ------------------------------------------------------------------------- *
 * The VPINN residual linear form, fused (DEVICE; P1 on one 2-D mesh):
 *   r_i = sum_T sum_q dx_q ( f(x_q) v_i(q) + flux_sign * grad v_i . g_q )
 * = integrate_linear_form of `rhs(x, y) * v - v_grad @ gradient(points).mT`
 * (examples/example_weak.py:64-75 with abstract_basis.py:95-112; flux_sign = -1 there).
 *   fq (n_elems, Q) source values, or `source` (HOST) a source program, or neither
 *   flux (n_elems, Q, 2) = g at the integration points, or NULL
 *   out_local (3, n_elems): the element vectors entry-major (out[i * n_elems + e]); the global
 *   vector follows from tfem_csr_gather with the gather map of the connectivity, as for
 *   tfem_tri_load_vector in LOCAL-VECTOR mode (no atomics, the reference's accumulation order).
 * backward: the adjoint, for the training step that differentiates through the form
 * (example_weak.py:132-152): from the cotangent of r (n_verts entries)
 *   grad_flux[e][q][k] = flux_sign * dx_q * sum_i cot[conn[e][i]] * (grad v_i)_k
 *   grad_fq[e][q]      = dx_q * sum_i cot[conn[e][i]] * v_i(q)
 * either may be NULL; every entry is written once, no atomics.
 * ------------------------------------------------------------------------- */
int tfem_p1_residual_local(const void *coords, int real_bytes, const void *conn, int idx_bytes,
                           int64_t n_elems, int64_t n_verts, int quad_order, const void *fq,
                           const tfem_source_program *source, const void *flux, double flux_sign,
                           void *out_local, void *stream);
int tfem_p1_residual_backward(const void *coords, int real_bytes, const void *conn, int idx_bytes,
                              int64_t n_elems, int64_t n_verts, int quad_order, const void *cotangent,
                              double flux_sign, void *grad_fq, void *grad_flux, void *stream);

/* ------------------------------------------------------------------------- *
 * P2 row plan (HOST, once per mesh) + P2 row kernels (DEVICE): alpha * stiffness +
 * beta * mass for the quadratic element in owner-computes ROW form (one lane per CSR
 * row of a vertex DoF or of an edge DoF, no atomics).  Same operation as
 * tfem_tri_bilinear_csr with poly_order = 2 (abstract_basis.py:74-93, element_tri.py:43-70);
 * requires DoFs numbered "vertices, then edges" with the local edge order (v0,v1), (v1,v2),
 * (v2,v0) (conn_dof (n_elems, 6), vertex DoF id = vertex id), at most 15 neighbours per
 * vertex and a numbering with locality; otherwise create returns TFEM_ERR_UNSUPPORTED and
 * the caller uses the local-block + gather path.  Vertex rows with 8 .. 15 neighbours (every
 * Delaunay mesh has them) are written by a third launch, one lane per such row.
 *   sizes : layout[24]: [0] vertex-row tiles [1] edge-row tiles [2] n_verts [3] n_edges
 *           [4] max local vertices of a vertex tile [5] of an edge tile [6] max halo of a
 *           vertex tile [7],[8] local vertices listed for vertex / edge tiles
 *           [10..15] byte offsets of desc, rows, vert_gid of the vertex tiles and of the edge
 *           tiles in the packed plan [16] bytes of the packed plan [17] byte offset and [18] number
 *           of the long vertex rows (32-dword records) [19..21] element codes of the load-vector
 *           launch (tfem_p2_load_rows)
 *   pack  : record and descriptor layout: csrc/tfem_p2rows_host.cpp
 * ------------------------------------------------------------------------- */
int tfem_p2_plan_create(const int32_t *conn_dof_host, int64_t n_elems, int64_t n_verts,
                        int64_t n_dofs, const double *coords_host, const int64_t *rowptr_host,
                        const int32_t *colind_host, void **plan_out);
int tfem_p2_plan_sizes(const void *plan, int64_t layout[24]);
int tfem_p2_plan_pack(const void *plan, void *blob_host);
void tfem_p2_plan_destroy(void *plan);
/* vals (nnz): every entry written exactly once (launches: vertex rows, long vertex rows if
 * any, edge rows). */
int tfem_p2_assemble_rows(const void *coords, int real_bytes, int quad_order, double alpha,
                          double beta, const void *plan_device, const int64_t *plan_layout_host,
                          void *vals, int64_t nnz, void *stream);
/* The P2 load vector in ROW form over the same plan (replaces integrate_linear_form of f v for
 * ElementTri(2, .): abstract_basis.py:95-112, element_tri.py:43-70, dx of basis.py:93-96):
 *   out[r] = sum over the triangles T of DoF r of det_T * sum_q fq[T][q] phi_loc(q) w_q / 2
 * fq (n_elems, Q) source values at the integration points of the elements in their stored frame,
 * out (n_dofs) written exactly once per DoF (launches: vertex rows, long vertex rows if any, edge
 * rows); no atomics, no element vectors.  The plan's layout[19], [20], [21]: byte offsets of the
 * element codes (element * 4 + local index of the DoF among the three of its kind) of the vertex
 * rows (8 dwords per row), the edge rows (2 dwords) and the long rows (16 dwords). */
int tfem_p2_load_rows(const void *coords, int real_bytes, int quad_order, const void *plan_device,
                      const int64_t *plan_layout_host, const void *fq, int64_t n_elems, void *out,
                      int64_t n_dofs, void *stream);

/* ------------------------------------------------------------------------- *
 * Interface exchange of the multi-GPU sharding (DEVICE; the path shards by element
 * range, DoFs on an inter-rank interface are summed with one all-reduce of a packed
 * buffer: SURVEY 8(e); the reference is single-process).  pack: buf (nbuf) is zeroed,
 * then buf[k_pos[i]] = vals[k_idx[i]] (i < nk) and buf[f_pos[i]] = f[f_idx[i]] (i < nf);
 * unpack: the reverse copies.  vals / f may be NULL (that part is skipped); index arrays
 * are int64 device arrays.  One kernel launch each.
 * ------------------------------------------------------------------------- */
int tfem_interface_pack(const void *vals, const void *f, int real_bytes, const int64_t *k_idx,
                        const int64_t *k_pos, int64_t nk, const int64_t *f_idx, const int64_t *f_pos,
                        int64_t nf, void *buf, int64_t nbuf, void *stream);
int tfem_interface_unpack(void *vals, void *f, int real_bytes, const int64_t *k_idx,
                          const int64_t *k_pos, int64_t nk, const int64_t *f_idx,
                          const int64_t *f_pos, int64_t nf, const void *buf, void *stream);
/* pack in ONE launch, for the repeated step of a sharded run (the exchange chain -- zero, pack,
 * all-reduce, unpack, each behind the other -- is what bounds such a step at 1e6 elements per rank):
 * src (nbuf, int64 device array): >= 0 an entry of vals, <= -2 entry -src - 2 of f, -1 a position
 * other ranks own (written as zero).  buf[p] for every p < nbuf. */
int tfem_interface_pack_dense(const void *vals, const void *f, int real_bytes, const int64_t *src,
                              int64_t nbuf, void *buf, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TFEM_ASSEMBLY_H */
