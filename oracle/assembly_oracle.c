/*
 * CPU ORACLE (C / OpenMP) -- test infrastructure, NOT product code.
 *
 * Plain-C restatement of the reference's P1 assembly path (Nicolas-Zamorano/
 * pytorch_fem_solver, torch_fem), one element per loop iteration, in the reference's
 * operation order; each step cites the reference file:line it follows.  Checked against
 * oracle/assembly_oracle.py (which is pinned to the reference-generated fixtures) by
 * tests/test_oracle_golden.py.  Used as the multi-core CPU baseline of bench.py and as a
 * full-size checker in the GPU tests.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  Build: __graft_entry__.build_oracle()
 * (gcc -O2 -fopenmp -ffp-contract=off).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* element_tri.py:77-130 (literals as written there) */
static int gauss_rule(int order, double nodes[6][2], double w[6]) {
  int q, nq;
  switch (order) {
    case 1:
      nq = 1;
      nodes[0][0] = 1.0 / 3; nodes[0][1] = 1.0 / 3; w[0] = 1.0;
      break;
    case 2:
      nq = 3;
      nodes[0][0] = 1.0 / 6; nodes[0][1] = 1.0 / 6;
      nodes[1][0] = 2.0 / 3; nodes[1][1] = 1.0 / 6;
      nodes[2][0] = 1.0 / 6; nodes[2][1] = 2.0 / 3;
      for (q = 0; q < 3; ++q) w[q] = 1.0 / 3;
      break;
    case 3:
      nq = 4;
      nodes[0][0] = 1.0 / 3; nodes[0][1] = 1.0 / 3;
      nodes[1][0] = 0.6; nodes[1][1] = 0.2;
      nodes[2][0] = 0.2; nodes[2][1] = 0.6;
      nodes[3][0] = 0.2; nodes[3][1] = 0.2;
      w[0] = -9.0 / 16; w[1] = w[2] = w[3] = 25.0 / 48;
      break;
    case 4:
      nq = 6;
      nodes[0][0] = 0.816847572980459; nodes[0][1] = 0.091576213509771;
      nodes[1][0] = 0.091576213509771; nodes[1][1] = 0.816847572980459;
      nodes[2][0] = 0.091576213509771; nodes[2][1] = 0.091576213509771;
      nodes[3][0] = 0.108103018168070; nodes[3][1] = 0.445948490915965;
      nodes[4][0] = 0.445948490915965; nodes[4][1] = 0.108103018168070;
      nodes[5][0] = 0.445948490915965; nodes[5][1] = 0.445948490915965;
      w[0] = w[1] = w[2] = 0.109951743655322;
      w[3] = w[4] = w[5] = 0.223381589678011;
      break;
    default:
      return 0;
  }
  return nq;
}

/* n > 0: number of OpenMP threads of the following calls (a container's CPU quota can be far
 * below the core count omp_get_max_threads() starts from) */
void oracle_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

int oracle_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/*
 * Local blocks of one mesh.  K_local (n_elems, 3, 3) = sum_q (alpha g_i.g_j + beta l_i l_j) dx_q,
 * f_local (n_elems, 3) = sum_q (fq[e][q] l_i(q)) dx_q (fq may be NULL), either output may
 * be NULL.  Returns the number of quadrature points, 0 for an unsupported order.
 */
int oracle_p1_local(const double *coords, const int32_t *conn, int64_t n_elems, int quad_order,
                    double alpha, double beta, const double *fq, double *K_local,
                    double *f_local) {
  double nodes[6][2], w[6], lam[6][3], hw[6];
  const int nq = gauss_rule(quad_order, nodes, w);
  int q;
  int64_t e;
  if (!nq) return 0;
  for (q = 0; q < nq; ++q) {
    lam[q][0] = 1.0 - nodes[q][0] - nodes[q][1]; /* element_tri.py:23-26 */
    lam[q][1] = nodes[q][0];
    lam[q][2] = nodes[q][1];
    hw[q] = 0.5 * w[q]; /* basis.py:93-96: reference_element_area * gaussian_weights */
  }
#pragma omp parallel for schedule(static)
  for (e = 0; e < n_elems; ++e) {
    const int32_t *c = conn + 3 * e;
    /* abstract_mesh.py:257-262 gather */
    const double x0 = coords[2 * c[0]], y0 = coords[2 * c[0] + 1];
    const double x1 = coords[2 * c[1]], y1 = coords[2 * c[1] + 1];
    const double x2 = coords[2 * c[2]], y2 = coords[2 * c[2] + 1];
    /* basis.py:87-88  J = X^T G */
    const double ja = x1 - x0, jb = x2 - x0, jc = y1 - y0, jd = y2 - y0;
    /* element_tri.py:132-145 */
    const double det = ja * jd - jb * jc;
    const double r = 1.0 / det;
    const double inv00 = r * jd, inv01 = r * (-jb), inv10 = r * (-jc), inv11 = r * ja;
    /* element_tri.py:41  v_grad = G @ inv */
    double g[3][2];
    double dx[6];
    int i, j, qq;
    g[0][0] = (-inv00) + (-inv10);
    g[0][1] = (-inv01) + (-inv11);
    g[1][0] = inv00; g[1][1] = inv01;
    g[2][0] = inv10; g[2][1] = inv11;
    for (qq = 0; qq < nq; ++qq) dx[qq] = hw[qq] * det;
    if (K_local) {
      for (i = 0; i < 3; ++i)
        for (j = 0; j < 3; ++j) {
          const double s = alpha * (g[i][0] * g[j][0] + g[i][1] * g[j][1]);
          double acc = 0.0; /* abstract_basis.py:83 */
          for (qq = 0; qq < nq; ++qq) acc = acc + (s + beta * (lam[qq][i] * lam[qq][j])) * dx[qq];
          K_local[9 * e + 3 * i + j] = acc;
        }
    }
    if (f_local && fq) {
      for (i = 0; i < 3; ++i) {
        double acc = 0.0; /* abstract_basis.py:104 */
        for (qq = 0; qq < nq; ++qq) acc = acc + (fq[nq * e + qq] * lam[qq][i]) * dx[qq];
        f_local[3 * e + i] = acc;
      }
    }
  }
  return nq;
}

/* abstract_basis.py:87-91 with the CSR slot map instead of the dense target:
 * vals[slots[e][i][j]] += K_local[e][i][j]  (slots already encode row conn[j], col conn[i]) */
void oracle_scatter_csr(const double *K_local, const int32_t *slots, int64_t n_entries,
                        double *vals, int64_t nnz) {
  int64_t k;
  memset(vals, 0, (size_t)nnz * sizeof(double));
#pragma omp parallel for schedule(static)
  for (k = 0; k < n_entries; ++k) {
#pragma omp atomic
    vals[slots[k]] += K_local[k];
  }
}

/* abstract_basis.py:106-110 */
void oracle_scatter_vector(const double *f_local, const int32_t *conn, int64_t n_entries,
                           double *f, int64_t n_dofs) {
  int64_t k;
  memset(f, 0, (size_t)n_dofs * sizeof(double));
#pragma omp parallel for schedule(static)
  for (k = 0; k < n_entries; ++k) {
#pragma omp atomic
    f[conn[k]] += f_local[k];
  }
}

/* basis.py:90-91  integration points (n_elems, Q, 2) */
int oracle_p1_points(const double *coords, const int32_t *conn, int64_t n_elems, int quad_order,
                     double *points) {
  double nodes[6][2], w[6];
  const int nq = gauss_rule(quad_order, nodes, w);
  int64_t e;
  if (!nq) return 0;
#pragma omp parallel for schedule(static)
  for (e = 0; e < n_elems; ++e) {
    const int32_t *c = conn + 3 * e;
    int q;
    for (q = 0; q < nq; ++q) {
      const double l0 = 1.0 - nodes[q][0] - nodes[q][1], l1 = nodes[q][0], l2 = nodes[q][1];
      points[2 * (nq * e + q)] = (l0 * coords[2 * c[0]] + l1 * coords[2 * c[1]]) + l2 * coords[2 * c[2]];
      points[2 * (nq * e + q) + 1] =
          (l0 * coords[2 * c[0] + 1] + l1 * coords[2 * c[1] + 1]) + l2 * coords[2 * c[2] + 1];
    }
  }
  return nq;
}
