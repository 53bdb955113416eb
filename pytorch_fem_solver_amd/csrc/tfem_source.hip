// Source programs (include/tfem_assembly.h, tfem_source.hpp): host-side validation and
// conversion, and the stand-alone evaluation kernel `tfem_source_eval` that writes the
// (n_elems, Q) source values for the assembly entry points that take them from memory.
#include <hip/hip_runtime.h>

#include <cstring>

#include "tfem_common.hpp"
#include "tfem_source.hpp"

namespace tfem {

int src_validate(const tfem_source_program *in) {
  if (!in) return fail(TFEM_ERR_INVALID_ARGUMENT, "source program is NULL");
  if (in->n_ops < 1 || in->n_ops > kSrcMaxOps)
    return fail(TFEM_ERR_INVALID_ARGUMENT, "source program: %d operations (1..%d)", in->n_ops, kSrcMaxOps);
  int depth = 0;
  for (int i = 0; i < in->n_ops; ++i) {
    const int op = in->ops[i];
    int pops = 0, pushes = 0;
    switch (op) {
      case TFEM_SRC_PUSH_X: case TFEM_SRC_PUSH_Y: case TFEM_SRC_PUSH_C: pushes = 1; break;
      case TFEM_SRC_ADD: case TFEM_SRC_SUB: case TFEM_SRC_SUB_R: case TFEM_SRC_MUL:
      case TFEM_SRC_DIV: case TFEM_SRC_DIV_R: pops = 2; pushes = 1; break;
      case TFEM_SRC_ADD_C: case TFEM_SRC_MUL_C: case TFEM_SRC_RSUB_C: case TFEM_SRC_RDIV_C:
      case TFEM_SRC_NEG: case TFEM_SRC_ABS: case TFEM_SRC_SIN: case TFEM_SRC_COS:
      case TFEM_SRC_EXP: case TFEM_SRC_SQRT: case TFEM_SRC_LOG: case TFEM_SRC_TANH:
        pops = 1; pushes = 1; break;
      case TFEM_SRC_POW_I:
        if (!(in->consts[i] >= 2.0 && in->consts[i] <= 8.0) || in->consts[i] != double(int(in->consts[i])))
          return fail(TFEM_ERR_INVALID_ARGUMENT, "source program: POW_I exponent %g (integers 2..8)", in->consts[i]);
        pops = 1; pushes = 1; break;
      default:
        return fail(TFEM_ERR_INVALID_ARGUMENT, "source program: unknown operation %d at %d", op, i);
    }
    if (depth < pops) return fail(TFEM_ERR_INVALID_ARGUMENT, "source program: operation %d pops an empty stack", i);
    depth += pushes - pops;
    if (depth > kSrcStack)
      return fail(TFEM_ERR_INVALID_ARGUMENT, "source program: more than %d stack entries at operation %d", kSrcStack, i);
  }
  if (depth != 1) return fail(TFEM_ERR_INVALID_ARGUMENT, "source program leaves %d values (1 expected)", depth);
  return TFEM_OK;
}

int src_depth(const tfem_source_program *in) {
  if (!in || in->n_ops < 1 || in->n_ops > kSrcMaxOps) return 0;
  int depth = 0, peak = 0;
  for (int i = 0; i < in->n_ops; ++i) {
    const int op = in->ops[i];
    if (op >= TFEM_SRC_PUSH_X && op <= TFEM_SRC_PUSH_C)
      ++depth;
    else if (op >= TFEM_SRC_ADD && op <= TFEM_SRC_DIV_R)
      --depth;
    peak = depth > peak ? depth : peak;
  }
  return peak;
}

template <typename T>
int src_convert(const tfem_source_program *in, SrcProgram<T> *out) {
  const int st = src_validate(in);
  if (st != TFEM_OK) return st;
  std::memset(out, 0, sizeof(*out));
  out->n_ops = in->n_ops;
  for (int i = 0; i < in->n_ops; ++i) {
    out->opw[i >> 2] |= uint32_t(in->ops[i]) << (8 * (i & 3));
    out->c[i] = T(in->consts[i]);  // the reference's python scalars meet tensors of the real type
  }
  return TFEM_OK;
}
template int src_convert<double>(const tfem_source_program *, SrcProgram<double> *);
template int src_convert<float>(const tfem_source_program *, SrcProgram<float> *);

namespace {

constexpr int kSrcBlock = 256;

template <typename T, typename I>
struct SrcEvalArgs {
  const T *coords;
  const I *conn;
  T *fq;
  int64_t n_elems;
  T lam[3][kMaxQuad];
  SrcProgram<T> src;
};

// One lane per element: gather (abstract_mesh.py:257-262), integration points (basis.py:90-91),
// f there.  The stores of a wave cover 64 Q consecutive values.
template <typename T, typename I, int Q>
__global__ __launch_bounds__(kSrcBlock) void k_source_eval(const SrcEvalArgs<T, I> a) {
  const int64_t e = int64_t(blockIdx.x) * kSrcBlock + threadIdx.x;
  const int64_t ec = e < a.n_elems ? e : a.n_elems - 1;  // whole waves run the program
  const I *c = a.conn + 3 * ec;
  const int64_t v0 = c[0], v1 = c[1], v2 = c[2];
  const T x0 = a.coords[2 * v0], y0 = a.coords[2 * v0 + 1];
  const T x1 = a.coords[2 * v1], y1 = a.coords[2 * v1 + 1];
  const T x2 = a.coords[2 * v2], y2 = a.coords[2 * v2 + 1];
  T xq[Q], yq[Q], fv[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    xq[q] = (a.lam[0][q] * x0 + a.lam[1][q] * x1) + a.lam[2][q] * x2;
    yq[q] = (a.lam[0][q] * y0 + a.lam[1][q] * y1) + a.lam[2][q] * y2;
  }
  using Args = SrcEvalArgs<T, I>;
  const SrcLanes<T> prog = src_load_lanes<T>(src_in_kernarg<T>(__builtin_offsetof(Args, src)));
  src_run<T, Q>(prog, xq, yq, fv);
  if (e < a.n_elems) {
#pragma unroll
    for (int q = 0; q < Q; ++q) a.fq[Q * e + q] = fv[q];
  }
}

template <typename T, typename I>
int run_source_eval(const void *coords, const void *conn, int64_t n_elems, int quad_order,
                    const tfem_source_program *program, void *fq, hipStream_t stream) {
  TriTables tables;
  if (!build_tri_tables(quad_order, int(sizeof(T)), &tables))
    return fail(TFEM_ERR_UNSUPPORTED, "Integration order not implemented");
  SrcEvalArgs<T, I> a;
  std::memset(&a, 0, sizeof(a));
  a.coords = static_cast<const T *>(coords);
  a.conn = static_cast<const I *>(conn);
  a.fq = static_cast<T *>(fq);
  a.n_elems = n_elems;
  for (int i = 0; i < 3; ++i)
    for (int q = 0; q < tables.nq; ++q) a.lam[i][q] = T(tables.lam[q][i]);
  const int st = src_convert<T>(program, &a.src);
  if (st != TFEM_OK) return st;
  if (n_elems == 0) return TFEM_OK;
  const dim3 grid(unsigned((n_elems + kSrcBlock - 1) / kSrcBlock)), block(kSrcBlock);
  switch (tables.nq) {
    case 1: hipLaunchKernelGGL((k_source_eval<T, I, 1>), grid, block, 0, stream, a); break;
    case 3: hipLaunchKernelGGL((k_source_eval<T, I, 3>), grid, block, 0, stream, a); break;
    case 4: hipLaunchKernelGGL((k_source_eval<T, I, 4>), grid, block, 0, stream, a); break;
    case 6: hipLaunchKernelGGL((k_source_eval<T, I, 6>), grid, block, 0, stream, a); break;
    default: return fail(TFEM_ERR_UNSUPPORTED, "Integration order not implemented");
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(TFEM_ERR_HIP, "source evaluation launch: %s", hipGetErrorString(e));
  return TFEM_OK;
}

}  // namespace
}  // namespace tfem

extern "C" {

int tfem_source_validate(const tfem_source_program *program) { return tfem::src_validate(program); }

int tfem_source_eval(const void *coords, int real_bytes, const void *conn, int idx_bytes,
                     int64_t n_elems, int64_t n_verts, int quad_order,
                     const tfem_source_program *program, void *fq, void *stream) {
  using namespace tfem;
  if ((real_bytes != 4 && real_bytes != 8) || (idx_bytes != 4 && idx_bytes != 8))
    return fail(TFEM_ERR_INVALID_ARGUMENT, "real_bytes / idx_bytes must be 4 or 8");
  if (n_elems < 0 || n_verts < 0 || (n_elems > 0 && (!coords || !conn || !fq)))
    return fail(TFEM_ERR_INVALID_ARGUMENT, "bad arguments");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (real_bytes == 8)
    return idx_bytes == 4 ? run_source_eval<double, int32_t>(coords, conn, n_elems, quad_order, program, fq, s)
                          : run_source_eval<double, int64_t>(coords, conn, n_elems, quad_order, program, fq, s);
  return idx_bytes == 4 ? run_source_eval<float, int32_t>(coords, conn, n_elems, quad_order, program, fq, s)
                        : run_source_eval<float, int64_t>(coords, conn, n_elems, quad_order, program, fq, s);
}

}  // extern "C"
