// Source programs: the user's f(x, y) of a linear form `f(x_q) * v` (abstract_basis.py:95-112,
// tests/test_assembly.py:75-84) as a short postfix program, evaluated INSIDE the assembly
// kernels at the integration points x_q = bar(q)^T X (basis.py:90-91) instead of being read
// back from HBM as 8 Q bytes per element of pre-evaluated values.  The Python tracer
// (basis/forms.py) records the arithmetic the callable applies to the coordinate columns of
// `basis.integration_points`; include/tfem_assembly.h (tfem_source_program) is the format.
//
// Evaluation: a stack of four entries, each entry the values at the Q integration points of
// one element (so the program is decoded once per element, not once per point).  The program
// sits in registers, one operation per lane, and is fetched with v_readlane; every branch is
// wave-uniform.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "tfem_assembly.h"

namespace tfem {

constexpr int kSrcMaxOps = TFEM_SOURCE_MAX_OPS;
constexpr int kSrcStack = TFEM_SOURCE_STACK;

// Device-side copy of a tfem_source_program in the real type of the launch.
template <typename T>
struct SrcProgram {
  int n_ops;
  int pad;
  uint32_t opw[kSrcMaxOps / 4];  // four op codes per dword
  T c[kSrcMaxOps];               // constant operand of op i (unused: 0)
};

// Host: validates (stack discipline, known ops) and converts.  Returns TFEM_OK or fails.
template <typename T>
int src_convert(const tfem_source_program *in, SrcProgram<T> *out);
int src_validate(const tfem_source_program *in);
// Most values a valid program holds at any time (1 .. TFEM_SOURCE_STACK); 0 for an invalid one.
int src_depth(const tfem_source_program *in);

#if defined(__HIPCC__)

template <typename T>
__device__ __forceinline__ T src_fma(T a, T b, T c) {
  if constexpr (sizeof(T) == 8) return __builtin_fma(a, b, c);
  else return __builtin_fmaf(a, b, c);
}

// sin / cos in fp64 for |x| < 1e9: k = rint(x / pi) by adding 1.5 * 2^52 inside one fused
// multiply-add (the parity of k is then bit 0 of the sum's low word), r = x - k pi with pi in two
// doubles (both steps fused multiply-adds: the reduction is exact to ~1e-33 k), a degree-17
// polynomial on [-pi/2, pi/2], sign from the parity of k.  16 instructions per value against
// ~60 on the short path of the library function; accuracy against correctly rounded values:
// tests/test_hip_source.py::test_fast_sin_cos_against_correctly_rounded_values.
// Larger arguments take the library function (wave-uniform branch).
__device__ __forceinline__ double src_sin_poly(double r) {
  // r + r^3 P(r^2), P of degree 7: interpolation of (sin r - r) / r^3 at the Chebyshev nodes of
  // [0, (pi/2)^2] (computed in 60-digit arithmetic, rounded to double): |error| <= 3.6e-17 on
  // [-pi/2, pi/2] -- two terms fewer than the Taylor polynomial of the same accuracy (r^21)
  const double r2 = r * r;
  double p = 2.73144475909634011e-15;
  p = __builtin_fma(p, r2, -7.64397029616579265e-13);
  p = __builtin_fma(p, r2, 1.60589773124442683e-10);
  p = __builtin_fma(p, r2, -2.50521076169958777e-08);
  p = __builtin_fma(p, r2, 2.75573192191632300e-06);
  p = __builtin_fma(p, r2, -1.98412698412549741e-04);
  p = __builtin_fma(p, r2, 8.33333333333331587e-03);
  p = __builtin_fma(p, r2, -1.66666666666666657e-01);
  return __builtin_fma(r * r2, p, r);
}

// c * (r + r^3 P(r^2)) with the factor folded in: (c r) + (c r) r^2 P -- one multiplication instead
// of one for r^3 and one for the factor behind the function
__device__ __forceinline__ double src_sin_poly_scaled(double r, double c) {
  const double r2 = r * r;
  double p = 2.73144475909634011e-15;
  p = __builtin_fma(p, r2, -7.64397029616579265e-13);
  p = __builtin_fma(p, r2, 1.60589773124442683e-10);
  p = __builtin_fma(p, r2, -2.50521076169958777e-08);
  p = __builtin_fma(p, r2, 2.75573192191632300e-06);
  p = __builtin_fma(p, r2, -1.98412698412549741e-04);
  p = __builtin_fma(p, r2, 8.33333333333331587e-03);
  p = __builtin_fma(p, r2, -1.66666666666666657e-01);
  const double cr = c * r;
  return __builtin_fma(cr * r2, p, cr);
}

constexpr double kSrcPiHi = 3.141592653589793116e+00, kSrcPiLo = 1.224646799147353207e-16;
constexpr double kSrcInvPi = 0.318309886183790671537767526745;
constexpr double kSrcTrigFastMax = 1.0e9;
constexpr double kSrcRoundMagic = 6755399441055744.0;  // 1.5 * 2^52: x + magic rounds x to an integer

// s with its sign flipped when bit 0 of `parity_word` is set
__device__ __forceinline__ double src_flip_sign(double s, unsigned parity_word) {
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  u32x2 b = __builtin_bit_cast(u32x2, s);
  b.y ^= parity_word << 31;
  return __builtin_bit_cast(double, b);
}

// The low mantissa word of k + magic is k modulo 2^32 (two's complement), |k| < 2^51.
__device__ __forceinline__ unsigned src_low_word(double kd) {
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(u32x2, kd).x;
}

__device__ __forceinline__ double src_sin_fast(double x) {
  const double kd = __builtin_fma(x, kSrcInvPi, kSrcRoundMagic);  // k = rint(x / pi), one rounding
  const double k = kd - kSrcRoundMagic;
  double r = __builtin_fma(-k, kSrcPiHi, x);
  r = __builtin_fma(-k, kSrcPiLo, r);
  return src_flip_sign(src_sin_poly(r), src_low_word(kd));
}

__device__ __forceinline__ double src_sin_fast_scaled(double x, double c) {
  const double kd = __builtin_fma(x, kSrcInvPi, kSrcRoundMagic);
  const double k = kd - kSrcRoundMagic;
  double r = __builtin_fma(-k, kSrcPiHi, x);
  r = __builtin_fma(-k, kSrcPiLo, r);
  return src_flip_sign(src_sin_poly_scaled(r, c), src_low_word(kd));
}

__device__ __forceinline__ double src_cos_fast_scaled(double x, double c) {
  const double kd = __builtin_fma(x, kSrcInvPi, -0.5) + kSrcRoundMagic;
  const double kh = (kd - kSrcRoundMagic) + 0.5;
  double r = __builtin_fma(-kh, kSrcPiHi, x);
  r = __builtin_fma(-kh, kSrcPiLo, r);
  return src_flip_sign(src_sin_poly_scaled(r, c), ~src_low_word(kd));
}

__device__ __forceinline__ double src_cos_fast(double x) {
  // cos x = (-1)^(k+1) sin(x - (k + 1/2) pi), k = rint(x / pi - 1/2)
  const double kd = __builtin_fma(x, kSrcInvPi, -0.5) + kSrcRoundMagic;
  const double kh = (kd - kSrcRoundMagic) + 0.5;
  double r = __builtin_fma(-kh, kSrcPiHi, x);
  r = __builtin_fma(-kh, kSrcPiLo, r);
  return src_flip_sign(src_sin_poly(r), ~src_low_word(kd));
}

// A library function applied to every entry with ONE inlined copy of it: the loop is not
// unrolled, each pass handles entry 0 and rotates the array by one (static register indices
// only: a dynamic index would send the array to scratch memory).
#define TFEM_SRC_ROTATE_APPLY(v, fn)                    \
  _Pragma("unroll 1") for (int it = 0; it < QL; ++it) { \
    const T t0 = fn(v[0]);                              \
    _Pragma("unroll") for (int q = 0; q + 1 < QL; ++q) v[q] = v[q + 1]; \
    v[QL - 1] = t0;                                     \
  }

#define TFEM_SRC_WIDE_APPLY(v, fn)                     \
  _Pragma("unroll 1") for (int it = 0; it < N; ++it) { \
    const T t0 = fn(v[0]);                             \
    _Pragma("unroll") for (int i = 0; i + 1 < N; ++i) v[i] = v[i + 1]; \
    v[N - 1] = t0;                                     \
  }

template <typename T, int QL>
__device__ __forceinline__ void src_sin(T (&v)[QL]) {
  if constexpr (sizeof(T) == 8) {
    bool big = false;
#pragma unroll
    for (int q = 0; q < QL; ++q) big = big || !(__builtin_fabs(v[q]) < kSrcTrigFastMax);
    if (__builtin_amdgcn_ballot_w64(big) == 0) {
#pragma unroll
      for (int q = 0; q < QL; ++q) v[q] = src_sin_fast(v[q]);
    } else {
      TFEM_SRC_ROTATE_APPLY(v, sin)
    }
  } else {
#pragma unroll
    for (int q = 0; q < QL; ++q) v[q] = sinf(v[q]);
  }
}

template <typename T, int QL>
__device__ __forceinline__ void src_cos(T (&v)[QL]) {
  if constexpr (sizeof(T) == 8) {
    bool big = false;
#pragma unroll
    for (int q = 0; q < QL; ++q) big = big || !(__builtin_fabs(v[q]) < kSrcTrigFastMax);
    if (__builtin_amdgcn_ballot_w64(big) == 0) {
#pragma unroll
      for (int q = 0; q < QL; ++q) v[q] = src_cos_fast(v[q]);
    } else {
      TFEM_SRC_ROTATE_APPLY(v, cos)
    }
  } else {
#pragma unroll
    for (int q = 0; q < QL; ++q) v[q] = cosf(v[q]);
  }
}

// top = c * sin(top) / c * cos(top): the factor goes into the polynomial's last steps on the fast path
template <typename T, int QL>
__device__ __forceinline__ void src_sin_scaled(T (&v)[QL], T c) {
  if constexpr (sizeof(T) == 8) {
    bool big = false;
#pragma unroll
    for (int q = 0; q < QL; ++q) big = big || !(__builtin_fabs(v[q]) < kSrcTrigFastMax);
    if (__builtin_amdgcn_ballot_w64(big) == 0) {
#pragma unroll
      for (int q = 0; q < QL; ++q) v[q] = src_sin_fast_scaled(v[q], c);
      return;
    }
  }
  src_sin<T, QL>(v);
#pragma unroll
  for (int q = 0; q < QL; ++q) v[q] = c * v[q];
}

template <typename T, int QL>
__device__ __forceinline__ void src_cos_scaled(T (&v)[QL], T c) {
  if constexpr (sizeof(T) == 8) {
    bool big = false;
#pragma unroll
    for (int q = 0; q < QL; ++q) big = big || !(__builtin_fabs(v[q]) < kSrcTrigFastMax);
    if (__builtin_amdgcn_ballot_w64(big) == 0) {
#pragma unroll
      for (int q = 0; q < QL; ++q) v[q] = src_cos_fast_scaled(v[q], c);
      return;
    }
  }
  src_cos<T, QL>(v);
#pragma unroll
  for (int q = 0; q < QL; ++q) v[q] = c * v[q];
}

template <typename T>
using src_const_ptr = const SrcProgram<T> __attribute__((address_space(4))) *;

// The program at byte `offset` of the kernel-argument segment (the launch structure is the
// kernel's only parameter).
template <typename T>
__device__ __forceinline__ src_const_ptr<T> src_in_kernarg(size_t offset) {
  typedef const char __attribute__((address_space(4))) *kbytes;
  return (src_const_ptr<T>)((kbytes)__builtin_amdgcn_kernarg_segment_ptr() + offset);
}

// The program as every wave keeps it while the kernel runs: lane i holds operation i and its
// constant (3 VGPRs).  The interpreter fetches operation pc with v_readlane -- a few cycles --
// instead of scalar loads from the kernel-argument segment, whose latency (two dependent loads
// per operation) nothing would hide.
template <typename T>
struct SrcLanes {
  uint32_t op;
  T c;
  int n_ops;  // wave-uniform
};

template <typename T>
__device__ __forceinline__ SrcLanes<T> src_load_lanes(src_const_ptr<T> p) {
  const int lane = threadIdx.x & 63;
  const int i = lane & (kSrcMaxOps - 1);
  SrcLanes<T> r;
  r.op = (p->opw[i >> 2] >> (8 * (i & 3))) & 0xFFu;
  r.c = p->c[i];
  r.n_ops = p->n_ops;
  return r;
}

template <typename T>
__device__ __forceinline__ T src_lane_const(const SrcLanes<T> &prog, int pc) {
  if constexpr (sizeof(T) == 8) {
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 b = __builtin_bit_cast(u32x2, prog.c);
    return __builtin_bit_cast(double, u32x2{unsigned(__builtin_amdgcn_readlane(int(b.x), pc)),
                                            unsigned(__builtin_amdgcn_readlane(int(b.y), pc))});
  } else {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, prog.c), pc));
  }
}

// f at the QL points (x[q], y[q]) of one element -> out.  Wave-uniform control flow.
template <typename T, int QL>
__device__ __forceinline__ void src_run(const SrcLanes<T> &prog, const T (&x)[QL], const T (&y)[QL],
                                        T (&out)[QL]) {
  T s0[QL], s1[QL], s2[QL], s3[QL], lo[QL];
#pragma unroll
  for (int q = 0; q < QL; ++q) s0[q] = s1[q] = s2[q] = s3[q] = lo[q] = T(0);
  const int n = prog.n_ops;
  // Every operation leaves its result in s0; the stack moves (push: everything one down; binary:
  // the second operand out into `lo`, everything one up) happen in two places in front of the
  // switch, so that the switch's cases only differ in s0 -- with the moves inside the cases
  // hipcc copies all four entries between register sets where the cases join.
#define TFEM_SRC_SET(expr)             \
  _Pragma("unroll") for (int q = 0; q < QL; ++q) { \
    const T t = s0[q];                 \
    const T l = lo[q];                 \
    (void)t;                           \
    (void)l;                           \
    s0[q] = (expr);                    \
  }
#pragma unroll 1
  for (int pc = 0; pc < n; ++pc) {
    const uint32_t op = uint32_t(__builtin_amdgcn_readlane(int(prog.op), pc));
    const T c = src_lane_const<T>(prog, pc);
    if (op <= TFEM_SRC_PUSH_C) {
#pragma unroll
      for (int q = 0; q < QL; ++q) {
        s3[q] = s2[q];
        s2[q] = s1[q];
        s1[q] = s0[q];
      }
    } else if (op <= TFEM_SRC_DIV_R) {
#pragma unroll
      for (int q = 0; q < QL; ++q) {
        lo[q] = s1[q];
        s1[q] = s2[q];
        s2[q] = s3[q];
      }
    }
    switch (op) {
      case TFEM_SRC_PUSH_X: TFEM_SRC_SET(c * x[q]) break;
      case TFEM_SRC_PUSH_Y: TFEM_SRC_SET(c * y[q]) break;
      case TFEM_SRC_PUSH_C: TFEM_SRC_SET(c) break;
      case TFEM_SRC_ADD: TFEM_SRC_SET(l + t) break;
      case TFEM_SRC_SUB: TFEM_SRC_SET(l - t) break;
      case TFEM_SRC_SUB_R: TFEM_SRC_SET(t - l) break;
      case TFEM_SRC_MUL: TFEM_SRC_SET(l * t) break;
      case TFEM_SRC_DIV: TFEM_SRC_SET(l / t) break;
      case TFEM_SRC_DIV_R: TFEM_SRC_SET(t / l) break;
      case TFEM_SRC_ADD_C: TFEM_SRC_SET(t + c) break;
      case TFEM_SRC_MUL_C: TFEM_SRC_SET(t * c) break;
      case TFEM_SRC_RSUB_C: TFEM_SRC_SET(c - t) break;
      case TFEM_SRC_RDIV_C: TFEM_SRC_SET(c / t) break;
      case TFEM_SRC_NEG: TFEM_SRC_SET(-t) break;
      case TFEM_SRC_ABS:
        if constexpr (sizeof(T) == 8) { TFEM_SRC_SET(__builtin_fabs(t)) } else { TFEM_SRC_SET(__builtin_fabsf(t)) }
        break;
      case TFEM_SRC_POW_I: {  // t^n, n = 2 .. 8 by multiplications from the left (torch: x*x, x*x*x)
        const int e = int(c);
#pragma unroll
        for (int q = 0; q < QL; ++q) {
          const T t = s0[q];
          T r = t * t;
          for (int i = 2; i < e; ++i) r = r * t;
          s0[q] = r;
        }
        break;
      }
      // the functions: top = c * fn(top)
      case TFEM_SRC_SIN: src_sin_scaled<T, QL>(s0, c); break;
      case TFEM_SRC_COS: src_cos_scaled<T, QL>(s0, c); break;
      case TFEM_SRC_EXP:
        if constexpr (sizeof(T) == 8) { TFEM_SRC_SET(c * exp(t)) } else { TFEM_SRC_SET(c * expf(t)) }
        break;
      case TFEM_SRC_SQRT:
        if constexpr (sizeof(T) == 8) { TFEM_SRC_SET(c * sqrt(t)) } else { TFEM_SRC_SET(c * sqrtf(t)) }
        break;
      case TFEM_SRC_LOG:
        if constexpr (sizeof(T) == 8) { TFEM_SRC_ROTATE_APPLY(s0, log) } else { TFEM_SRC_ROTATE_APPLY(s0, logf) }
        TFEM_SRC_SET(c * t)
        break;
      case TFEM_SRC_TANH:
        if constexpr (sizeof(T) == 8) { TFEM_SRC_ROTATE_APPLY(s0, tanh) } else { TFEM_SRC_ROTATE_APPLY(s0, tanhf) }
        TFEM_SRC_SET(c * t)
        break;
      default: break;
    }
  }
#undef TFEM_SRC_SET
#pragma unroll
  for (int q = 0; q < QL; ++q) out[q] = s0[q];
}

// The same for NE elements at once, for programs that never hold more than TWO values (host:
// src_depth): every stack entry holds the NE * QL values of NE elements, so the program is
// decoded once per NE elements -- the decode (three v_readlane, a compare tree, the stack
// moves) costs about as much as four fp64 additions on QL = 4 values, and with one element per
// pass it was 40 % of the time of a sin * sin source.  Two entries of 12 doubles take the
// registers four entries of 4 plus the points took; the coordinates of the elements' vertices
// are read from LDS (`xyc`, tile-local ids in `codes`) when a PUSH needs them, not kept.
template <typename T, int QL, int NE>
__device__ __forceinline__ void src_run_wide(const SrcLanes<T> &prog, const T *xyc, const unsigned (&codes)[NE],
                                             const T (&lam)[3][kMaxQuad], T (&out)[NE * QL]) {
  constexpr int N = NE * QL;
  // no zero fill: a valid program (src_validate) writes an entry before it reads it; the empty
  // asm gives the registers a defined value without an instruction
  T s0[N], s1[N];
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("" : "=v"(s0[i]), "=v"(s1[i]));
  const int n = prog.n_ops;
  int depth = 0;  // entries on the stack (wave-uniform)
#define TFEM_SRC_SET(expr)             \
  _Pragma("unroll") for (int i = 0; i < N; ++i) { \
    const T t = s0[i];                 \
    const T l = s1[i];                 \
    (void)t;                           \
    (void)l;                           \
    s0[i] = (expr);                    \
  }
#pragma unroll 1
  for (int pc = 0; pc < n; ++pc) {
    const uint32_t op = uint32_t(__builtin_amdgcn_readlane(int(prog.op), pc));
    const T c = src_lane_const<T>(prog, pc);
    if (op <= TFEM_SRC_PUSH_C) {  // a push moves the top entry down -- when there is one
      if (depth > 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) s1[i] = s0[i];
      }
      depth = __builtin_amdgcn_readfirstlane(depth + 1);
    } else if (op <= TFEM_SRC_DIV_R) {
      depth = __builtin_amdgcn_readfirstlane(depth - 1);  // the two-operand operations leave one entry
    }
    switch (op) {
      case TFEM_SRC_PUSH_X:
      case TFEM_SRC_PUSH_Y: {
        const int comp = op == TFEM_SRC_PUSH_Y ? 1 : 0;  // wave-uniform
        // the LDS addresses are formed here, from the packed ids: formed once in front of the
        // program loop they cost nine registers the kernel does not have (they went to scratch:
        // +45 MB of writes per launch at 1e7 elements).  All reads are issued before the first use.
        // (The points keep the reference's form l^T X, basis.py:90-91: the 4-point rule's structure --
        // b S + d X_i -- would save nine operations per element, but sin(pi x) at the boundary x -> 1
        // amplifies a last-bit difference of x_q by 1 / (1 - x) ~ 1e4, and the load vector's entries
        // there left the 1e-12 entry-wise bound against the oracle: 2e-12 at 1e7 elements.)
        unsigned code[NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          code[e] = codes[e];
          asm volatile("" : "+v"(code[e]));
        }
        T vert[NE][3];
#pragma unroll
        for (int e = 0; e < NE; ++e)
#pragma unroll
          for (int k = 0; k < 3; ++k)  // one bit-field extract + one shift-add per vertex
            vert[e][k] = xyc[2 * __builtin_amdgcn_ubfe(code[e], 10 * k, 10) + comp];
#pragma unroll
        for (int e = 0; e < NE; ++e)
#pragma unroll
          for (int q = 0; q < QL; ++q)  // fused multiply-adds (the file is compiled without contraction)
            s0[e * QL + q] = c * src_fma<T>(lam[2][q], vert[e][2], src_fma<T>(lam[1][q], vert[e][1], lam[0][q] * vert[e][0]));
        break;
      }
      case TFEM_SRC_PUSH_C: TFEM_SRC_SET(c) break;
      case TFEM_SRC_ADD: TFEM_SRC_SET(l + t) break;
      case TFEM_SRC_SUB: TFEM_SRC_SET(l - t) break;
      case TFEM_SRC_SUB_R: TFEM_SRC_SET(t - l) break;
      case TFEM_SRC_MUL: TFEM_SRC_SET(l * t) break;
      case TFEM_SRC_DIV: TFEM_SRC_SET(l / t) break;
      case TFEM_SRC_DIV_R: TFEM_SRC_SET(t / l) break;
      case TFEM_SRC_ADD_C: TFEM_SRC_SET(t + c) break;
      case TFEM_SRC_MUL_C: TFEM_SRC_SET(t * c) break;
      case TFEM_SRC_RSUB_C: TFEM_SRC_SET(c - t) break;
      case TFEM_SRC_RDIV_C: TFEM_SRC_SET(c / t) break;
      case TFEM_SRC_NEG: TFEM_SRC_SET(-t) break;
      case TFEM_SRC_ABS:
        if constexpr (sizeof(T) == 8) { TFEM_SRC_SET(__builtin_fabs(t)) } else { TFEM_SRC_SET(__builtin_fabsf(t)) }
        break;
      case TFEM_SRC_POW_I: {
        const int e = int(c);
#pragma unroll
        for (int i = 0; i < N; ++i) {
          const T t = s0[i];
          T r = t * t;
          for (int k = 2; k < e; ++k) r = r * t;
          s0[i] = r;
        }
        break;
      }
#if defined(TFEM_SRC_ABL) && (TFEM_SRC_ABL & 8)
      case TFEM_SRC_SIN: TFEM_SRC_SET(c * t) break;
      case TFEM_SRC_COS: TFEM_SRC_SET(c * t) break;
#else
      case TFEM_SRC_SIN: src_sin_scaled<T, N>(s0, c); break;
      case TFEM_SRC_COS: src_cos_scaled<T, N>(s0, c); break;
#endif
      case TFEM_SRC_EXP:
        if constexpr (sizeof(T) == 8) { TFEM_SRC_WIDE_APPLY(s0, exp) } else { TFEM_SRC_WIDE_APPLY(s0, expf) }
        TFEM_SRC_SET(c * t)
        break;
      case TFEM_SRC_SQRT:
        if constexpr (sizeof(T) == 8) { TFEM_SRC_SET(c * sqrt(t)) } else { TFEM_SRC_SET(c * sqrtf(t)) }
        break;
      case TFEM_SRC_LOG:
        if constexpr (sizeof(T) == 8) { TFEM_SRC_WIDE_APPLY(s0, log) } else { TFEM_SRC_WIDE_APPLY(s0, logf) }
        TFEM_SRC_SET(c * t)
        break;
      case TFEM_SRC_TANH:
        if constexpr (sizeof(T) == 8) { TFEM_SRC_WIDE_APPLY(s0, tanh) } else { TFEM_SRC_WIDE_APPLY(s0, tanhf) }
        TFEM_SRC_SET(c * t)
        break;
      default: break;
    }
  }
#undef TFEM_SRC_SET
#pragma unroll
  for (int i = 0; i < N; ++i) out[i] = s0[i];
}

#endif  // __HIPCC__

}  // namespace tfem
