// The instantiations of k_p1_rings (tfem_rings_kernel.hpp) that evaluate a source program in
// the launch (SRC = true).  A translation unit of its own because it is built with
// `-mllvm -disable-machine-licm` (__graft_entry__.py): inside the interpreter loop of
// tfem_source.hpp hipcc's machine-level loop-invariant code motion hoists the materialisation of
// every floating-point literal of every operation (polynomial coefficients of sin, cos, exp,
// log ...) out of the loop and keeps them all in registers at once -- 114 VGPRs for a ONE-point
// interpreter against 38 without the pass; the fused launch would drop from 3 to 2 workgroups
// per CU.  The other kernels keep the default pipeline.
#include "tfem_rings_kernel.hpp"

namespace tfem {

#ifdef TFEM_SRC_BENCH_ONLY
// developer build (tools/ablate_src.py): the one instantiation bench.py's step takes
template <typename T>
void *pick_ring_src_kernel(int slots, bool mass, bool chunk, int nq, bool kmat, bool wide) {
  if constexpr (sizeof(T) == 8) {
    if (slots == 7 && !mass && chunk && nq == 4 && kmat && wide)
      return reinterpret_cast<void *>(k_p1_rings<T, 7, false, true, 4, false, true, 2>);
    if (slots == 7 && !mass && chunk && nq == 4 && !kmat && wide)
      return reinterpret_cast<void *>(k_p1_rings<T, 7, false, true, 4, false, false, 2>);
  }
  return nullptr;
}
template void *pick_ring_src_kernel<double>(int, bool, bool, int, bool, bool);
template void *pick_ring_src_kernel<float>(int, bool, bool, int, bool, bool);
}  // namespace tfem
#else

template <typename T, int SLOTS, bool CHUNK, int SRC>
static void *pick_src_load_only(int nq) {
  switch (nq) {
    case 1: return reinterpret_cast<void *>(k_p1_rings<T, SLOTS, false, CHUNK, 1, false, false, SRC>);
    case 3: return reinterpret_cast<void *>(k_p1_rings<T, SLOTS, false, CHUNK, 3, false, false, SRC>);
    case 4: return reinterpret_cast<void *>(k_p1_rings<T, SLOTS, false, CHUNK, 4, false, false, SRC>);
    case 6: return reinterpret_cast<void *>(k_p1_rings<T, SLOTS, false, CHUNK, 6, false, false, SRC>);
    default: return nullptr;
  }
}

template <typename T, int SLOTS, bool MASS, bool CHUNK, int SRC>
static void *pick_src_q(int nq) {
  switch (nq) {
    case 1: return reinterpret_cast<void *>(k_p1_rings<T, SLOTS, MASS, CHUNK, 1, false, true, SRC>);
    case 3: return reinterpret_cast<void *>(k_p1_rings<T, SLOTS, MASS, CHUNK, 3, false, true, SRC>);
    case 4: return reinterpret_cast<void *>(k_p1_rings<T, SLOTS, MASS, CHUNK, 4, false, true, SRC>);
    case 6: return reinterpret_cast<void *>(k_p1_rings<T, SLOTS, MASS, CHUNK, 6, false, true, SRC>);
    default: return nullptr;
  }
}

template <typename T, int SLOTS, bool CHUNK, int SRC>
static void *pick_src_mass(bool kmat, bool mass, int nq) {
  if (!kmat) return pick_src_load_only<T, SLOTS, CHUNK, SRC>(nq);
  return mass ? pick_src_q<T, SLOTS, true, CHUNK, SRC>(nq) : pick_src_q<T, SLOTS, false, CHUNK, SRC>(nq);
}

template <typename T, int SRC>
static void *pick_src_slots(int slots, bool mass, bool chunk, int nq, bool kmat) {
  if (slots == 7)
    return chunk ? pick_src_mass<T, 7, true, SRC>(kmat, mass, nq) : pick_src_mass<T, 7, false, SRC>(kmat, mass, nq);
  return chunk ? pick_src_mass<T, 15, true, SRC>(kmat, mass, nq) : pick_src_mass<T, 15, false, SRC>(kmat, mass, nq);
}

template <typename T>
void *pick_ring_src_kernel(int slots, bool mass, bool chunk, int nq, bool kmat, bool wide) {
  // the wide interpreter with Q = 6 would hold 2 x 18 doubles: it keeps the one-element form
  if (wide && nq <= 4) return pick_src_slots<T, 2>(slots, mass, chunk, nq, kmat);
  return pick_src_slots<T, 1>(slots, mass, chunk, nq, kmat);
}

template void *pick_ring_src_kernel<double>(int, bool, bool, int, bool, bool);
template void *pick_ring_src_kernel<float>(int, bool, bool, int, bool, bool);

}  // namespace tfem
#endif  // TFEM_SRC_BENCH_ONLY
