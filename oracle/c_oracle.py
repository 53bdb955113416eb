"""ctypes loader of the C/OpenMP oracle (oracle/assembly_oracle.c) -- test infrastructure.

Same rules as oracle/assembly_oracle.py: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module.
"""

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle_assembly.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(f"{LIB_PATH}: run __graft_entry__.build_oracle()")
        lib = ctypes.CDLL(LIB_PATH)
        lib.oracle_threads.restype = ctypes.c_int
        lib.oracle_p1_local.restype = ctypes.c_int
        lib.oracle_p1_points.restype = ctypes.c_int
        _lib = lib
    return _lib


def _p(a):
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


def threads():
    return load().oracle_threads()


def set_threads(n):
    load().oracle_set_threads(int(n))


def p1_local(vertices, triangles, order, alpha=1.0, beta=0.0, fq=None, want_k=True):
    """K_local (N_T,3,3) and f_local (N_T,3) (None when fq is None)."""
    lib = load()
    v = np.ascontiguousarray(vertices, dtype=np.float64)
    t = np.ascontiguousarray(triangles, dtype=np.int32)
    n = t.shape[0]
    k = np.empty((n, 3, 3)) if want_k else None
    f = np.empty((n, 3)) if fq is not None else None
    fq_c = None if fq is None else np.ascontiguousarray(fq, dtype=np.float64)
    nq = lib.oracle_p1_local(_p(v), _p(t), ctypes.c_int64(n), int(order), ctypes.c_double(alpha),
                             ctypes.c_double(beta), _p(fq_c), _p(k), _p(f))
    if nq == 0:
        raise NotImplementedError("Integration order not implemented")
    return k, f


def points(vertices, triangles, order):
    lib = load()
    v = np.ascontiguousarray(vertices, dtype=np.float64)
    t = np.ascontiguousarray(triangles, dtype=np.int32)
    nq = {1: 1, 2: 3, 3: 4, 4: 6}[order]
    out = np.empty((t.shape[0], nq, 2))
    lib.oracle_p1_points(_p(v), _p(t), ctypes.c_int64(t.shape[0]), int(order), _p(out))
    return out


def scatter_csr(k_local, slots, nnz):
    lib = load()
    vals = np.empty(nnz)
    s = np.ascontiguousarray(slots, dtype=np.int32).reshape(-1)
    lib.oracle_scatter_csr(_p(np.ascontiguousarray(k_local)), _p(s), ctypes.c_int64(s.size), _p(vals),
                           ctypes.c_int64(nnz))
    return vals


def scatter_vector(f_local, triangles, n_dofs):
    lib = load()
    f = np.empty(n_dofs)
    t = np.ascontiguousarray(triangles, dtype=np.int32).reshape(-1)
    lib.oracle_scatter_vector(_p(np.ascontiguousarray(f_local)), _p(t), ctypes.c_int64(t.size), _p(f),
                              ctypes.c_int64(n_dofs))
    return f
