"""ctypes binding of libtfem_hip.so (include/tfem_assembly.h).

There is deliberately no fallback: if the library is missing, or a hot-path call is
made without a GPU, this module raises.  The assembly path of this package IS the HIP
library.
"""

from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_int, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
#: TFEM_HIP_LIB: developer switch (an alternative build of the same library, e.g. other flags)
LIB_PATH = os.environ.get("TFEM_HIP_LIB") or os.path.join(_HERE, "csrc", "libtfem_hip.so")

_lib = None

#: TFEM_ABI_VERSION of include/tfem_assembly.h this binding was written against
ABI_VERSION = 2

#: tfem_source_program (include/tfem_assembly.h)
SOURCE_MAX_OPS = 32
SOURCE_STACK = 4


class SourceProgram(ctypes.Structure):
    _fields_ = [
        ("n_ops", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
        ("ops", ctypes.c_uint8 * SOURCE_MAX_OPS),
        ("consts", ctypes.c_double * SOURCE_MAX_OPS),
    ]


#: every symbol include/tfem_assembly.h declares -> (restype, argtypes)
SIGNATURES = {
    "tfem_abi_version": (c_int, []),
    "tfem_status_string": (c_char_p, [c_int]),
    "tfem_last_error": (c_char_p, []),
    "tfem_device_count": (c_int, []),
    "tfem_quadrature_size": (c_int, [c_int]),
    "tfem_quadrature_rule": (c_int, [c_int, c_void_p, c_void_p]),
    "tfem_csr_symbolic_count": (
        c_int,
        [c_void_p, c_int, c_int64, c_int, c_int64, c_void_p, c_void_p],
    ),
    "tfem_csr_symbolic_fill": (
        c_int,
        [c_void_p, c_int, c_int64, c_int, c_int64, c_void_p, c_void_p, c_void_p],
    ),
    "tfem_csr_pattern_create": (c_int, [c_void_p, c_int, c_int64, c_int, c_int64, c_void_p, c_void_p]),
    "tfem_csr_pattern_export": (c_int, [c_void_p, c_void_p, c_void_p]),
    "tfem_csr_pattern_destroy": (None, [c_void_p]),
    "tfem_csr_symbolic_slots": (
        c_int, [c_void_p, c_int, c_int64, c_int, c_int64, c_void_p, c_void_p, c_void_p],
    ),
    "tfem_tri_geometry": (
        c_int,
        [c_void_p, c_int, c_void_p, c_int, c_int64, c_int64, c_int, c_int]
        + [c_void_p] * 5,
    ),
    "tfem_tri_bilinear_csr": (
        c_int,
        [c_void_p, c_int, c_void_p, c_int, c_int64, c_int64, c_int, c_int, c_double, c_double,
         c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int64, c_void_p],
    ),
    "tfem_tri_load_vector": (
        c_int,
        [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int64, c_int64, c_int, c_int, c_void_p,
         c_void_p, c_int64, c_void_p, c_int, c_int64, c_void_p],
    ),
    "tfem_reduce_scatter_bilinear": (
        c_int,
        [c_void_p, c_int, c_int64, c_int64, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p,
         c_int64, c_void_p],
    ),
    "tfem_reduce_scatter_linear": (
        c_int,
        [c_void_p, c_int, c_int64, c_int64, c_void_p, c_int64, c_int, c_int, c_void_p, c_int,
         c_void_p, c_int64, c_void_p],
    ),
    "tfem_reduce_functional": (
        c_int,
        [c_void_p, c_int, c_int64, c_int64, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p],
    ),
    "tfem_tile_plan_create": (
        c_int,
        [c_void_p, c_int, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
         c_int, c_void_p],
    ),
    "tfem_tile_plan_sizes": (c_int, [c_void_p, c_void_p]),
    "tfem_tile_plan_pack": (c_int, [c_void_p, c_void_p]),
    "tfem_tile_plan_destroy": (None, [c_void_p]),
    "tfem_tile_capacity": (c_int, [c_int]),
    "tfem_p1_assemble_tiles": (
        c_int,
        [c_void_p, c_int, c_int64, c_int, c_double, c_double, c_void_p, c_void_p, c_void_p,
         c_int64, c_void_p, c_int64, c_void_p, c_void_p],
    ),
    "tfem_ring_plan_create": (
        c_int,
        [c_void_p, c_int, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p],
    ),
    "tfem_ring_plan_create_priority": (
        c_int,
        [c_void_p, c_int, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p,
         c_void_p],
    ),
    "tfem_ring_plan_create_from_pattern": (
        c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p],
    ),
    "tfem_p1_assemble_rings_range": (
        c_int,
        [c_void_p, c_int, c_int64, c_int, c_double, c_double, c_void_p, c_void_p, c_void_p, c_int64,
         c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p],
    ),
    "tfem_ring_plan_sizes": (c_int, [c_void_p, c_void_p]),
    "tfem_ring_plan_pack": (c_int, [c_void_p, c_void_p]),
    "tfem_ring_plan_destroy": (None, [c_void_p]),
    "tfem_ring_capacity": (c_int, [c_int]),
    "tfem_p1_assemble_rings": (
        c_int,
        [c_void_p, c_int, c_int64, c_int, c_double, c_double, c_void_p, c_void_p, c_void_p,
         c_int64, c_void_p, c_int64, c_void_p, c_void_p],
    ),
    "tfem_p2_plan_create": (
        c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p],
    ),
    "tfem_p2_plan_sizes": (c_int, [c_void_p, c_void_p]),
    "tfem_p2_plan_pack": (c_int, [c_void_p, c_void_p]),
    "tfem_p2_plan_destroy": (None, [c_void_p]),
    "tfem_p2_assemble_rows": (
        c_int,
        [c_void_p, c_int, c_int, c_double, c_double, c_void_p, c_void_p, c_void_p, c_int64, c_void_p],
    ),
    "tfem_p2_load_rows": (
        c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p],
    ),
    "tfem_csr_spmv": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_void_p, c_void_p, c_void_p]),
    "tfem_edge_interpolate_p1": (
        c_int,
        [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p,
         c_void_p],
    ),
    "tfem_edge_interpolate_p1_backward": (
        c_int,
        [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p,
         c_int64, c_void_p],
    ),
    "tfem_edge_interpolate_p1_backward_rows": (
        c_int,
        [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p,
         c_void_p, c_void_p, c_int64, c_void_p],
    ),
    "tfem_edge_interpolate_p1_fracture": (
        c_int,
        [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64,
         c_int64, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p],
    ),
    "tfem_csr_gather_map": (c_int, [c_void_p, c_int64, c_int, c_int64, c_void_p, c_void_p]),
    "tfem_csr_gather": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "tfem_interface_pack": (
        c_int,
        [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64,
         c_void_p, c_int64, c_void_p],
    ),
    "tfem_interface_unpack": (
        c_int,
        [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64,
         c_void_p, c_void_p],
    ),
    "tfem_interface_pack_dense": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int64, c_void_p, c_void_p]),
    "tfem_source_validate": (c_int, [c_void_p]),
    "tfem_source_eval": (
        c_int,
        [c_void_p, c_int, c_void_p, c_int, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p],
    ),
    "tfem_p1_assemble_rings_source": (
        c_int,
        [c_void_p, c_int, c_int64, c_int, c_double, c_double, c_void_p, c_void_p, c_void_p,
         c_int64, c_void_p, c_int64, c_void_p, c_void_p],
    ),
    "tfem_p1_residual_local": (
        c_int,
        [c_void_p, c_int, c_void_p, c_int, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p,
         c_double, c_void_p, c_void_p],
    ),
    "tfem_p1_residual_backward": (
        c_int,
        [c_void_p, c_int, c_void_p, c_int, c_int64, c_int64, c_int, c_void_p, c_double, c_void_p,
         c_void_p, c_void_p],
    ),
    "tfem_csr_to_dense": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_void_p, c_void_p],
    ),
}


class NativeLibraryMissing(RuntimeError):
    pass


def load():
    """Load libtfem_hip.so (once).  Raises NativeLibraryMissing if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryMissing(
            f"{LIB_PATH} not found: the HIP assembly library is not built. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` at the repository root. "
            "There is no CPU fallback for the assembly path."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError = ABI mismatch, surface it
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.tfem_abi_version() != ABI_VERSION:
        raise NativeLibraryMissing(f"ABI version {lib.tfem_abi_version()} != {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def check(status: int):
    """Map a tfem_status to the exception the reference would raise at that point."""
    if status == 0:
        return
    lib = load()
    message = lib.tfem_last_error().decode() or lib.tfem_status_string(status).decode()
    if status == 2:  # TFEM_ERR_UNSUPPORTED <-> the reference's NotImplementedError sites
        raise NotImplementedError(message)
    if status in (1, 4):
        raise ValueError(message)
    raise RuntimeError(f"libtfem_hip: {message}")


def ptr(tensor):
    """Device/host address of a torch tensor (None -> NULL)."""
    return None if tensor is None else c_void_p(tensor.data_ptr())


def current_stream(device):
    import torch

    return c_void_p(torch.cuda.current_stream(device).cuda_stream)
