"""Launches the HIP assembly kernels for one (mesh, element) pair.

Owns the device-resident inputs (coordinates, connectivity), the symbolic CSR pattern
and the slot map, and hands torch tensors to libtfem_hip through the C ABI
(include/tfem_assembly.h).  There is no torch/CPU implementation of these operations in
this package: without the library or without a GPU every method raises.
"""

from __future__ import annotations

import ctypes
from ctypes import c_void_p

import numpy as np
import torch

from .. import _native
from ..sparse import CSRMatrix


class NoDeviceError(RuntimeError):
    pass


def _compute_device(home: torch.device) -> torch.device:
    if home.type == "cuda":
        return home
    if torch.cuda.is_available():
        return torch.device("cuda", torch.cuda.current_device())
    raise NoDeviceError(
        "torch_fem assembly runs on an MI355X through libtfem_hip and has no CPU "
        "fallback, but no GPU is visible to this process"
    )


def symbolic_host(conn_dof, n_dofs):
    """Host symbolic phase of the C ABI (replaces basis.py:64-85): CSR pattern of the
    operator and, per element entry, the CSR position it adds to.  numpy in, numpy out:
    rowptr int64 (N+1), colind int32 (nnz), slots int32 (E*n*n)."""
    lib = _native.load()
    conn = np.ascontiguousarray(np.asarray(conn_dof).astype(np.int32))
    conn = conn.reshape(-1, conn.shape[-1])
    e, n = conn.shape
    rowptr = np.zeros(int(n_dofs) + 1, dtype=np.int64)
    nnz = ctypes.c_int64(0)
    _native.check(
        lib.tfem_csr_symbolic_count(
            c_void_p(conn.ctypes.data), 4, e, n, int(n_dofs),
            c_void_p(rowptr.ctypes.data), ctypes.byref(nnz),
        )
    )
    colind = np.zeros(max(nnz.value, 1), dtype=np.int32)
    slots = np.zeros(max(e * n * n, 1), dtype=np.int32)
    _native.check(
        lib.tfem_csr_symbolic_fill(
            c_void_p(conn.ctypes.data), 4, e, n, int(n_dofs),
            c_void_p(rowptr.ctypes.data), c_void_p(colind.ctypes.data),
            c_void_p(slots.ctypes.data),
        )
    )
    return rowptr, colind[: nnz.value], slots[: e * n * n]


class AssemblyEngine:
    def __init__(self, coords, conn_geo, conn_dof, n_dofs, poly_order, quad_order, fracture=None):
        """coords (N_v,2) or (F,N_v,2); conn_geo (N_T,3) or (F,N_T,3) vertex ids (per mesh);
        conn_dof (E,n) global DoF ids; fracture = (pinv (F,2,3), det (F,1,1)) or None."""
        self.lib = _native.load()
        self.home = coords.device
        self.dtype = coords.dtype
        if self.dtype not in (torch.float64, torch.float32):
            raise TypeError(f"unsupported real dtype {self.dtype}")
        self.real_bytes = 8 if self.dtype == torch.float64 else 4
        self.poly_order = int(poly_order)
        self.quad_order = int(quad_order)
        self.n_quad = self.lib.tfem_quadrature_size(self.quad_order)
        if self.n_quad == 0:
            raise NotImplementedError("Integration order not implemented")
        self.n_dofs = int(n_dofs)
        self.lead_shape = tuple(conn_geo.shape[:-1])
        self.n_elems = int(np.prod(self.lead_shape)) if len(self.lead_shape) else 0
        self.n_local = int(conn_dof.shape[-1])
        self.n_fractures = int(coords.shape[0]) if coords.dim() == 3 else 0
        self.coords_per_mesh = int(coords.shape[-2])
        self._host_coords = coords
        self._host_conn_geo = conn_geo
        self._host_conn_dof = conn_dof.reshape(-1, self.n_local)
        self._host_fracture = fracture
        self._dev = None
        self._csr = None

    # ------------------------------------------------------------------ device state
    @property
    def device(self):
        return _compute_device(self.home)

    def _inputs(self):
        if self._dev is None:
            dev = self.device
            d = {
                "coords": self._host_coords.to(dev).contiguous(),
                "conn_geo": self._host_conn_geo.to(dev, torch.int32).contiguous(),
                "conn_dof": self._host_conn_dof.to(dev, torch.int32).contiguous(),
                "pinv": None,
                "fdet": None,
            }
            if self._host_fracture is not None:
                pinv, det = self._host_fracture
                d["pinv"] = pinv.to(dev, self.dtype).contiguous()
                d["fdet"] = det.to(dev, self.dtype).reshape(-1).contiguous()
            self._dev = d
        return self._dev

    def _stream(self):
        return _native.current_stream(self.device)

    def _home(self, tensor):
        return tensor if tensor.device == self.home else tensor.to(self.home)

    # ------------------------------------------------------------------ symbolic phase
    def csr_structure(self):
        """(rowptr int64, colind int32, slots int32 (E,n,n)) on the compute device."""
        if self._csr is None:
            conn = self._host_conn_dof.cpu().numpy()
            rowptr, colind, slots = symbolic_host(conn, self.n_dofs)
            dev = self.device
            self._csr = (
                torch.from_numpy(rowptr).to(dev),
                torch.from_numpy(colind).to(dev),
                torch.from_numpy(slots).to(dev),
            )
        return self._csr

    def kernel_name(self):
        """Name of the dominant numeric kernel as rocprofv3 reports it."""
        return "k_p1_bilinear_atomic" if self.poly_order == 1 else "k_p2_bilinear_atomic"

    def wrap_csr(self, vals):
        rowptr, colind, _ = self.csr_structure()
        return CSRMatrix(rowptr, colind, vals, (self.n_dofs, self.n_dofs))

    # ------------------------------------------------------------------ kernels
    def geometry(self):
        """v_grad, dx, points, inv_jac in flat layouts (see tfem_tri_geometry)."""
        d = self._inputs()
        dev, e, q = self.device, self.n_elems, self.n_quad
        n = 3 if self.poly_order == 1 else 6
        vg_shape = (e, 3, 2) if self.poly_order == 1 else (e, q, 6, 2)
        v_grad = torch.empty(vg_shape, dtype=self.dtype, device=dev)
        dx = torch.empty((e, q), dtype=self.dtype, device=dev)
        points = torch.empty((e, q, 2), dtype=self.dtype, device=dev)
        inv = torch.empty((e, 2, 2), dtype=self.dtype, device=dev)
        with torch.cuda.device(dev):
            # per-mesh vertex ids: offset the coordinates per fracture on the host side of
            # the ABI by flattening (F, N_v, 2) and shifting ids once
            conn = d["conn_geo"]
            coords = d["coords"]
            if self.n_fractures:
                shift = (torch.arange(self.n_fractures, device=dev, dtype=torch.int32)
                         * self.coords_per_mesh)[:, None, None]
                conn = (conn + shift).contiguous()
            _native.check(
                self.lib.tfem_tri_geometry(
                    _native.ptr(coords), self.real_bytes, _native.ptr(conn), 4, e,
                    coords.numel() // 2, self.poly_order, self.quad_order,
                    _native.ptr(v_grad), _native.ptr(dx), _native.ptr(points), _native.ptr(inv),
                    self._stream(),
                )
            )
        return v_grad, dx, points, inv

    def bilinear(self, alpha: float, beta: float):
        """CSR values of alpha*stiffness + beta*mass (fused kernel)."""
        d = self._inputs()
        _, colind, slots = self.csr_structure()
        nnz = int(colind.shape[0])
        vals = torch.empty(nnz, dtype=self.dtype, device=self.device)
        with torch.cuda.device(self.device):
            _native.check(
                self.lib.tfem_tri_bilinear_csr(
                    _native.ptr(d["coords"]), self.real_bytes, _native.ptr(d["conn_geo"]), 4,
                    self.n_elems, self.coords_per_mesh, self.poly_order, self.quad_order,
                    float(alpha), float(beta), _native.ptr(slots), _native.ptr(vals), nnz,
                    _native.ptr(d["pinv"]), _native.ptr(d["fdet"]), self.n_fractures,
                    self.coords_per_mesh, self._stream(),
                )
            )
        return vals

    def load(self, fq):
        """(N_dof,) vector of sum_q f_q phi_i dx_q; fq is (E, Q) on any device."""
        d = self._inputs()
        fq = fq.to(self.device, self.dtype).reshape(self.n_elems, self.n_quad).contiguous()
        out = torch.empty(self.n_dofs, dtype=self.dtype, device=self.device)
        with torch.cuda.device(self.device):
            _native.check(
                self.lib.tfem_tri_load_vector(
                    _native.ptr(d["coords"]), self.real_bytes, _native.ptr(d["conn_geo"]),
                    _native.ptr(d["conn_dof"]), 4, self.n_elems, self.coords_per_mesh,
                    self.poly_order, self.quad_order, _native.ptr(fq), _native.ptr(out),
                    self.n_dofs, _native.ptr(d["fdet"]), self.n_fractures, self.coords_per_mesh,
                    self._stream(),
                )
            )
        return out

    def _flatten_integrand(self, integrand, dx_shape, inner):
        """Broadcast against dx (..., Q, 1, 1) and view as (E, Q, *inner) with the inner
        block contiguous; returns (tensor, element stride, quadrature stride)."""
        integrand = integrand.to(self.device, self.dtype)
        full = torch.broadcast_shapes(tuple(integrand.shape), tuple(dx_shape))
        if tuple(full[-2:]) != tuple(inner) or full[-3] != self.n_quad:
            raise ValueError(
                f"integrand broadcasts to {tuple(full)}, expected (..., {self.n_quad}, {inner[0]}, {inner[1]})"
            )
        expanded = integrand.expand(full)
        target = (self.n_elems, self.n_quad) + tuple(inner)
        try:
            flat = expanded.view(target)
        except RuntimeError:
            flat = expanded.contiguous().view(target)
        if flat.stride(-1) != 1 or (inner[0] > 1 and flat.stride(-2) != inner[1]):
            flat = flat.contiguous()
        return flat, flat.stride(0), flat.stride(1)

    def reduce_bilinear(self, integrand, dx):
        n = self.n_local
        flat, es, qs = self._flatten_integrand(integrand, dx.shape, (n, n))
        dxf = dx.to(self.device, self.dtype).reshape(self.n_elems, self.n_quad).contiguous()
        _, colind, slots = self.csr_structure()
        nnz = int(colind.shape[0])
        vals = torch.empty(nnz, dtype=self.dtype, device=self.device)
        with torch.cuda.device(self.device):
            _native.check(
                self.lib.tfem_reduce_scatter_bilinear(
                    _native.ptr(flat), self.real_bytes, es, qs, _native.ptr(dxf), self.n_elems,
                    self.n_quad, n, _native.ptr(slots), _native.ptr(vals), nnz, self._stream(),
                )
            )
        return vals

    def reduce_linear(self, integrand, dx):
        n = self.n_local
        d = self._inputs()
        flat, es, qs = self._flatten_integrand(integrand, dx.shape, (n, 1))
        dxf = dx.to(self.device, self.dtype).reshape(self.n_elems, self.n_quad).contiguous()
        out = torch.empty(self.n_dofs, dtype=self.dtype, device=self.device)
        with torch.cuda.device(self.device):
            _native.check(
                self.lib.tfem_reduce_scatter_linear(
                    _native.ptr(flat), self.real_bytes, es, qs, _native.ptr(dxf), self.n_elems,
                    self.n_quad, n, _native.ptr(d["conn_dof"]), 4, _native.ptr(out), self.n_dofs,
                    self._stream(),
                )
            )
        return out

    def reduce_functional(self, integrand, dx):
        integrand = integrand.to(self.device, self.dtype)
        full = torch.broadcast_shapes(tuple(integrand.shape), tuple(dx.shape))
        if full[-1] != 1:
            raise NotImplementedError("functional integrands with a trailing dimension > 1")
        n_inner = int(full[-2])
        flat, es, qs = self._flatten_integrand(integrand, dx.shape, (n_inner, 1))
        dxf = dx.to(self.device, self.dtype).reshape(self.n_elems, self.n_quad).contiguous()
        out = torch.empty(self.n_elems, dtype=self.dtype, device=self.device)
        with torch.cuda.device(self.device):
            _native.check(
                self.lib.tfem_reduce_functional(
                    _native.ptr(flat), self.real_bytes, es, qs, _native.ptr(dxf), self.n_elems,
                    self.n_quad, n_inner, _native.ptr(out), self._stream(),
                )
            )
        return out.reshape(tuple(full[:-3]) + (1,))
