"""SURVEY 8(f) f-4 on the device (-m gpu): the symbolic phase (CSR pattern, element -> entry map)
built from a device-resident connectivity against the host builders byte for byte, and the
device-resident mesh topology / P2 numbering against the fixture the reference generated."""

import numpy as np
import pytest
import torch

from conftest import load_golden, mesh_from_golden
from oracle import assembly_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _gpu_defaults():
    assert torch.cuda.is_available()
    torch.set_default_dtype(torch.float64)
    torch.set_default_device("cuda")
    yield
    torch.set_default_device("cpu")
    torch.set_default_dtype(torch.float32)


def _meshes():
    from pytorch_fem_solver_amd import meshgen

    d = meshgen.delaunay_square(30000, 5)
    yield "structured", meshgen.unit_square(150, 0.25, 0)
    yield "delaunay", d
    yield "delaunay_morton", meshgen.permute_mesh(d, vertex_order=meshgen.morton_order(d["vertices"]))
    holes = meshgen.unit_square(60, 0.25, 1)
    keep = np.random.default_rng(3).random(holes["triangles"].shape[0]) > 0.2  # open fans, isolated vertices
    yield "with_holes", dict(holes, triangles=np.ascontiguousarray(holes["triangles"][keep]))


@pytest.mark.parametrize("idx", [torch.int32, torch.int64])
def test_device_pattern_and_slots_equal_the_host_builders(idx):
    from pytorch_fem_solver_amd.basis.engine import symbolic_host
    from pytorch_fem_solver_amd.basis.symbolic_device import pattern_device, slots_device
    from pytorch_fem_solver_amd import dofs

    for name, mesh in _meshes():
        nv = mesh["vertices"].shape[0]
        cases = [("P1", mesh["triangles"], nv)]
        if name != "with_holes":  # P2: vertices, then edges -- numbered on the device, equal to the host numbering
            conn6_h, coords_h, _ = dofs.p2_dofs_numpy(mesh["vertices"], mesh["triangles"], mesh["edges"],
                                                      mesh["edge_markers"], mesh["vertex_markers"])
            dev = lambda a: torch.as_tensor(np.ascontiguousarray(a), device="cuda")  # noqa: E731
            conn6_d, coords_d, _ = dofs.p2_dofs_torch(dev(mesh["vertices"]), dev(mesh["triangles"]), dev(mesh["edges"]),
                                                      dev(mesh["edge_markers"]), dev(mesh["vertex_markers"]))
            assert conn6_d.is_cuda and np.array_equal(conn6_d.cpu().numpy(), conn6_h)
            assert np.array_equal(coords_d.cpu().numpy(), coords_h)
            cases.append(("P2", conn6_h, coords_h.shape[0]))
        for kind, conn_np, n in cases:
            rowptr_h, colind_h, slots_h = symbolic_host(conn_np, n)
            conn = torch.as_tensor(np.ascontiguousarray(conn_np), device="cuda").to(idx)
            rowptr, colind = pattern_device(conn, n)
            assert rowptr.is_cuda and rowptr.dtype == torch.int64 and colind.dtype == torch.int32
            assert np.array_equal(rowptr.cpu().numpy(), rowptr_h), (name, kind)
            assert np.array_equal(colind.cpu().numpy(), colind_h), (name, kind)
            slots = slots_device(conn, n, rowptr, colind)
            assert np.array_equal(slots.cpu().numpy().reshape(-1), slots_h), (name, kind)
            # and both are the oracle's pattern (numpy, from the index tensors of basis.py:64-85)
            o_rowptr, o_colind, o_slots = orc.csr_pattern(conn_np, n)
            assert np.array_equal(rowptr_h, o_rowptr) and np.array_equal(colind_h, o_colind)
            assert np.array_equal(slots_h.reshape(o_slots.shape), o_slots)


def test_device_pattern_at_full_size_against_the_host_builder():
    """S(2236) = 9,999,392 elements: 35,011,289 entries, the same bytes, and the time of both."""
    import time

    from pytorch_fem_solver_amd import meshgen
    from pytorch_fem_solver_amd.basis.engine import pattern_host
    from pytorch_fem_solver_amd.basis.symbolic_device import pattern_device

    mesh = meshgen.unit_square(2236, 0.25, 0)
    nv = mesh["vertices"].shape[0]
    conn = torch.as_tensor(mesh["triangles"], device="cuda")
    pattern_device(conn[:1000], nv)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rowptr, colind = pattern_device(conn, nv)
    torch.cuda.synchronize()
    t_dev = time.perf_counter() - t0
    t0 = time.perf_counter()
    rowptr_h, colind_h = pattern_host(mesh["triangles"], nv)
    t_host = time.perf_counter() - t0
    assert np.array_equal(rowptr.cpu().numpy(), rowptr_h) and np.array_equal(colind.cpu().numpy(), colind_h)
    print(f"CSR pattern of S(2236): device {t_dev * 1e3:.1f} ms, host builder {t_host * 1e3:.1f} ms")


def test_device_resident_topology_against_the_reference_fixture():
    """MeshTri on a device-resident triangulation: every topology array the reference derives
    (abstract_mesh.py:104-309) is computed by device code (torch sort / unique / search on the
    GPU, mesh/topology.py) and equals the fixture the reference itself produced."""
    import pytorch_fem_solver_amd as tf

    d = load_golden("mesh_topology_n4.npz")
    mesh = tf.MeshTri(triangulation=mesh_from_golden(d))  # default device cuda: the tensors live there
    checked = 0
    assert mesh["cells", "coordinates"].is_cuda
    assert np.array_equal(mesh["cells", "coordinates"].cpu().numpy(), d["out_cells_coordinates"])
    for group in ("interior_edges", "boundary_edges"):
        for key, value in mesh[group].items():
            assert value.is_cuda, (group, key)
            got, want = value.cpu().numpy(), d[f"out_{group}_{key}"]
            assert got.shape == want.shape and got.dtype == want.dtype, (group, key)
            if np.issubdtype(want.dtype, np.integer):
                assert np.array_equal(got, want), (group, key)
            else:  # lengths and normals: the device's sqrt / division round like the host's to 1 ulp
                assert np.abs(got - want).max() <= 4e-16 * max(1.0, np.abs(want).max()), (group, key)
            checked += 1
    assert np.abs(mesh["cells", "length"].cpu().numpy() - d["out_cells_length"]).max() <= 4e-16
    assert checked >= 8
