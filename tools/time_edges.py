"""Developer tool (GPU box): Basis.interpolate on the interior edges -- the
tfem_edge_interpolate_p1 launch against the torch-expression sequence of the reference
(basis.py:98-177), and the jump functional built on it."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytorch_fem_solver_amd as tf  # noqa: E402
from pytorch_fem_solver_amd import meshgen  # noqa: E402

torch.set_default_dtype(torch.float64)
torch.set_default_device("cuda")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 707
mesh_np = meshgen.unit_square(n, 0.25, 0)
mesh_np.pop("neighbors", None)  # edge -> cells by matching (see topology.cells_of_edges_by_matching)
t0 = time.perf_counter()
mesh = tf.MeshTri(mesh_np)
basis = tf.Basis(mesh, tf.ElementTri(1, 3))
edge_basis = tf.InteriorEdgesBasis(mesh, tf.ElementLine(1, 2))
n_edges, n_points = edge_basis.integration_points.shape[0], edge_basis.integration_points.shape[-2]
torch.cuda.synchronize()
print(f"S({n}): {mesh_np['triangles'].shape[0]} elements, {n_edges} interior edges, Q_e = {n_points}; "
      f"bases built in {time.perf_counter() - t0:.2f} s")
xy = torch.as_tensor(mesh_np["vertices"])
u = torch.sin(3 * xy[:, :1]) * torch.cos(2 * xy[:, 1:])


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        out = fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3, out


us_k, (val, grad) = timed(lambda: basis.interpolate(edge_basis, u), 50)
expressions = tf.Basis(mesh, tf.ElementTri(1, 3))
expressions.edge_kernel = False  # the reference's expression sequence, run by torch
us_t, (val_t, grad_t) = timed(lambda: expressions.interpolate(edge_basis, u), 5)
scale = float(grad_t.detach().abs().max())
# compulsory traffic per edge: 2 cell ids (16 B), 2 x 3 vertex ids (24 B), Q points (16 Q B),
# outputs 2 Q values + 4 gradient entries; vertex data comes from cache (each vertex ~6 edges)
algo = n_edges * (16 + 24 + 16 * n_points + 8 * (2 * n_points + 4)) + xy.shape[0] * 24
print(f"kernel (tfem_edge_interpolate_p1): {us_k:.1f} us = {algo / us_k / 1e3:.0f} GB/s on {algo / 1e6:.0f} MB algorithmic; "
      f"torch expressions: {us_t:.1f} us ({us_t / us_k:.0f}x); max difference value "
      f"{float((val - val_t.detach()).abs().max()):.2e}, gradient {float((grad - grad_t.detach()).abs().max()) / scale:.2e} (scaled)")
jump = (grad[:, 0] - grad[:, 1])
print(f"gradient jump over the edges: max {float(jump.abs().max()):.3e}")


def loss_step(b):
    u_var = u.clone().requires_grad_(True)
    v_, g_ = b.interpolate(edge_basis, u_var)
    ((g_[:, 0] - g_[:, 1]) ** 2).sum().backward()
    return u_var.grad


us_kb, gk = timed(lambda: loss_step(basis), 20)
us_tb, gt = timed(lambda: loss_step(expressions), 5)
print(f"forward + backward of a squared-jump loss: kernels {us_kb:.0f} us, torch expressions {us_tb:.0f} us "
      f"({us_tb / us_kb:.0f}x); max scaled difference of the nodal gradient "
      f"{float((gk - gt).abs().max() / gt.abs().max()):.2e}")
