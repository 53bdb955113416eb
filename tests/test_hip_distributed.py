"""BASELINE config 4 on the GPU with more than one rank (-m gpu): N processes of
tests/dist/check_sharded_assembly.py under torch.distributed.run share the test box's one card and
exchange over gloo -- shards, ring plans with the interface tiles first, the two-stream step of
bench.py, pack / all-reduce / unpack -- and every rank compares what it holds with the operator of
the whole mesh (fp64, 1e-12).  Over RCCL (one rank per GPU) the same script runs with
--backend nccl; that needs more than one GPU and is the driver's scaling run."""

import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(REPO, "tests", "dist", "check_sharded_assembly.py")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(world, out_dir, *extra):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), SCRIPT, "--out-dir", str(out_dir), *extra]
    done = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=REPO)
    assert done.returncode == 0, done.stdout[-3000:] + done.stderr[-3000:]
    ranks = []
    for r in range(world):
        with open(os.path.join(out_dir, f"rank{r}.json")) as fh:
            ranks.append(json.load(fh))
    return ranks


@pytest.mark.parametrize("world,layout", [(2, "strips"), (3, "strips"), (2, "partition"), (4, "partition")])
def test_sharded_assembly_with_interface_tiles_first(world, layout, tmp_path):
    ranks = _run(world, tmp_path, "--layout", layout)
    for r in ranks:
        assert r["err_values"] <= 1e-12 and r["err_vector"] <= 1e-12, r
        assert 0 < r["priority_tiles"] < r["tiles"], r
        assert r["interface_entries"] > 0
    assert len({r["interface_entries"] for r in ranks}) == 1  # one global interface numbering


@pytest.mark.parametrize("world,layout,stepper", [(2, "partition", "sharded"), (3, "partition", "sharded-first"),
                                                  (2, "strips", "sharded")])
def test_sharded_steps_of_the_bench(world, layout, stepper, tmp_path):
    """parallel.ShardedSteps (the steps bench.py --gpus N times): rotating pairs, the exchange of a
    step beside the next step's launch; every pair ends with the operator of the whole mesh."""
    ranks = _run(world, tmp_path, "--layout", layout, "--stepper", stepper)
    for r in ranks:
        assert r["err_values"] <= 1e-12 and r["err_vector"] <= 1e-12 and r["mode"] == "eager", r


@pytest.mark.parametrize("stepper", ["graph", "graph-first"])
def test_steps_recorded_into_hip_graphs_over_rccl_one_rank(stepper, tmp_path):
    """The HIP-graph form of the steps needs RCCL (gloo cannot be captured) and the test box has one
    GPU: ONE rank over RCCL, the exchange carrying an arbitrary set of entries through pack ->
    all-reduce -> unpack inside the captured graph; replays and eager steps mixed; against the
    operator of the whole mesh."""
    ranks = _run(1, tmp_path, "--layout", "partition", "--backend", "nccl", "--stepper", stepper)
    assert ranks[0]["mode"] == "graph" and ranks[0]["err_values"] <= 1e-12 and ranks[0]["err_vector"] <= 1e-12, ranks


def test_bench_default_scaling_is_config_4_and_one_rank_over_rccl_takes_graphs(tmp_path):
    """bench.py's N > 1 defaults: --scaling strong (BASELINE config 4).  And the whole N > 1 machinery
    with ONE rank over RCCL (torch.distributed.run, world size 1 is treated as N = 1 by bench.py, so
    the check of the graph mode over RCCL is tests/dist/check_sharded_assembly.py above)."""
    sys.path.insert(0, REPO)
    import bench

    old = sys.argv
    try:
        sys.argv = ["bench.py", "--gpus", "8"]
        args = bench.parse()
    finally:
        sys.argv = old
    assert args.scaling == "strong" and args.step_mode == "auto" and not args.interface_first


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_runs_with_two_ranks(scaling, tmp_path):
    """The driver's multi-GPU command line (bench.py under torch.distributed.run), two ranks on the
    one card over gloo, small mesh: one JSON line from rank 0 with the contract's keys."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"),
           "--gpus", "2", "--steps", "20", "--warmup", "5", "--grid", "256", "--backend", "gloo", "--scaling", scaling]
    done = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=REPO)
    assert done.returncode == 0, done.stdout[-3000:] + done.stderr[-3000:]
    lines = [line for line in done.stdout.splitlines() if line.startswith('{"metric"')]
    assert len(lines) == 1, done.stdout[-2000:]
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in out, key
    assert out["n_gpus"] == 2 and out["steps"] == 20 and out["warmup"] == 5 and out["scaling"] == scaling
    assert out["value"] > 0 and out["config"]["interface_tiles_first"] is None  # one launch per step by default
    assert out["config"]["world_size"] == 2 and out["config"]["backend"] == "gloo" and len(out["config"]["devices"]) == 2
    assert out["config"]["step_mode"] == "eager" and out["config"]["host_us_per_step"] > 0
    n_elems = 2 * 256 * 256
    assert out["config"]["elements_per_gpu"] == (n_elems if scaling == "weak" else n_elems // 2)
