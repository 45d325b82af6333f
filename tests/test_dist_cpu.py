"""CPU, world_size 2 over gloo: the bucketed gradient exchange used for the N>1 path.

The GradReducer is model-agnostic, so it is exercised here with the oracle model (allowed in
tests/): two ranks each take half of a global batch; after `finish()` every rank must hold the
gradient of the mean loss over the GLOBAL batch, identical to a single-process run."""
import importlib
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make(seed=4):
    from oracle import hwgat_oracle as O
    cfg = dict(kp_dim=2, temporal_dim=8, num_classes=5, embed_dim=128, num_kps=32)
    params = O.synth_params(seed, **cfg)
    plist = {k: torch.nn.Parameter(v.clone(), requires_grad=k not in ("B", "pos_encoder.pe")) for k, v in params.items()}
    model = O.OracleHWGAT(plist, num_kps=32, temporal_dim=8)
    g = torch.Generator().manual_seed(1)
    x = torch.rand(4, 8, 32, 2, generator=g)
    y = torch.randint(0, 5, (4,), generator=g)
    return O, model, plist, x, y


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    dmod = importlib.import_module("sl-hwgat_amd.dist")
    O, model, plist, x, y = _make()
    if rank != 0:                                   # ranks start different; broadcast must fix it
        for p in plist.values():
            p.data.add_(1.0)
    dmod.broadcast_parameters(torch.nn.ParameterList(plist.values()))
    trainable = [p for p in plist.values() if p.requires_grad]
    red = dmod.GradReducer(trainable, bucket_bytes=1 << 20)
    assert len(red.buckets) > 3                     # several buckets -> overlap path exercised
    for step in range(2):                           # second step checks zero_grad/bucket reuse
        red.zero_grad()
        xs, ys = x[rank * 2:(rank + 1) * 2], y[rank * 2:(rank + 1) * 2]
        O.smoothed_cross_entropy(model.forward(xs), ys).backward()
        red.finish()
    torch.save({k: p.grad.clone() for k, p in plist.items() if p.grad is not None},
               os.path.join(out_dir, f"grads{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_bucketed_allreduce_matches_single_process(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    O, model, plist, x, y = _make()
    O.smoothed_cross_entropy(model.forward(x), y).backward()
    g0 = torch.load(os.path.join(tmp_path, "grads0.pt"))
    g1 = torch.load(os.path.join(tmp_path, "grads1.pt"))
    assert set(g0) == {k for k, p in plist.items() if p.grad is not None}
    for k, p in plist.items():
        if p.grad is None:
            continue
        assert torch.equal(g0[k], g1[k]), k                      # ranks agree bit for bit
        err = (g0[k] - p.grad).norm() / p.grad.norm().clamp_min(1e-12)
        assert err < 1e-5, (k, err)                              # == gradient of the global-batch mean loss


def test_single_process_reducer_is_a_noop_wrapper():
    dmod = importlib.import_module("sl-hwgat_amd.dist")
    lin = torch.nn.Linear(8, 4)
    red = dmod.GradReducer(lin.parameters(), bucket_bytes=64)
    red.zero_grad()
    lin(torch.ones(2, 8)).sum().backward()
    red.finish()
    assert torch.allclose(lin.weight.grad, torch.full((4, 8), 2.0))
    assert lin.weight.grad.data_ptr() >= red.buckets[-1]["flat"].data_ptr() or len(red.buckets) > 1
    red.zero_grad()
    assert float(lin.weight.grad.abs().sum()) == 0.0


def test_reducer_with_gradient_accumulation():
    dmod = importlib.import_module("sl-hwgat_amd.dist")
    lin = torch.nn.Linear(8, 4)
    red = dmod.GradReducer(lin.parameters(), bucket_bytes=64)
    red.zero_grad(n_accum=3)
    for _ in range(3):
        lin(torch.ones(2, 8)).sum().backward()
        assert all(b["pending"] >= 0 for b in red.buckets)
    assert all(b["pending"] == 0 for b in red.buckets)
    red.finish()
    assert torch.allclose(lin.weight.grad, torch.full((4, 8), 6.0))


def _run_bench(extra_env, *argv):
    import subprocess
    env = dict(os.environ, HWGAT_BENCH_DRYRUN="1", **extra_env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        if k not in extra_env:
            env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True,
                          text=True, timeout=240)


@pytest.mark.timeout(300)
def test_bench_parent_spawns_ranks_and_relays_one_json_line():
    """`python bench.py --gpus 2` with no launcher: the parent starts 2 rank processes (gloo dry run: no model, no
    GPU), relays rank 0's line, and reports the launcher-visible fields (n_gpus, global batch = 64 * N)."""
    import json
    res = _run_bench({}, "--gpus", "2", "--steps", "3", "--warmup", "1")
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1
    assert rec["config"]["global_batch"] == 128 and rec["config"]["parallelism"] == "dp2"
    assert abs(rec["max_elapsed"] - 0.002) < 1e-12          # MAX over ranks, not rank 0's own value


@pytest.mark.timeout(300)
def test_bench_parent_fails_when_a_rank_fails_or_the_launcher_disagrees():
    res = _run_bench({"HWGAT_BENCH_DRYRUN_FAIL_RANK": "1"}, "--gpus", "2")
    assert res.returncode != 0 and res.stdout.strip() == "" and "rank 1 exited with code 3" in res.stderr
    res = _run_bench({"WORLD_SIZE": "2", "RANK": "0"}, "--gpus", "4")
    assert res.returncode != 0 and "WORLD_SIZE=2" in res.stderr
