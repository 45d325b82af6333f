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


class _StandIn(torch.nn.Module):
    """small torch model in place of the HIP one (which has no CPU path): same per-clip independence, several layers
    so that the gradients fill several buckets in backward order"""

    def __init__(self):
        super().__init__()
        torch.manual_seed(3)
        self.l0 = torch.nn.Linear(16, 48)
        self.l1 = torch.nn.Linear(48, 48)
        self.l2 = torch.nn.Linear(48, 48)
        self.head = torch.nn.Linear(48, 5)
        self.frozen = torch.nn.Parameter(torch.ones(3), requires_grad=False)      # like the Fourier matrix `B`

    def forward(self, x):
        h = torch.nn.functional.gelu(self.l0(x))
        h = h + torch.nn.functional.gelu(self.l1(h))
        h = h + torch.nn.functional.gelu(self.l2(h))
        return self.head(h.mean(1))


def _standin_batch():
    g = torch.Generator().manual_seed(9)
    return torch.rand(8, 6, 16, generator=g), torch.randint(0, 5, (8,), generator=g)


def _trainstep_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    dmod = importlib.import_module("sl-hwgat_amd.dist")
    tmod = importlib.import_module("sl-hwgat_amd.train")
    model = _StandIn()
    if rank != 0:
        for p in model.parameters():
            p.data.mul_(1.5)                                   # ranks start apart; broadcast_parameters must fix it
    dmod.broadcast_parameters(model)
    assert model.rank_salt == rank                              # every rank its own dropout stream
    red = dmod.GradReducer(model.parameters(), bucket_bytes=4 << 10)
    assert len(red.buckets) >= 3
    red.trace = trace = []
    # fires when the gradient of the FIRST layer lands, i.e. at the very end of a backward pass (registered after
    # the reducer's own hook on the same parameter, so it runs after it)
    model.l0.weight.register_post_accumulate_grad_hook(lambda _p: trace.append(("l0.weight grad", None, None)))
    opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=5e-4)
    step = tmod.TrainStep(model, opt, red, micro_batch=2)
    x, y = _standin_batch()
    losses = []
    for _ in range(2):                                          # second step: bucket reuse with accumulation
        del trace[:]
        step(x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4])
        losses.append(float(step.loss))
        # two micro-batches of 2 clips -> two backward passes; collectives go out in the LAST one only, from the hooks,
        # and the buckets of the late layers are issued BEFORE backward has reached the first layer
        ends = [i for i, e in enumerate(trace) if e[0] == "l0.weight grad"]
        launches = [i for i, e in enumerate(trace) if e[0] == "launch"]
        assert len(ends) == 2 and len(launches) == len(red.buckets), trace
        assert all(trace[i][2] == "hook" for i in launches), trace          # none left for finish()
        assert all(i > ends[0] for i in launches), trace                    # nothing reduced after the first pass
        assert sum(i < ends[1] for i in launches) >= len(red.buckets) - 1, trace   # overlap with the rest of backward
    torch.save({"params": {k: v.detach().clone() for k, v in model.state_dict().items()}, "losses": losses},
               os.path.join(out_dir, f"ts{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_trainstep_with_micro_batches_equals_single_process_and_overlaps(tmp_path):
    """the real train.TrainStep (micro-batch 2) + GradReducer + broadcast_parameters on 2 gloo ranks: after two
    optimizer steps both ranks hold the parameters a single process gets from the global batch, and every bucket's
    all-reduce was issued from inside backward (all but the last before backward reached the first layer)."""
    port = _free_port()
    mp.spawn(_trainstep_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    tmod = importlib.import_module("sl-hwgat_amd.train")
    model = _StandIn()
    opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=5e-4)
    step = tmod.TrainStep(model, opt, None, micro_batch=2)
    x, y = _standin_batch()
    for _ in range(2):
        step(x, y)
    r0 = torch.load(os.path.join(tmp_path, "ts0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "ts1.pt"))
    for k, v in model.state_dict().items():
        assert torch.equal(r0["params"][k], r1["params"][k]), k                # ranks stay bit-identical
        assert torch.allclose(r0["params"][k], v, rtol=0, atol=2e-6), (k, (r0["params"][k] - v).abs().max())
    # each rank reports the mean loss of ITS shard; their mean is the global-batch loss
    assert abs(0.5 * (r0["losses"][-1] + r1["losses"][-1]) - float(step.loss)) < 1e-5


def test_bench_counts_gpus_without_touching_the_runtime(monkeypatch):
    """the spawning parent of `bench.py --gpus N` must not bring up HIP: devices are counted from the environment / KFD"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1,2")
    assert bench.visible_gpu_count() == 3
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "GPU-abc")
    assert bench.visible_gpu_count() == 1
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES")
    monkeypatch.delenv("CUDA_VISIBLE_DEVICES", raising=False)
    assert bench.visible_gpu_count() in (None, 0) or bench.visible_gpu_count() >= 1
    src = open(os.path.join(ROOT, "bench.py")).read()
    body = src[src.index("def spawn_ranks"):src.index("def dry_run")]
    assert "torch.cuda" not in body.replace("torch.cuda or HIP", "")


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
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
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
    # the proof of what the process group was: gathered from every rank, not asserted by rank 0
    d = rec["dist"]
    assert d["backend"] == "gloo" and d["world_size_seen"] == 2 and d["distinct_processes"] == 2
    assert d["devices"] == ["dry-run device of rank 0", "dry-run device of rank 1"]
    assert d["step_ms_min"] < d["step_ms_max"]
    assert len(d["cpu_pin"]) == 2 and all(p["pinned"] is False for p in d["cpu_pin"])   # no KFD topology in this container


def _fake_sysfs(root, gpus, cpulists):
    """a KFD topology + PCI tree like an 8-GPU MI355X host's: node 0.. = CPUs (no SIMDs), then one node per GPU"""
    for i in range(2):
        os.makedirs(os.path.join(root, f"class/kfd/kfd/topology/nodes/{i}"))
        with open(os.path.join(root, f"class/kfd/kfd/topology/nodes/{i}/properties"), "w") as fh:
            fh.write("cpu_cores_count 64\nsimd_count 0\nlocation_id 0\ndomain 0\n")
    for g, (bus, cpus) in enumerate(zip(gpus, cpulists)):
        node = os.path.join(root, f"class/kfd/kfd/topology/nodes/{2 + g}")
        os.makedirs(node)
        with open(os.path.join(node, "properties"), "w") as fh:
            fh.write(f"cpu_cores_count 0\nsimd_count 1024\nlocation_id {bus << 8}\ndomain 0\n")
        dev = os.path.join(root, f"bus/pci/devices/0000:{bus:02x}:00.0")
        os.makedirs(dev)
        with open(os.path.join(dev, "local_cpulist"), "w") as fh:
            fh.write(cpus + "\n")


def test_rank_cpu_pinning_reads_the_gpu_numa_node_from_sysfs(tmp_path, monkeypatch):
    """bench.gpu_local_cpus / pin_rank_to_gpu_cpus: GPU index -> KFD node -> PCI address -> local_cpulist, the ranks
    whose GPUs share a NUMA node split its CPUs in rank order; *_VISIBLE_DEVICES re-maps; nothing here touches a GPU."""
    sys.path.insert(0, ROOT)
    import bench
    n_cpu = len(os.sched_getaffinity(0))
    have = sorted(os.sched_getaffinity(0))
    lo, hi = have[:n_cpu // 2], have[n_cpu // 2:]
    as_list = lambda c: ",".join(str(v) for v in c)
    root = str(tmp_path / "sys")
    _fake_sysfs(root, [0x05, 0x15, 0x85, 0x95], [as_list(lo), as_list(lo), as_list(hi), as_list(hi)])
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    assert bench.gpu_local_cpus(0, root) == (lo, "0000:05:00.0")
    assert bench.gpu_local_cpus(3, root) == (hi, "0000:95:00.0")
    assert bench.gpu_local_cpus(4, root)[0] is None
    assert bench._cpulist("0-3,8,10-11") == [0, 1, 2, 3, 8, 10, 11]
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "2,0")
    assert bench.gpu_local_cpus(0, root) == (hi, "0000:85:00.0") and bench.gpu_local_cpus(1, root)[1] == "0000:05:00.0"
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    before = os.sched_getaffinity(0)
    try:
        if len(lo) >= 4:
            rec = bench.pin_rank_to_gpu_cpus(1, 4, root)          # rank 1 shares the first node with rank 0: its second half
            assert rec["pinned"] and rec["ranks_sharing_node"] == 2 and rec["pci"] == "0000:15:00.0"
            assert os.sched_getaffinity(0) == set(lo[len(lo) // 2:2 * (len(lo) // 2)])
            os.sched_setaffinity(0, before)
        assert bench.pin_rank_to_gpu_cpus(0, 1, str(tmp_path / "nothing"))["pinned"] is False   # unreadable: left alone
        assert os.sched_getaffinity(0) == before
    finally:
        os.sched_setaffinity(0, before)


@pytest.mark.timeout(300)
def test_bench_ranks_pin_themselves_in_the_dry_run(tmp_path):
    """the spawned ranks (gloo dry run) pin themselves from a sysfs tree and the JSON line carries what each one did"""
    import json
    have = sorted(os.sched_getaffinity(0))
    if len(have) < 4:
        pytest.skip("needs 4 CPUs")
    as_list = lambda c: ",".join(str(v) for v in c)
    root = str(tmp_path / "sys")
    _fake_sysfs(root, [0x05, 0x85], [as_list(have[:len(have) // 2]), as_list(have[len(have) // 2:])])
    res = _run_bench({"HWGAT_BENCH_SYSFS": root}, "--gpus", "2", "--steps", "2", "--warmup", "1")
    assert res.returncode == 0, res.stderr[-2000:]
    d = json.loads(res.stdout.strip())["dist"]
    pins = d["cpu_pin"]
    assert [p["pinned"] for p in pins] == [True, True] and [p["pci"] for p in pins] == ["0000:05:00.0", "0000:85:00.0"]
    assert pins[0]["last_cpu"] < pins[1]["first_cpu"] and pins[0]["cpus"] == len(have) // 2


@pytest.mark.timeout(300)
def test_bench_parent_fails_when_a_rank_fails_or_the_launcher_disagrees():
    res = _run_bench({"HWGAT_BENCH_DRYRUN_FAIL_RANK": "1"}, "--gpus", "2")
    assert res.returncode != 0 and res.stdout.strip() == "" and "rank 1 exited with code 3" in res.stderr
    res = _run_bench({"WORLD_SIZE": "2", "RANK": "0"}, "--gpus", "4")
    assert res.returncode != 0 and "WORLD_SIZE=2" in res.stderr
