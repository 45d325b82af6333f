"""GPU, world size 1 over RCCL (backend "nccl"): the HIP model's gradients pass through dist.GradReducer
(bucket views, post-accumulate hooks, async all-reduce(AVG), finish()) exactly as on the N>1 path and must equal
the gradients of a plain backward.  The N=2 exchange itself is covered on CPU (tests/test_dist_cpu.py, gloo)."""
import importlib
import os
import socket

import pytest
import torch
import torch.distributed as dist

from oracle import hwgat_oracle as O
from helpers import rel_err

pytestmark = pytest.mark.gpu
hw = importlib.import_module("sl-hwgat_amd")
dmod = importlib.import_module("sl-hwgat_amd.dist")
tmod = importlib.import_module("sl-hwgat_amd.train")
DEV = torch.device("cuda:0")


@pytest.fixture(scope="module")
def nccl_group():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=DEV)
    yield
    dist.destroy_process_group()


def _model(T=16, nW=2, nc=6, seed=21):
    cfg = dict(kp_dim=2, temporal_dim=T, num_classes=nc, embed_dim=128, num_kps=nW * 16)
    hp = hw.HWGATEParams({"src_len": T, "num_class": nc}, 2, DEV, num_kps=nW * 16)
    hp.drop_rate = 0.0
    model = hw.Model(*hp.get_model_params())
    model.load_state_dict(O.synth_params(seed, **cfg), strict=False)
    model.train()
    model.threshold_override = [0.3, 0.1, 0.5, 0.2, 0.07, 0.4, 0.25, 0.6]
    return model.to(DEV)


def test_hip_model_gradients_through_the_reducer_on_rccl(nccl_group):
    g = torch.Generator().manual_seed(5)
    x = torch.rand(4, 16, 32, 2, generator=g).to(DEV)
    y = torch.randint(0, 6, (4,), generator=g).to(DEV)
    plain = _model()
    tmod.TrainStep(plain)(x, y)
    ref = {k: p.grad.clone() for k, p in plain.named_parameters() if p.grad is not None}

    model = _model()
    dmod.broadcast_parameters(model)
    assert model.rank_salt == 0
    red = dmod.GradReducer(model.parameters(), bucket_bytes=4 << 20, always_reduce=True)
    assert len(red.buckets) >= 3 and red._use_avg
    step = tmod.TrainStep(model, None, red, micro_batch=2)          # two accumulation passes into the buckets
    for _ in range(2):                                              # second step: zero_grad + bucket reuse
        loss = step(x, y)
    assert not red._handles                                         # finish() waited for every bucket
    got = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    assert set(got) == set(ref)
    for k in ref:
        assert rel_err(got[k].cpu(), ref[k].cpu()) < 2e-5, k        # fp32 split-M atomics: rounding-level
        lo, hi = red.buckets[0]["flat"].data_ptr(), None
        assert any(b["flat"].data_ptr() <= got[k].data_ptr() < b["flat"].data_ptr() + b["flat"].numel() * 4
                   for b in red.buckets), k                         # .grad still lives inside a bucket
    assert torch.isfinite(loss)
