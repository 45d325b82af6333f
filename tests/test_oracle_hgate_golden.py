"""CPU: pin the HGATE oracle (oracle/hgat_oracle.py) to golden vectors produced by the reference's
hwgat/models/HGATE.py (tests/golden/make_fixtures_hgate.py)."""
import numpy as np
import pytest
import torch

from oracle import hgat_oracle as OH
from helpers import load_fixture, hgate_oracle_from_fixture, sub, rel_err, grad_digest_check

TOL = 2e-5


def test_skeleton_adjacency_and_shift_masks_match_reference():
    fx = load_fixture("hgate_a.npz")
    a = OH.block_adjacency()
    assert a.shape == (58, 58) and np.array_equal(a.numpy(), fx["adj"])
    j = OH.skeleton_adjacency()
    assert len(OH.skeleton_edges()) == 34 and int(j.sum()) == 29 + 2 * 34 and torch.equal(j, j.t())
    n = 0
    for k, v in fx.items():
        if k.startswith("mask."):
            F = 128 // 2 ** int(k.split(".")[2])
            assert np.array_equal(OH.block_shift_mask(F, 29).numpy().astype(np.uint8), v), k
            n += 1
    assert n == 4


@pytest.mark.parametrize("name", ["hgate_a.npz", "hgate_b.npz"])
def test_eval_forward_taps(name):
    fx = load_fixture(name)
    model, params, cfg = hgate_oracle_from_fixture(fx)
    with torch.no_grad():
        logits = model.forward(torch.from_numpy(fx["x"]), tap=True)
    assert rel_err(logits, fx["eval.logits"]) < TOL
    assert rel_err(model.taps["feat"], fx["eval.feat"]) < TOL
    for b in range(8):
        assert rel_err(sub(model.taps[f"block{b}"]), fx[f"eval.block{b}"]) < TOL, b
    for i in range(2):
        assert rel_err(sub(model.taps[f"merge{i}"]), fx[f"eval.merge{i}"]) < TOL
    assert rel_err(model.taps["block1"][0, -2:], fx["eval.block1.full"]) < TOL


@pytest.mark.parametrize("name", ["hgate_a.npz", "hgate_b.npz"])
def test_backward(name):
    fx = load_fixture(name)
    model, params, cfg = hgate_oracle_from_fixture(fx)
    ps = {k: v.clone().requires_grad_(k not in ("B", "pos_encoder.pe")) for k, v in params.items()}
    model.p = ps
    loss = OH.smoothed_cross_entropy(model.forward(torch.from_numpy(fx["x"])), torch.from_numpy(fx["y"]))
    loss.backward()
    assert abs(loss.item() - float(fx["evalbwd.loss"])) < 1e-5
    grad_digest_check({k: v.grad for k, v in ps.items() if v.grad is not None}, fx, "evalbwd.", 2e-4)
