"""CPU: pin the WGATE oracle (oracle/wgat_oracle.py) to golden vectors produced by the reference's
hwgat/models/WGATE.py (tests/golden/make_fixtures_wgate.py)."""
import numpy as np
import pytest
import torch

from oracle import wgat_oracle as OW
from helpers import load_fixture, wgate_oracle_from_fixture, rel_err, grad_digest_check

TOL = 2e-5


def _sub(t):
    return t[:, ::3, ::5, ::7]


def test_band_adjacency_and_mask_match_reference():
    fx = load_fixture("wgate_a.npz")
    a = OW.band_adjacency(32, 4)
    assert a.shape == (4, 512, 512) and np.array_equal(a[0].numpy().astype(np.uint8), fx["adj_w0"])
    assert np.array_equal(OW.additive_mask(a)[0, :48, :48].numpy(), fx["adj_mask_w0_head"])
    assert torch.equal(a[0], a[3])


@pytest.mark.parametrize("name", ["wgate_a.npz", "wgate_b.npz"])
def test_eval_forward_taps(name):
    fx = load_fixture(name)
    model, params, cfg = wgate_oracle_from_fixture(fx)
    with torch.no_grad():
        logits = model.forward(torch.from_numpy(fx["x"]), tap=True)
    assert rel_err(logits, fx["eval.logits"]) < TOL
    assert rel_err(model.taps["feat"], fx["eval.feat"]) < TOL
    for b in range(cfg["depths"]):
        assert rel_err(_sub(model.taps[f"block{b}"]), fx[f"eval.block{b}"]) < TOL, b
    assert rel_err(model.taps["block0"][0, :3], fx["eval.block0.full"]) < TOL
    assert rel_err(model.taps["block0"][0, -2:], fx["eval.block0.tail"]) < TOL


@pytest.mark.parametrize("name", ["wgate_a.npz", "wgate_b.npz"])
def test_backward(name):
    fx = load_fixture(name)
    model, params, cfg = wgate_oracle_from_fixture(fx)
    ps = {k: v.clone().requires_grad_(k not in ("B", "pos_encoder.pe")) for k, v in params.items()}
    model.p = ps
    loss = OW.smoothed_cross_entropy(model.forward(torch.from_numpy(fx["x"])), torch.from_numpy(fx["y"]))
    loss.backward()
    assert abs(loss.item() - float(fx["evalbwd.loss"])) < 1e-5
    grad_digest_check({k: v.grad for k, v in ps.items() if v.grad is not None}, fx, "evalbwd.", 2e-4)
