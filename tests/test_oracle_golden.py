"""CPU: pin the oracle (oracle/hwgat_oracle.py) to golden vectors produced by
the reference model (tests/golden/make_fixtures.py)."""
import numpy as np
import torch

from oracle import hwgat_oracle as O
from helpers import load_fixture, oracle_from_fixture, sub, rel_err, grad_digest_check

TOL = 2e-5      # fp32 restatement vs fp32 reference, same ATen kernels, different op order


def _grads(model, params, x, y, thr):
    ps = {k: v.clone().requires_grad_(k not in ("B", "pos_encoder.pe")) for k, v in params.items()}
    model.p = ps
    logits = model.forward(x, thresholds=thr)
    loss = O.smoothed_cross_entropy(logits, y)
    loss.backward()
    return logits.detach(), loss.item(), {k: v.grad for k, v in ps.items() if v.grad is not None}


def test_adjacency_and_shift_mask_match_reference():
    fx = load_fixture("cfg1.npz")
    assert np.array_equal(O.window_adjacency(2).numpy(), fx["adj"])
    a = O.window_adjacency(1)[0]
    assert int(a.sum()) == 164 and torch.equal(a, a.t())
    for k, v in fx.items():
        if k.startswith("mask."):
            stage = int(k.split(".")[2])
            F = 32 // 2 ** stage
            assert np.array_equal(O.shift_mask(F, 2).numpy(), v), k


def test_eval_forward_taps_cfg1():
    fx = load_fixture("cfg1.npz")
    model, params, cfg = oracle_from_fixture(fx)
    x = torch.from_numpy(fx["x"])
    with torch.no_grad():
        logits = model.forward(x, tap=True)
    assert rel_err(logits, fx["eval.logits"]) < TOL
    assert rel_err(model.taps["feat"], fx["eval.feat"]) < TOL
    assert rel_err(sub(model.taps["pe"]), fx["eval.pe"]) < TOL
    for b in range(8):
        assert rel_err(sub(model.taps[f"block{b}"]), fx[f"eval.block{b}"]) < TOL, b
    for i in range(2):
        assert rel_err(sub(model.taps[f"merge{i}"]), fx[f"eval.merge{i}"]) < TOL
    assert rel_err(model.taps["block0"][0, :4], fx["eval.block0.full"]) < TOL
    assert rel_err(model.taps["block1"][0, -4:], fx["eval.block1.full"]) < TOL


def test_eval_backward_cfg1():
    fx = load_fixture("cfg1.npz")
    model, params, cfg = oracle_from_fixture(fx)
    x, y = torch.from_numpy(fx["x"]), torch.from_numpy(fx["y"])
    _, loss, grads = _grads(model, params, x, y, None)
    assert abs(loss - float(fx["evalbwd.loss"])) < 1e-5
    grad_digest_check(grads, fx, "evalbwd.", 2e-4)


def test_train_thresholds_cfg1():
    fx = load_fixture("cfg1.npz")
    x, y = torch.from_numpy(fx["x"]), torch.from_numpy(fx["y"])
    for tag in ("mid", "lo", "hi"):
        model, params, cfg = oracle_from_fixture(fx)
        thr = [float(v) for v in fx[f"train.{tag}.thr"]]
        logits, loss, grads = _grads(model, params, x, y, thr)
        assert rel_err(logits, fx[f"train.{tag}.logits"]) < 5e-5, tag
        assert abs(loss - float(fx[f"train.{tag}.loss"])) < 1e-5
        grad_digest_check(grads, fx, f"train.{tag}.", 5e-4)
    # thr ~ 1 is a no-op (equals eval); thr ~ 0 masks every row -> different logits
    assert rel_err(fx["train.hi.logits"], fx["eval.logits"]) < 1e-6
    assert rel_err(fx["train.lo.logits"], fx["eval.logits"]) > 1e-3


def test_nw5_and_hd128_variants():
    fx = load_fixture("nw5.npz")
    model, params, cfg = oracle_from_fixture(fx)
    x, y = torch.from_numpy(fx["x"]), torch.from_numpy(fx["y"])
    with torch.no_grad():
        assert rel_err(model.forward(x), fx["eval.logits"]) < TOL
    thr = [float(v) for v in fx["train.thr"]]
    logits, loss, grads = _grads(model, params, x, y, thr)
    assert rel_err(logits, fx["train.logits"]) < 5e-5
    grad_digest_check(grads, fx, "train.", 5e-4)

    fx = load_fixture("hd128.npz")
    model, params, cfg = oracle_from_fixture(fx)
    x, y = torch.from_numpy(fx["x"]), torch.from_numpy(fx["y"])
    logits, loss, grads = _grads(model, params, x, y, None)
    assert rel_err(logits, fx["eval.logits"]) < TOL
    assert abs(loss - float(fx["eval.loss"])) < 1e-5
    grad_digest_check(grads, fx, "eval.", 2e-4)


def test_benchmarked_shapes_one_clip_each():
    """BASELINE configs[1] (T128, nW5, d0 128) and configs[4] (T256, nW7, C3, d0 256) at B=1: eval fwd+bwd,
    block taps, and a train-threshold run, all from the reference"""
    for name in ("cfg2_clip.npz", "cfg5_clip.npz"):
        fx = load_fixture(name)
        model, params, cfg = oracle_from_fixture(fx)
        x, y = torch.from_numpy(fx["x"]), torch.from_numpy(fx["y"])
        with torch.no_grad():
            model.forward(x, tap=True)
        for b in range(8):
            assert rel_err(model.taps[f"block{b}"][:, ::9, ::7, ::11], fx[f"eval.block{b}"]) < TOL, (name, b)
        logits, loss, grads = _grads(model, params, x, y, None)
        assert rel_err(logits, fx["eval.logits"]) < TOL, name
        assert abs(loss - float(fx["eval.loss"])) < 1e-5
        grad_digest_check(grads, fx, "eval.", 2e-4)
        if name == "cfg2_clip.npz":
            logits, loss, grads = _grads(model, params, x, y, [float(v) for v in fx["train.thr"]])
            assert rel_err(logits, fx["train.logits"]) < 5e-5
            grad_digest_check(grads, fx, "train.", 5e-4)


def test_fp64_oracle_agrees_with_fp32():
    """the fp64 instance is what GPU kernels are compared against"""
    fx = load_fixture("nw5.npz")
    m32, _, _ = oracle_from_fixture(fx)
    m64, _, _ = oracle_from_fixture(fx, torch.float64)
    x = torch.from_numpy(fx["x"])
    with torch.no_grad():
        a, b = m32.forward(x), m64.forward(x.double())
    assert rel_err(a, b) < 2e-5
