"""Bit-reproducible training (`model.deterministic_train = True`, VERDICT round 3 "missing" item 4): the reference on one device
with fixed seeds repeats itself bit for bit (plain ATen ops, hwgat/utils.py:93-116); by default this backend combines weight / bias
/ LayerNorm-parameter gradients, the pooled sum and some row statistics with fp32 atomics, whose order varies from run to run.
The deterministic mode routes every one of them through fixed-order reductions (hwgat_linear_tn_*_det, hwgat_ln_bwd_det,
hwgat_lnpool_fwd_det, order-fixed row statistics): two runs from the same state are `torch.equal` in every gradient and every
weight after several optimizer steps -- and the results are the default path's up to summation order."""
import importlib

import pytest
import torch

pytestmark = pytest.mark.gpu
hw = importlib.import_module("sl-hwgat_amd")
HF = hw.functional
train = importlib.import_module("sl-hwgat_amd.train")
DEV = torch.device("cuda:0")


def _build(kind, dtype, d0=128):
    torch.manual_seed(21)
    if kind == "hwgate":
        hp = hw.HWGATEParams({"src_len": 32, "num_class": 9}, 2, DEV, num_kps=32, embed_dim=d0)
        model = hw.Model(*hp.get_model_params()).to(DEV)
    elif kind == "hgate":
        hp = hw.HGATEParams({"src_len": 32, "num_class": 9}, 2, DEV)
        model = hw.HGATEModel(*hp.get_model_params()).to(DEV)
    else:
        hp = hw.WGATEParams({"src_len": 32, "num_class": 9}, 2, DEV, num_kps=32)
        model = hw.WGATEModel(*hp.get_model_params()).to(DEV)
    model.set_activation_dtype(dtype)
    return model.train()


def _run(model, x, y, steps, deterministic):
    """`steps` AdamW steps from the model's CURRENT weights with fixed seeds; returns (losses, first-step grads, final weights)"""
    model.deterministic_train = deterministic
    model._drop_calls = 5
    torch.manual_seed(99)                                        # the train-mode thresholds of HWGATE.py:96 come from this generator
    opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=5e-4, fused=True)
    step = train.TrainStep(model, opt, None)
    losses, grads = [], None
    for k in range(steps):
        losses.append(step(x, y).clone())
        if k == 0:
            grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    return losses, grads, {n: p.detach().clone() for n, p in model.named_parameters()}


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("kind,d0", [("hwgate", 128), ("hwgate", 256), ("hgate", 128), ("wgate", 128)])
def test_deterministic_train_mode_repeats_bit_for_bit(kind, d0, dtype):
    model = _build(kind, dtype, d0)
    assert model.drop_rate == 0.1                                 # dropout is ON (the reference default)
    g = torch.Generator(device=DEV).manual_seed(4)
    x = torch.rand(8, 32, model.num_kps, 2, device=DEV, generator=g)      # M = 8 * 32 * K: every stage a multiple of 32 rows
    y = torch.randint(0, 9, (8,), device=DEV, generator=g)
    w0 = {k: v.clone() for k, v in model.state_dict().items()}
    runs = []
    for _ in range(2):
        model.load_state_dict(w0)
        junk = torch.randn(1 << 22, device=DEV).sum()            # unrelated work in between: different block timing
        runs.append(_run(model, x, y, 3, deterministic=True))
        assert junk.isfinite()
    (l1, g1, w1), (l2, g2, w2) = runs
    assert all(torch.equal(a, b) for a, b in zip(l1, l2))
    assert set(g1) == set(g2) and len(g1) == sum(1 for p in model.parameters() if p.requires_grad)
    for n in g1:
        assert torch.equal(g1[n], g2[n]), n                       # every gradient, every bit
    for n in w1:
        assert torch.equal(w1[n], w2[n]), n                       # ... and three optimizer steps later every weight
    # the deterministic mode computes what the default mode computes, up to summation order
    model.load_state_dict(w0)
    l0, g0, _ = _run(model, x, y, 1, deterministic=False)
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert abs(float(l0[0]) - float(l1[0])) <= tol * max(1.0, abs(float(l1[0])))
    worst = max(float((g0[n].double() - g1[n].double()).norm() / g1[n].double().norm().clamp_min(1e-30)) for n in g1)
    assert worst < (1e-4 if dtype == torch.float32 else 5e-2), worst


def test_deterministic_gradient_entry_points_against_the_atomic_forms():
    """hwgat_linear_tn_*_det / hwgat_ln_bwd_det alone: equal to the default kernels up to summation order, identical across
    repeats, for every prologue the fused block uses; ragged token counts are refused (one launch, one reduction)"""
    g = torch.Generator(device=DEV).manual_seed(1)
    for dt in (torch.float32, torch.bfloat16):
        for M, N, K in ((4096, 256, 128), (8192, 512, 512), (2048, 384, 128)):
            A = torch.randn(M, N, device=DEV, generator=g).to(dt)
            Bm = torch.randn(M, K, device=DEV, generator=g).to(dt)
            gamma, beta = 1 + 0.2 * torch.randn(K, device=DEV, generator=g), 0.2 * torch.randn(K, device=DEV, generator=g)
            mean, rstd = HF.ln_stats(Bm, gamma, beta)
            for kw in ({}, {"pro_seed": 7, "pro_p": 0.1}, {"ln": (mean, rstd, gamma, beta)}):
                ref_w, ref_b = torch.zeros(N, K, device=DEV), torch.zeros(N, device=DEV)
                HF.linear_tn(A, Bm, ref_w, ref_b, **kw)
                outs = []
                for _ in range(2):
                    dw, db = torch.zeros(N, K, device=DEV), torch.zeros(N, device=DEV)
                    HF.linear_tn(A, Bm, dw, db, deterministic=True, **kw)
                    outs.append((dw, db))
                assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), (dt, M, N, K, list(kw))
                assert float((outs[0][0] - ref_w).norm() / ref_w.norm()) < 1e-5, (dt, M, N, K, list(kw))
                assert float((outs[0][1] - ref_b).norm() / ref_b.norm()) < 1e-5
        with pytest.raises(NotImplementedError):
            HF.linear_tn(A[:100], Bm[:100], torch.zeros(N, K, device=DEV), deterministic=True)
        # LayerNorm backward: all four output combinations
        n, d = 5000, 256
        x = torch.randn(n, d, device=DEV, generator=g).to(dt)
        dy = torch.randn(n, d, device=DEV, generator=g).to(dt)
        res = torch.randn(n, d, device=DEV, generator=g).to(dt)
        gamma, beta = 1 + 0.2 * torch.randn(d, device=DEV, generator=g), 0.2 * torch.randn(d, device=DEV, generator=g)
        mean, rstd = HF.ln_stats(x, gamma, beta)
        for kw in ({}, {"mask": (11, 0.1)}, {"beta": beta}, {"mask": (11, 0.1), "beta": beta}):
            dg0 = torch.zeros(2, d, device=DEV)
            ref = HF.ln_backward(dy, x, mean, rstd, gamma, res, dg0[0], dg0[1], **kw)
            ref = ref if isinstance(ref, tuple) else (ref,)
            rep = []
            for _ in range(2):
                dg = torch.zeros(2, d, device=DEV)
                out = HF.ln_backward(dy, x, mean, rstd, gamma, res, dg[0], dg[1], deterministic=True, **kw)
                rep.append((out if isinstance(out, tuple) else (out,), dg))
            assert torch.equal(rep[0][1], rep[1][1])
            assert len(rep[0][0]) == len(ref) and all(torch.equal(a, b) for a, b in zip(rep[0][0], ref))   # dx / masked copy / LN(x): same code
            assert float((rep[0][1] - dg0).norm() / dg0.norm()) < 1e-5
