"""GPU: a short TRAINING trajectory (AdamW, lr 5e-4 as in the reference's configs.py:84 /
utils.py:76) on the HIP backend vs the CPU oracle -- same parameters, same batch, injected
train-mode thresholds, dropout off.  Checks loss per step and the parameters after the last step."""
import importlib

import pytest
import torch

from oracle import hwgat_oracle as O
from helpers import rel_err

pytestmark = pytest.mark.gpu
hw = importlib.import_module("sl-hwgat_amd")
DEV = torch.device("cuda:0")


def test_three_adamw_steps_match_the_oracle():
    T, nW, C, nc, B, steps = 16, 2, 2, 6, 4, 3
    cfg = dict(kp_dim=C, temporal_dim=T, num_classes=nc, embed_dim=128, num_kps=nW * 16)
    params = O.synth_params(31, **cfg)
    thr = [[0.4, 0.15, 0.3, 0.1, 0.55, 0.2, 0.08, 0.7], [0.25] * 8, [0.6, 0.05, 0.35, 0.12, 0.45, 0.3, 0.2, 0.5]]
    g = torch.Generator().manual_seed(5)
    x = torch.rand(B, T, nW * 16, C, generator=g)
    y = torch.randint(0, nc, (B,), generator=g)

    # ---- oracle trajectory (fp32 CPU, torch AdamW)
    ref_p = {k: torch.nn.Parameter(v.clone(), requires_grad=k not in ("B", "pos_encoder.pe")) for k, v in params.items()}
    oracle = O.OracleHWGAT(ref_p, num_kps=nW * 16, temporal_dim=T)
    opt_r = torch.optim.AdamW([p for p in ref_p.values() if p.requires_grad], lr=5e-4)
    ref_losses = []
    for s in range(steps):
        opt_r.zero_grad()
        loss = O.smoothed_cross_entropy(oracle.forward(x, thresholds=thr[s]), y)
        loss.backward()
        opt_r.step()
        ref_losses.append(loss.item())

    # ---- HIP backend trajectory
    hp = hw.HWGATEParams({"src_len": T, "num_class": nc}, C, DEV, num_kps=nW * 16)
    hp.drop_rate = 0.0
    model = hw.Model(*hp.get_model_params())
    model.load_state_dict(params, strict=False)
    model.train()
    tr = importlib.import_module("sl-hwgat_amd.train")
    opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=5e-4, fused=True)
    step = tr.TrainStep(model, opt)
    losses = []
    for s in range(steps):
        model.threshold_override = thr[s]
        losses.append(float(step(x.to(DEV), y.to(DEV))))

    for a, b in zip(losses, ref_losses):
        assert abs(a - b) < 2e-4, (losses, ref_losses)
    assert ref_losses[0] != ref_losses[-1]                     # the trajectory actually moved
    sd = model.state_dict()
    errs = {}
    for k, v in ref_p.items():
        if not v.requires_grad:
            continue
        a, b = sd[k].cpu(), v.detach()
        if k.endswith("attn.qkv.bias"):
            # the key bias has an exactly-zero true gradient (softmax is invariant to it), so both sides feed
            # rounding noise to Adam, which turns it into +-lr updates: compare only the q and v thirds
            d = a.numel() // 3
            a, b = torch.cat([a[:d], a[2 * d:]]), torch.cat([b[:d], b[2 * d:]])
        errs[k] = rel_err(a, b)
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:3]
    print("losses", losses, "ref", ref_losses, "worst param rel errs", worst)
    assert worst[0][1] < 2e-4


def test_twenty_adamw_cosine_steps_match_the_oracle():
    """utils.py:76,84-88: AdamW(lr 5e-4) + CosineAnnealingLR(T_max=20), one scheduler step per optimizer step, 20
    steps on the HIP backend vs the oracle (which tests/test_checkpoint_cpu.py ties to the REFERENCE model's own
    20-step trajectory): learning rate per step identical, loss per step and final parameters within tolerance."""
    T, nW, C, nc, B, steps = 16, 2, 2, 6, 4, 20
    cfg = dict(kp_dim=C, temporal_dim=T, num_classes=nc, embed_dim=128, num_kps=nW * 16)
    params = O.synth_params(31, **cfg)
    thr = [[0.25 + 0.03 * ((s + k) % 7) for k in range(8)] for s in range(steps)]
    g = torch.Generator().manual_seed(5)
    x = torch.rand(B, T, nW * 16, C, generator=g)
    y = torch.randint(0, nc, (B,), generator=g)
    ck = hw.checkpoint

    ref_p = {k: torch.nn.Parameter(v.clone(), requires_grad=k not in ("B", "pos_encoder.pe")) for k, v in params.items()}
    oracle = O.OracleHWGAT(ref_p, num_kps=nW * 16, temporal_dim=T)
    opt_r = ck.get_optimizer(torch.nn.ParameterList(ref_p.values()))
    sch_r = ck.get_scheduler(opt_r)
    ref_losses, ref_lr = [], []
    for s in range(steps):
        opt_r.zero_grad()
        loss = O.smoothed_cross_entropy(oracle.forward(x, thresholds=thr[s]), y)
        loss.backward()
        opt_r.step()
        ref_lr.append(opt_r.param_groups[0]["lr"])
        sch_r.step()
        ref_losses.append(loss.item())

    hp = hw.HWGATEParams({"src_len": T, "num_class": nc}, C, DEV, num_kps=nW * 16)
    hp.drop_rate = 0.0
    model = hw.Model(*hp.get_model_params())
    model.load_state_dict(params, strict=False)
    model.train()
    tr = importlib.import_module("sl-hwgat_amd.train")
    opt = ck.get_optimizer(model, fused=True)
    sch = ck.get_scheduler(opt)
    step = tr.TrainStep(model, opt)
    losses, lrs = [], []
    for s in range(steps):
        model.threshold_override = thr[s]
        losses.append(float(step(x.to(DEV), y.to(DEV))))
        lrs.append(opt.param_groups[0]["lr"])
        sch.step()
    assert max(abs(a - b) for a, b in zip(lrs, ref_lr)) < 1e-12
    worst_loss = max(abs(a - b) for a, b in zip(losses, ref_losses))
    sd = model.state_dict()
    worst = max(rel_err(sd[k].cpu(), v.detach()) for k, v in ref_p.items()
                if v.requires_grad and not k.endswith("attn.qkv.bias"))
    print("20 steps: worst loss diff", worst_loss, "worst param rel err", worst, "loss", losses[0], "->", losses[-1])
    assert worst_loss < 1e-3 and losses[-1] < losses[0]
    assert worst < 2e-3
