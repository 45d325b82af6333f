"""CPU (development container): SURVEY 8f-4 second half.

 * a checkpoint FILE written by this backend (`checkpoint.save_checkpoint`, the reference's dict layout,
   hwgat/utils.py:164-176) loads into the reference `Model` by the reference's own loading rules
   (utils.py:185-214: "model." prefix strip, shape filter), and a file in that layout written from the reference
   model -- with a "model." prefix and a head of another size -- loads into this backend's model;
 * optimizer + scheduler + epoch + metric lists survive `load_checkpoint` (utils.py:216-237);
 * a 20-step AdamW(lr 5e-4) + CosineAnnealingLR(T_max=20) trajectory (utils.py:76,84-88) of the REFERENCE model
   equals the oracle's: loss per step, learning rate per step, parameters after the last step.
The reference's utils.py itself cannot be imported (decord / matplotlib); its rules are restated in
sl-hwgat_amd/checkpoint.py, and the reference MODEL class is the real one."""
import importlib
import math
import os
import sys
import types

import pytest
import torch

from oracle import hwgat_oracle as O
from helpers import rel_err

hw = importlib.import_module("sl-hwgat_amd")
ck = hw.checkpoint
REF = "/root/reference/hwgat"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="needs /root/reference (development container)")


def _reference():
    for name in ("timm", "timm.models", "timm.models.layers"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["timm.models.layers"].trunc_normal_ = torch.nn.init.trunc_normal_    # init-only import, HWGATE.py:4
    sys.path.insert(0, REF)
    try:
        mod = importlib.import_module("models.HWGATE")
        par = importlib.import_module("models.model_params")
        loss = importlib.import_module("losses.SmoothCrossEntropy")
    finally:
        sys.path.remove(REF)
    return mod.Model, par.HWGATEParams, loss.SmoothedCrossEntropyLoss


def _pair(nc_ref=11, nc_mine=11, T=16):
    Model, Params, _ = _reference()
    rp = Params({"src_len": T, "num_class": nc_ref}, 2, torch.device("cpu"))
    hp = hw.HWGATEParams({"src_len": T, "num_class": nc_mine}, 2, torch.device("cpu"))
    return Model(*rp.get_model_params()), hw.Model(*hp.get_model_params())


def test_checkpoint_file_written_here_loads_into_the_reference_class(tmp_path):
    ref, mine = _pair()
    for p in mine.parameters():
        if p.requires_grad:
            p.data.normal_(0, 0.05)
    opt = ck.get_optimizer(mine)
    sch = ck.get_scheduler(opt)
    assert isinstance(opt, torch.optim.AdamW) and opt.defaults["lr"] == 5e-4
    assert isinstance(sch, torch.optim.lr_scheduler.CosineAnnealingLR) and sch.T_max == 20
    path = str(tmp_path / "best_loss.pt")
    ck.save_checkpoint(path, mine, opt, sch, [0.1], [2.0], [0.2], [1.9], 6, 4.5e-4)
    raw = torch.load(path, map_location="cpu", weights_only=True)
    assert set(raw) == {"model_state_dict", "optimizer_state_dict", "train_loss_list", "val_loss_list",
                        "train_acc_list", "val_acc_list", "epoch", "learning_rate", "scheduler"}      # utils.py:166-175
    assert list(raw["model_state_dict"]) == list(ref.state_dict())
    # the reference's loading rules (utils.py:186-212), applied to the REFERENCE class
    rep = {}
    ck.load_weights_from_pretrained(ref, path, "cpu", rep)
    assert rep == {"unknown": [], "mismatched": [], "missing": []}
    sd_r, sd_m = ref.state_dict(), mine.state_dict()
    assert all(torch.equal(sd_r[k], sd_m[k]) for k in sd_m)
    # the reference's optimizer / scheduler accept the saved states (utils.py:228-229)
    opt_r = torch.optim.AdamW(ref.parameters(), lr=5e-4)
    sch_r = torch.optim.lr_scheduler.CosineAnnealingLR(opt_r, T_max=20, last_epoch=-1)
    opt_r.load_state_dict(raw["optimizer_state_dict"])                          # same param-group size (B included)
    sch_r.load_state_dict(raw["scheduler"])
    assert sch_r.last_epoch == sch.last_epoch
    assert len(opt_r.param_groups[0]["params"]) == len(opt.param_groups[0]["params"]) == len(list(ref.parameters()))


def test_resume_from_a_file_whose_optimizer_state_the_reference_wrote(tmp_path):
    """utils.py:76 builds AdamW(model.parameters()) -- frozen `B` included -- and utils.py:228 loads that state back:
    a file written from the REFERENCE model + optimizer after two steps must resume here, moments per parameter name."""
    ref, mine = _pair()
    opt_r = torch.optim.AdamW(ref.parameters(), lr=5e-4)
    sch_r = torch.optim.lr_scheduler.CosineAnnealingLR(opt_r, T_max=20, last_epoch=-1)
    g = torch.Generator().manual_seed(5)
    for _ in range(2):
        for p in ref.parameters():
            p.grad = torch.randn(p.shape, generator=g) * 1e-3 if p.requires_grad else None
        opt_r.step()
        sch_r.step()
    path = str(tmp_path / "ref_written.pt")
    torch.save({"model_state_dict": ref.state_dict(), "optimizer_state_dict": opt_r.state_dict(), "train_loss_list": [1.0],
                "val_loss_list": [1.1], "train_acc_list": [0.1], "val_acc_list": [0.2], "epoch": 1,
                "learning_rate": sch_r.get_last_lr()[0], "scheduler": sch_r.state_dict()}, path)
    opt = ck.get_optimizer(mine)
    sch = ck.get_scheduler(opt)
    mine, opt, sch, lists, start = ck.load_checkpoint(path, mine, opt, sch)
    assert start == 2 and sch.last_epoch == 2
    names_r = [n for n, _ in ref.named_parameters()]
    names_m = [n for n, _ in mine.named_parameters()]
    assert names_r == names_m                                                   # same order -> same state indices
    st_r, st_m = opt_r.state_dict()["state"], opt.state_dict()["state"]
    assert set(st_r) == set(st_m) and len(st_m) == len(names_m) - 1             # every trainable parameter, not B
    for i, n in enumerate(names_m):
        if i in st_m:
            assert torch.equal(st_m[i]["exp_avg"], st_r[i]["exp_avg"]), n
            assert torch.equal(st_m[i]["exp_avg_sq"], st_r[i]["exp_avg_sq"]), n


def test_reference_layout_file_with_prefix_and_other_head_loads_here(tmp_path):
    ref, mine = _pair(nc_ref=7, nc_mine=11)
    init_head = mine.head.weight.detach().clone()
    sd = {"model." + k: v for k, v in ref.state_dict().items()}                 # a wrapped model's keys
    sd["model.extra.buffer"] = torch.zeros(3)                                   # a key this model does not have
    opt_r = torch.optim.AdamW(ref.parameters(), lr=5e-4)
    sch_r = torch.optim.lr_scheduler.CosineAnnealingLR(opt_r, T_max=20, last_epoch=-1)
    path = str(tmp_path / "pretrained.pt")
    torch.save({"model_state_dict": sd, "optimizer_state_dict": opt_r.state_dict(), "train_loss_list": [],
                "val_loss_list": [], "train_acc_list": [], "val_acc_list": [], "epoch": 0, "learning_rate": 5e-4,
                "scheduler": sch_r.state_dict()}, path)
    rep = {}
    ck.load_weights_from_pretrained(mine, path, "cpu", rep)
    assert rep["unknown"] == ["extra.buffer"] and rep["missing"] == []
    assert sorted(rep["mismatched"]) == ["head.bias", "head.weight"]            # 7 vs 11 classes: kept as initialised
    assert torch.equal(mine.head.weight, init_head)
    sd_m = mine.state_dict()
    for k, v in ref.state_dict().items():
        if not k.startswith("head."):
            assert torch.equal(sd_m[k], v), k


def test_resume_restores_optimizer_scheduler_epoch_and_lists(tmp_path):
    _, a = _pair()
    opt = ck.get_optimizer(a)
    sch = ck.get_scheduler(opt)
    for p in a.parameters():                                                    # three optimizer + scheduler steps
        if p.requires_grad:
            p.grad = torch.full_like(p, 1e-3)
    for _ in range(3):
        opt.step()
        sch.step()
    path = str(tmp_path / "last.pt")
    ck.save_checkpoint(path, a, opt, sch, [0.3, 0.4], [1.5, 1.2], [0.2, 0.3], [1.6, 1.4], 2, sch.get_last_lr()[0])
    _, b = _pair()
    opt_b = ck.get_optimizer(b)
    sch_b = ck.get_scheduler(opt_b)
    b, opt_b, sch_b, lists, start = ck.load_checkpoint(path, b, opt_b, sch_b)
    assert start == 3 and lists == [[1.5, 1.2], [1.6, 1.4], [0.3, 0.4], [0.2, 0.3]]      # utils.py:230-236 order
    assert sch_b.last_epoch == 3 and abs(sch_b.get_last_lr()[0] - sch.get_last_lr()[0]) < 1e-12
    assert abs(opt_b.param_groups[0]["lr"] - opt.param_groups[0]["lr"]) < 1e-12
    sa, sb = opt.state_dict()["state"], opt_b.state_dict()["state"]
    assert sa.keys() == sb.keys() and all(torch.equal(sa[i]["exp_avg"], sb[i]["exp_avg"]) for i in sa)
    assert all(torch.equal(x, y) for x, y in zip(a.state_dict().values(), b.state_dict().values()))


@pytest.mark.timeout(600)
def test_twenty_adamw_cosine_steps_reference_equals_oracle():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from make_fixtures import forced_thresholds
    Model, Params, Loss = _reference()
    T, nW, C, nc, B, steps = 16, 2, 2, 6, 4, 20
    cfg = dict(kp_dim=C, temporal_dim=T, num_classes=nc, embed_dim=128, num_kps=nW * 16)
    params = O.synth_params(31, **cfg)
    rp = Params({"src_len": T, "num_class": nc}, C, torch.device("cpu"))
    rp.num_kps, rp.drop_rate = nW * 16, 0.0
    rp.edges = [rp.edges[0]] * nW
    rp.adj_mat = torch.tensor(rp.get_adj_mat(), dtype=torch.float32)
    ref = Model(*rp.get_model_params())
    ref.load_state_dict(params, strict=False)
    ref.train()
    g = torch.Generator().manual_seed(5)
    x = torch.rand(B, T, nW * 16, C, generator=g)
    y = torch.randint(0, nc, (B,), generator=g)
    thr = [[0.25 + 0.03 * ((s + k) % 7) for k in range(8)] for s in range(steps)]
    crit = Loss()
    opt = torch.optim.AdamW(ref.parameters(), lr=5e-4)                                          # utils.py:76
    sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=20, last_epoch=-1)              # utils.py:86
    ref_loss, ref_lr = [], []
    for s in range(steps):
        opt.zero_grad()
        with forced_thresholds(thr[s]):
            loss = crit(ref(x), y)
        loss.backward()
        opt.step()
        ref_lr.append(opt.param_groups[0]["lr"])
        sch.step()
        ref_loss.append(loss.item())

    op = {k: torch.nn.Parameter(v.clone(), requires_grad=k not in ("B", "pos_encoder.pe")) for k, v in params.items()}
    oracle = O.OracleHWGAT(op, num_kps=nW * 16, temporal_dim=T)
    holder = torch.nn.ParameterList(op.values())
    opt_o = ck.get_optimizer(holder)
    sch_o = ck.get_scheduler(opt_o)
    o_loss, o_lr = [], []
    for s in range(steps):
        opt_o.zero_grad()
        loss = O.smoothed_cross_entropy(oracle.forward(x, thresholds=thr[s]), y)
        loss.backward()
        opt_o.step()
        o_lr.append(opt_o.param_groups[0]["lr"])
        sch_o.step()
        o_loss.append(loss.item())
    assert o_lr == ref_lr
    for s in range(steps):                                                     # closed form of utils.py:86
        assert abs(ref_lr[s] - 5e-4 * (1 + math.cos(math.pi * s / 20)) / 2) < 1e-12
    assert max(abs(a - b) for a, b in zip(ref_loss, o_loss)) < 1e-4, (ref_loss, o_loss)
    assert abs(ref_loss[0] - o_loss[0]) < 1e-5 and ref_loss[-1] < ref_loss[0]  # same start, and it trains
    sd = ref.state_dict()
    worst = max(rel_err(op[k].detach(), sd[k]) for k in op if op[k].requires_grad and not k.endswith("attn.qkv.bias"))
    print("20-step reference vs oracle: worst loss diff", max(abs(a - b) for a, b in zip(ref_loss, o_loss)),
          "worst param rel err", worst)
    assert worst < 1e-4
