"""The whole train step in a HIP graph (train.GraphedTrainStep): device-resident dropout seeds
make a replay draw the masks the eager step draws; losses and weights follow the eager run (reference loop:
hwgat/utils.py:93-116, dropout sites hwgat/models/HWGATE.py:27,116,133,135)."""
import importlib

import pytest
import torch

pytestmark = pytest.mark.gpu
hw = importlib.import_module("sl-hwgat_amd")
HF = hw.functional
train = importlib.import_module("sl-hwgat_amd.train")
DEV = torch.device("cuda:0")


def _build(dtype, kind="hwgate"):
    torch.manual_seed(11)
    if kind == "hwgate":
        hp = hw.HWGATEParams({"src_len": 16, "num_class": 7}, 2, DEV, num_kps=32)
        model = hw.Model(*hp.get_model_params()).to(DEV)
    elif kind == "hgate":
        hp = hw.HGATEParams({"src_len": 16, "num_class": 7}, 2, DEV)
        model = hw.HGATEModel(*hp.get_model_params()).to(DEV)
    else:
        hp = hw.WGATEParams({"src_len": 16, "num_class": 7}, 2, DEV, num_kps=32)
        model = hw.WGATEModel(*hp.get_model_params()).to(DEV)
    assert model.drop_rate == 0.1                         # the reference default: dropout is ON in these runs
    model.set_activation_dtype(dtype)
    model.train()
    if kind == "hwgate":
        model.threshold_override = [0.3, 0.1, 0.5, 0.2, 0.07, 0.4, 0.25, 0.6]   # HWGATE.py:96 draws these at random
    return model


def _batch(model, B=8):
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.rand(B, model.temporal_dim, model.num_kps, model.kp_dim, device=DEV, generator=g)
    y = torch.randint(0, 7, (B,), device=DEV, generator=g)
    return x, y


def test_device_seed_word_is_the_host_mirror_and_enters_every_mask():
    """hwgat_seed_set / hwgat_seed_advance leave in state[1] what functional.seed_base_value computes on the host, and a
    kernel given (site seed, seed_base) draws the mask of the plain seed site + base"""
    st = torch.zeros(4, dtype=torch.int32, device=DEV)
    HF.seed_set(st, 41, 1001, 3)
    want = [41, HF.seed_base_value(1001, 41, 3), 1001, 3]
    assert [v & 0xFFFFFFFF for v in st.tolist()] == want
    HF.seed_advance(st)
    HF.seed_advance(st)
    assert [v & 0xFFFFFFFF for v in st.tolist()] == [43, HF.seed_base_value(1001, 43, 3), 1001, 3]
    base = HF.seed_base_value(1001, 43, 3)
    for site in (0, 12345, 0xFFFFFFF0):                   # incl. 32-bit wrap of site + base
        a = HF.dropout_mask((4096,), site, 0.3, DEV, seed_base=st[1:2])
        b = HF.dropout_mask((4096,), (site + base) & 0xFFFFFFFF, 0.3, DEV)
        assert torch.equal(a, b)
    assert not torch.equal(HF.dropout_mask((4096,), 5, 0.3, DEV, seed_base=st[1:2]), HF.dropout_mask((4096,), 5, 0.3, DEV))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("kind", ["hwgate", "hgate", "wgate"])
def test_graphed_train_step_equals_the_eager_step(dtype, kind):
    steps, c0 = 5, 17
    # --- eager
    m1 = _build(dtype, kind)
    x, y = _batch(m1)
    o1 = torch.optim.AdamW([p for p in m1.parameters() if p.requires_grad], lr=5e-4, fused=True, capturable=True)
    m1._drop_calls = c0
    s1 = train.TrainStep(m1, o1, None)
    eager_loss, eager_masks = [], []
    for _ in range(steps):
        eager_loss.append(s1(x, y).clone())
        eager_masks.append(HF.dropout_mask((2048,), m1._site_seeds(2)[1], 0.5, DEV, seed_base=m1._seed_base()))
    # --- graphed, from the same weights / optimizer state / step counter
    m2 = _build(dtype, kind)
    o2 = torch.optim.AdamW([p for p in m2.parameters() if p.requires_grad], lr=5e-4, fused=True, capturable=True)
    m2._drop_calls = c0
    w0 = [p.detach().clone() for p in m2.parameters()]
    s2 = train.GraphedTrainStep(m2, o2, x, y)
    assert all(torch.equal(a, b.detach()) for a, b in zip(w0, m2.parameters()))      # capture left the weights alone
    assert m2._drop_calls == c0
    graph_loss = []
    for k in range(steps):
        graph_loss.append(s2(x, y).clone())
        # the dropout-only probe: the mask a kernel draws NOW from the device word, replay k vs eager step k -- identical
        probe = HF.dropout_mask((2048,), m2._site_seeds(2)[1], 0.5, DEV, seed_base=m2._seed_base())
        assert torch.equal(probe, eager_masks[k]), k
        assert torch.equal(probe, HF.dropout_mask((2048,), m2._seeds(2)[1], 0.5, DEV)), k     # and the host mirror agrees
    assert not torch.equal(eager_masks[0], eager_masks[1])                                   # fresh masks every step
    tol = 2e-5 if dtype == torch.float32 else 2e-2        # fp32: summation-order noise of the atomics; bf16: storage rounding of a different order
    for k in range(steps):
        a, b = float(eager_loss[k]), float(graph_loss[k])
        assert abs(a - b) <= tol * max(1.0, abs(a)), (k, a, b)
    assert float(graph_loss[-1]) < float(graph_loss[0])                                     # and it trains
    # weights follow the optimizer: the UPDATE the graphed run applied is the eager run's.  (Entry-wise equality is not a
    # meaningful bar under Adam: an entry whose gradient is ~0 moves by +-lr per step on the sign of summation-order noise.)
    num = den = 0.0
    for a, p1, p2 in zip(w0, m1.parameters(), m2.parameters()):
        if p1.requires_grad:
            u1, u2 = (p1.detach() - a).double(), (p2.detach() - a).double()
            num += float((u1 - u2).pow(2).sum())
            den += float(u1.pow(2).sum())
    assert den > 0 and (num / den) ** 0.5 < (0.02 if dtype == torch.float32 else 0.25), (num / den) ** 0.5
    # shape guard and reallocation guard
    with pytest.raises(ValueError):
        s2(x[:4], y[:4])
    w = m2.head.weight
    w.data = w.data.clone()
    with pytest.raises(RuntimeError, match="capture again"):
        s2(x, y)


def test_graphed_step_needs_a_capturable_optimizer_and_train_mode():
    m = _build(torch.float32)
    x, y = _batch(m)
    with pytest.raises(ValueError, match="capturable"):
        train.GraphedTrainStep(m, torch.optim.AdamW(m.parameters(), lr=5e-4, fused=True), x, y)
    o = torch.optim.AdamW([p for p in m.parameters() if p.requires_grad], lr=5e-4, fused=True, capturable=True)
    with pytest.raises(ValueError, match="train"):
        train.GraphedTrainStep(m.eval(), o, x, y)
