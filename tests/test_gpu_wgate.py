"""GPU parity for the WGATE sibling model: the band-attention kernels through the
C-ABI vs the fp64 oracle (which evaluates the attention densely over all T*16 keys, like the reference),
and the whole `WGATEModel` vs the reference-generated fixtures.

fp32: bound 1e-3 relative (north_star), observed ~1e-6; bf16 storage: 1e-2.
"""
import importlib

import pytest
import torch

from oracle import wgat_oracle as OW
from libgemm_path import use_library_linears
from helpers import load_fixture, wgate_oracle_from_fixture, rel_err, grad_digest_check

pytestmark = pytest.mark.gpu
hw = importlib.import_module("sl-hwgat_amd")
HF = hw.functional
DEV = "cuda:0"
F32_TOL, BF16_TOL = 2e-5, 1e-2


def _oracle_attn(qkv, adj, n_heads, attn_keep=None):
    """natural-order qkv (B,F,K,3d) -> o (B,F,K,d) through the oracle's partition / dense attention / reverse"""
    B, F, K, d3 = qkv.shape
    d = d3 // 3
    hd = d // n_heads
    w = OW.to_windows(qkv).reshape(B, K // 16, F * 16, 3, n_heads, hd).permute(3, 0, 1, 4, 2, 5)
    o, _ = OW.band_attention(w[0], w[1], w[2], OW.additive_mask(adj.to(qkv.dtype)), attn_keep)
    return OW.from_windows(o, F)


def test_mfma16_operand_layout():
    g = torch.Generator().manual_seed(0)
    a = torch.randint(-4, 5, (16, 4), generator=g).float()
    b = torch.randint(-4, 5, (4, 16), generator=g).float()
    out = torch.empty(64, 4, device=DEV)
    ad, bd = a.to(DEV), b.to(DEV)
    hw._lib.call("hwgat_debug_mfma16x16x4", hw._lib.ptr(ad), hw._lib.ptr(bd), hw._lib.ptr(out), hw._lib.stream())
    out = out.cpu()
    d = a @ b
    for lane in range(64):
        for r in range(4):
            assert out[lane, r] == d[4 * (lane >> 4) + r, lane & 15], (lane, r)


@pytest.mark.parametrize("hd,nH,nW,F,B", [(16, 8, 2, 8, 2), (16, 2, 4, 37, 1), (32, 4, 3, 5, 2), (16, 4, 1, 1, 3),
                                          (32, 2, 2, 2, 1), (16, 8, 4, 64, 1),
                                          # head counts around the workgroup sizes of band_attn_f32.hip (8 heads forward where
                                          # nH % 8 == 0, else 4; 4 backward): two full groups, three groups of four, a part group
                                          (16, 16, 1, 9, 1), (16, 12, 2, 6, 1), (16, 6, 1, 20, 2)])
def test_band_attention_fwd_bwd(hd, nH, nW, F, B):
    g = torch.Generator().manual_seed(hd + nW + F)
    d, K = nH * hd, nW * 16
    qkv = torch.randn(B, F, K, 3 * d, generator=g) * 0.8
    do = torch.randn(B, F, K, d, generator=g)
    adj = OW.band_adjacency(F, nW)
    rows = HF.band_mask_rows(adj, F).to(DEV)

    ref_in = qkv.double().requires_grad_(True)
    ref = _oracle_attn(ref_in, adj, nH)
    ref.backward(do.double())

    x = qkv.to(DEV).requires_grad_(True)
    out = HF.band_attention(x, rows, nH)
    out.backward(do.to(DEV))
    assert rel_err(out.detach().cpu(), ref.detach()) < F32_TOL
    assert rel_err(x.grad.cpu(), ref_in.grad) < F32_TOL

    xb = qkv.to(DEV, torch.bfloat16).requires_grad_(True)
    refb_in = xb.detach().cpu().double().requires_grad_(True)
    refb = _oracle_attn(refb_in, adj, nH)
    refb.backward(do.double())
    outb = HF.band_attention(xb, rows, nH)
    outb.backward(do.to(DEV, torch.bfloat16))
    assert rel_err(outb.detach().float().cpu(), refb.detach()) < BF16_TOL
    assert rel_err(xb.grad.float().cpu(), refb_in.grad) < 2 * BF16_TOL


@pytest.mark.parametrize("hd,nH,nW,F,B", [(16, 8, 2, 8, 2), (16, 2, 4, 37, 1), (32, 4, 3, 5, 2), (16, 4, 1, 1, 3), (32, 2, 2, 2, 1)])
def test_band_attention_with_attention_dropout(hd, nH, nW, F, B):
    """attn_drop_rate > 0 (reference WGATE.py:81,103): the mask is the library's hash over the element index of the
    reference's DENSE (B nW, nH, F 16, F 16) attention tensor -- the kernels only ever evaluate its band --, so
    hwgat_dropout_mask_f32 hands the whole of it to the (dense) oracle; forward and backward, the fp32 kernels and the
    workgroup-staged bf16 kernels, ragged frame segments and a last head group of 2 included."""
    g = torch.Generator().manual_seed(7 * hd + nW + F)
    d, K, p_drop, seed = nH * hd, nW * 16, 0.2, 0xBEEF02
    qkv = torch.randn(B, F, K, 3 * d, generator=g) * 0.8
    do = torch.randn(B, F, K, d, generator=g)
    adj = OW.band_adjacency(F, nW)
    rows = HF.band_mask_rows(adj, F).to(DEV)
    keep = HF.dropout_mask((B, nW, nH, F * 16, F * 16), seed, p_drop, DEV).cpu().double()

    ref_in = qkv.double().requires_grad_(True)
    ref = _oracle_attn(ref_in, adj, nH, keep)
    ref.backward(do.double())
    x = qkv.to(DEV).requires_grad_(True)
    out = HF.band_attention(x, rows, nH, drop=(seed, p_drop))
    out.backward(do.to(DEV))
    assert rel_err(out.detach().cpu(), ref.detach()) < F32_TOL
    assert rel_err(x.grad.cpu(), ref_in.grad) < F32_TOL
    other = HF.band_attention(x.detach(), rows, nH, drop=(seed + 1, p_drop))
    assert rel_err(other.cpu(), ref.detach()) > 0.05
    assert torch.equal(HF.band_attention(x.detach(), rows, nH, drop=(seed, 0.0)), HF.band_attention(x.detach(), rows, nH))
    base = torch.tensor([77], dtype=torch.int32, device=DEV)
    assert torch.equal(HF.band_attention(x.detach(), rows, nH, drop=(seed - 77, p_drop, base)), out.detach())

    xb = qkv.to(DEV, torch.bfloat16).requires_grad_(True)
    refb_in = xb.detach().cpu().double().requires_grad_(True)
    refb = _oracle_attn(refb_in, adj, nH, keep)
    refb.backward(do.double())
    outb = HF.band_attention(xb, rows, nH, drop=(seed, p_drop))
    outb.backward(do.to(DEV, torch.bfloat16))
    assert rel_err(outb.detach().float().cpu(), refb.detach()) < BF16_TOL
    assert rel_err(xb.grad.float().cpu(), refb_in.grad) < 2 * BF16_TOL


def test_band_attention_general_blocks_and_rejections():
    """asymmetric off-diagonal blocks (prev != next) exercise the t = 0 / t = 2 bit fields separately"""
    g = torch.Generator().manual_seed(9)
    B, F, nW, nH, hd = 2, 7, 2, 2, 16
    d, K = nH * hd, nW * 16
    diag = ((torch.rand(nW, 16, 16, generator=g) < 0.3) | torch.eye(16, dtype=torch.bool)).float()
    prev = (torch.rand(nW, 16, 16, generator=g) < 0.2).float()
    nxt = (torch.rand(nW, 16, 16, generator=g) < 0.2).float()
    adj = torch.zeros(nW, F, 16, F, 16)
    for f in range(F):
        adj[:, f, :, f, :] = diag
        if f > 0:
            adj[:, f, :, f - 1, :] = prev
        if f + 1 < F:
            adj[:, f, :, f + 1, :] = nxt
    adj = adj.reshape(nW, F * 16, F * 16)
    rows = HF.band_mask_rows(adj, F).to(DEV)
    qkv = torch.randn(B, F, K, 3 * d, generator=g)
    do = torch.randn(B, F, K, d, generator=g)
    ref_in = qkv.double().requires_grad_(True)
    ref = _oracle_attn(ref_in, adj, nH)
    ref.backward(do.double())
    x = qkv.to(DEV).requires_grad_(True)
    out = HF.band_attention(x, rows, nH)
    out.backward(do.to(DEV))
    assert rel_err(out.detach().cpu(), ref.detach()) < F32_TOL
    assert rel_err(x.grad.cpu(), ref_in.grad) < F32_TOL
    # bf16 storage = the bf16-MFMA kernels (band_attn_bf16.hip): same masks, bf16-representable inputs
    xb = qkv.to(DEV, torch.bfloat16).requires_grad_(True)
    refb_in = xb.detach().cpu().double().requires_grad_(True)
    refb = _oracle_attn(refb_in, adj, nH)
    refb.backward(do.double())
    outb = HF.band_attention(xb, rows, nH)
    outb.backward(do.to(DEV, torch.bfloat16))
    assert rel_err(outb.detach().float().cpu(), refb.detach()) < BF16_TOL / 2
    assert rel_err(xb.grad.float().cpu(), refb_in.grad) < BF16_TOL / 2
    bad = adj.clone()
    bad[0, 0, 5 * 16 + 3] = 1                      # frame 0 sees frame 5
    with pytest.raises(NotImplementedError):
        HF.band_mask_rows(bad, F)
    bad = adj.clone()
    bad[1, 3 * 16 + 2, 3 * 16 + 9] = 1 - bad[1, 3 * 16 + 2, 3 * 16 + 9]      # one frame differs
    with pytest.raises(NotImplementedError):
        HF.band_mask_rows(bad, F)
    L = hw._lib
    o = torch.empty(B, F, K, d, device=DEV)
    assert L.lib().hwgat_band_attn_fwd(L.ptr(x), L.ptr(o), L.ptr(rows), B, F, nW, nH, 64, 0, None) < 0    # head_dim
    assert L.lib().hwgat_band_attn_fwd(L.ptr(x), L.ptr(o), None, B, F, nW, nH, 16, 0, None) < 0
    assert L.lib().hwgat_band_attn_bwd(L.ptr(x), L.ptr(o), L.ptr(x), L.ptr(rows), B, 0, nW, nH, 16, 0, None) < 0


def test_full_size_properties():
    """WGATE at the headline batch (B64 T128 K64 d128, 8 heads): size-independent properties"""
    B, F, nW, nH, hd = 64, 128, 4, 8, 16
    d, K = nH * hd, nW * 16
    g = torch.Generator(device=DEV).manual_seed(0)
    qkv = torch.randn(B, F, K, 3 * d, device=DEV, generator=g)
    rows = HF.band_mask_rows(OW.band_adjacency(F, nW), F).to(DEV)
    q1 = qkv.clone()
    q1[..., 2 * d:] = 1.0
    assert (HF.band_attention(q1, rows, nH) - 1).abs().max() < 1e-5                # rows of P sum to 1
    a = HF.band_attention(qkv, rows, nH)
    q2 = qkv.clone()
    q2[..., 2 * d:] *= -2.0
    assert (HF.band_attention(q2, rows, nH) + 2 * a).abs().max() < 1e-4            # linear in V
    perm = torch.randperm(B, device=DEV)
    assert torch.equal(HF.band_attention(qkv[perm].contiguous(), rows, nH), a[perm])
    # a clip computed alone (different frame segmentation: more segments at B = 1) gives the same bits
    assert torch.equal(HF.band_attention(qkv[5:6].contiguous(), rows, nH), a[5:6])
    ref = _oracle_attn(qkv[7:8, :, :16].cpu().double(), OW.band_adjacency(F, 1), nH)
    assert rel_err(a[7:8, :, :16].cpu(), ref) < F32_TOL
    x = qkv.clone().requires_grad_(True)
    HF.band_attention(x, rows, nH).sum().backward()
    assert (x.grad[..., 2 * d:].sum(dim=(1, 2)) - F * K).abs().max() < 0.05         # dO == 1: dV sums to #queries
    x = qkv.clone()
    x[..., d:2 * d] = 1.0                                                           # k == 1 -> dq = 0
    x.requires_grad_(True)
    g2 = torch.randn(B, F, K, d, device=DEV, generator=g)
    HF.band_attention(x, rows, nH).backward(g2)
    assert x.grad[..., :d].abs().max() < 1e-3
    # the fp32 backward (band_attn_f32.hip): batch order and frame segmentation do not change a bit, and one window of one
    # clip equals the dense fp64 oracle
    x = qkv.clone().requires_grad_(True)
    HF.band_attention(x, rows, nH).backward(g2)
    perm = torch.randperm(B, device=DEV)
    xp = qkv[perm].contiguous().requires_grad_(True)
    HF.band_attention(xp, rows, nH).backward(g2[perm].contiguous())
    assert torch.equal(xp.grad, x.grad[perm])
    x1 = qkv[5:6].contiguous().requires_grad_(True)
    HF.band_attention(x1, rows, nH).backward(g2[5:6].contiguous())
    assert torch.equal(x1.grad, x.grad[5:6])
    ref_in = qkv[7:8, :, :16].cpu().double().requires_grad_(True)
    _oracle_attn(ref_in, OW.band_adjacency(F, 1), nH).backward(g2[7:8, :, :16].cpu().double())
    assert rel_err(x.grad[7:8, :, :16].cpu(), ref_in.grad) < F32_TOL


def test_full_size_properties_bf16():
    """the bf16-MFMA kernels at the headline batch: properties that do not depend on the size, and independence of the
    frame segmentation (forward and backward cut the clip into segments by grid size; halo frames are recomputed)"""
    B, F, nW, nH, hd = 64, 128, 4, 8, 16
    d, K = nH * hd, nW * 16
    g = torch.Generator(device=DEV).manual_seed(1)
    qkv = torch.randn(B, F, K, 3 * d, device=DEV, generator=g).to(torch.bfloat16)
    do = torch.randn(B, F, K, d, device=DEV, generator=g).to(torch.bfloat16)
    rows = HF.band_mask_rows(OW.band_adjacency(F, nW), F).to(DEV)
    q1 = qkv.clone()
    q1[..., 2 * d:] = 1.0
    assert (HF.band_attention(q1, rows, nH).float() - 1).abs().max() < 2e-2          # rows of P (rounded to bf16) sum to 1
    x = qkv.clone().requires_grad_(True)
    a = HF.band_attention(x, rows, nH)
    a.backward(do)
    perm = torch.randperm(B, device=DEV)
    xp = qkv[perm].contiguous().requires_grad_(True)
    ap = HF.band_attention(xp, rows, nH)
    ap.backward(do[perm].contiguous())
    assert torch.equal(ap, a[perm]) and torch.equal(xp.grad, x.grad[perm])
    # one clip alone runs with more, shorter segments: same bits, forward and backward
    x1 = qkv[5:6].contiguous().requires_grad_(True)
    a1 = HF.band_attention(x1, rows, nH)
    a1.backward(do[5:6].contiguous())
    assert torch.equal(a1, a[5:6]) and torch.equal(x1.grad, x.grad[5:6])
    # one window of one clip against the dense fp64 oracle, forward and backward
    ref_in = qkv[7:8, :, :16].cpu().double().requires_grad_(True)
    ref = _oracle_attn(ref_in, OW.band_adjacency(F, 1), nH)
    ref.backward(do[7:8, :, :16].cpu().double())
    assert rel_err(a[7:8, :, :16].float().cpu(), ref.detach()) < BF16_TOL / 2
    assert rel_err(x.grad[7:8, :, :16].float().cpu(), ref_in.grad) < BF16_TOL / 2


# ------------------------------------------------------------------------------------------ whole model
def _model_from_fixture(fx, dtype=torch.float32):
    oracle, params, cfg = wgate_oracle_from_fixture(fx)
    hp = hw.WGATEParams({"src_len": cfg["temporal_dim"], "num_class": cfg["num_classes"]}, cfg["kp_dim"], DEV,
                        num_kps=cfg["num_kps"], embed_dim=cfg["embed_dim"])
    hp.num_heads, hp.depths, hp.drop_rate = cfg["num_heads"], cfg["depths"], 0.0
    model = hw.WGATEModel(*hp.get_model_params())
    res = model.load_state_dict(params, strict=False)
    assert not res.unexpected_keys and res.missing_keys == ["adj_mask"]
    return model.set_activation_dtype(dtype), params


@pytest.mark.parametrize("name", ["wgate_a.npz", "wgate_b.npz"])
@pytest.mark.parametrize("fused", [True, False])
def test_model_matches_reference_fixture(name, fused):
    fx = load_fixture(name)
    model, _ = _model_from_fixture(fx)
    if not fused:
        use_library_linears(model)
    x = torch.from_numpy(fx["x"]).to(DEV)
    y = torch.from_numpy(fx["y"]).to(DEV)
    crit = importlib.import_module("sl-hwgat_amd.train").SmoothedCrossEntropyLoss()
    for mode in ("eval", "train"):                    # drop 0: the same function
        getattr(model, mode)()
        model.zero_grad()
        logits = model(x)
        loss = crit(logits, y)
        loss.backward()
        assert rel_err(logits.detach().cpu(), fx["eval.logits"]) < 1e-4, mode
        assert abs(loss.item() - float(fx["evalbwd.loss"])) < 1e-4
        grad_digest_check({k: p.grad for k, p in model.named_parameters() if p.grad is not None}, fx, "evalbwd.", 1e-3)
    model.eval()
    with torch.no_grad():
        assert rel_err(model.forward_features(x).cpu(), fx["eval.feat"]) < 1e-4


def test_model_bf16_and_seeded_dropout():
    fx = load_fixture("wgate_a.npz")
    model, params = _model_from_fixture(fx, torch.bfloat16)
    model.eval()
    x = torch.from_numpy(fx["x"]).to(DEV)
    with torch.no_grad():
        logits = model(x)
    assert rel_err(logits.float().cpu(), fx["eval.logits"]) < 1e-2                  # BASELINE configs[2] contract: bf16 within 1e-2 of fp32
    model.set_activation_dtype(torch.float32)
    model.drop_rate = 0.1
    model.train()
    a = model(x)
    model._drop_calls = 0
    b = model(x)
    assert torch.allclose(a, b, atol=1e-6) and torch.isfinite(a).all()
    model.eval()
    assert (model(x) - a).abs().max() > 1e-3
    a.sum().backward()
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
