"""GPU parity: every HIP kernel (through the C-ABI) vs the fp64 oracle on seeded inputs.

fp32 kernels: bound 1e-3 relative (north_star), observed ~1e-6; bf16 storage: 1e-2.
"""
import importlib

import pytest
import torch

from oracle import hwgat_oracle as O
from helpers import rel_err, tie_free_threshold

pytestmark = pytest.mark.gpu
hw = importlib.import_module("sl-hwgat_amd")
HF = hw.functional
DEV = "cuda:0"
F32_TOL, BF16_TOL = 2e-5, 1e-2


def _windows(qkv, n_heads, shifted):
    """natural-order qkv (B,F,K,3d) -> (q, k, v) of shape (B, f, nW, nH, 32, hd) after the oracle's roll / partition"""
    B, F, K, d3 = qkv.shape
    d, nW = d3 // 3, K // 16
    hd = d // n_heads
    x = torch.roll(qkv, -1, 1) if shifted else qkv
    w = O.to_windows(x)                                             # (B,f,nW,32,3d)
    return w.reshape(B, F // 2, nW, 32, 3, n_heads, hd).permute(4, 0, 1, 2, 5, 3, 6)


def _oracle_attn(qkv, adj, n_heads, shifted, thr, attn_keep=None):
    """natural-order qkv (B,F,K,3d) fp64 -> o (B,F,K,d) with the oracle's
    roll / partition / attention / reverse / roll chain"""
    B, F, K, d3 = qkv.shape
    nW = K // 16
    w = _windows(qkv, n_heads, shifted)
    sm = O.shift_mask(F, nW, 2, 1, qkv.dtype).view(F // 2, nW, 32, 32) if shifted else None
    o, _ = O.window_attention(w[0], w[1], w[2], adj.to(qkv.dtype), sm, thr, attn_keep)
    o = O.from_windows(o)
    return torch.roll(o, 1, 1) if shifted else o


def _unmasked_p0(qkv, n_heads, shifted):
    """the selector's input (HWGATE.py:97): softmax over all 32 keys of the UNMASKED scaled scores, fp64"""
    w = _windows(qkv.detach(), n_heads, shifted)
    hd = w.shape[-1]
    return torch.softmax((w[0] * hd ** -0.5) @ w[1].transpose(-2, -1), dim=-1)


def _entrywise(a, b):
    """largest entry-wise error relative to the largest reference entry"""
    a, b = torch.as_tensor(a, dtype=torch.float64), torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def test_mfma_operand_layout():
    g = torch.Generator().manual_seed(0)
    a = torch.randint(-4, 5, (32, 2), generator=g).float()
    b = torch.randint(-4, 5, (2, 32), generator=g).float()          # asymmetric on purpose
    out = torch.empty(64, 16, device=DEV)
    ad, bd = a.to(DEV), b.to(DEV)                                   # keep alive across the launch
    hw._lib.call("hwgat_debug_mfma32x32x2", hw._lib.ptr(ad), hw._lib.ptr(bd), hw._lib.ptr(out), hw._lib.stream())
    out = out.cpu()
    d = a @ b
    for lane in range(64):
        for r in range(16):
            row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
            assert out[lane, r] == d[row, lane & 31], (lane, r)


@pytest.mark.parametrize("hd,nH,nW,F,B", [(64, 2, 2, 8, 2), (64, 4, 5, 4, 3), (128, 2, 1, 6, 2), (32, 4, 3, 4, 1)])
@pytest.mark.parametrize("shifted", [False, True])
@pytest.mark.parametrize("thr", [None, 0.07, 0.0005, 0.9999])
def test_window_attention_fwd_bwd(hd, nH, nW, F, B, shifted, thr):
    g = torch.Generator().manual_seed(hd + nW + F)
    d, K = nH * hd, nW * 16
    qkv = torch.randn(B, F, K, 3 * d, generator=g) * 0.8
    do = torch.randn(B, F, K, d, generator=g)
    adj = O.window_adjacency(nW)
    bits = HF.mask_bits(adj).to(DEV)
    thr_t = None if thr is None else torch.tensor([thr], device=DEV)

    ref_in = qkv.double().requires_grad_(True)
    ref = _oracle_attn(ref_in, adj, nH, shifted, thr)
    ref.backward(do.double())

    x = qkv.to(DEV).requires_grad_(True)
    out = HF.window_attention(x, bits, thr_t, nH, shifted)
    out.backward(do.to(DEV))
    assert rel_err(out.detach().cpu(), ref.detach()) < F32_TOL
    assert rel_err(x.grad.cpu(), ref_in.grad) < F32_TOL

    # bf16 storage (config 3): same math, bf16 in/out.  The train-mode selector [P0 <= thr] (HWGATE.py:94-100) is a
    # discontinuity: the threshold is moved to the centre of the widest gap of the fp64 oracle's P0 (on the SAME
    # bf16-rounded inputs) near the nominal value, so that no probability is within `margin` of it -- the kernel forms S
    # from the bf16 operands with fp32 accumulation (error ~1e-6), far inside that margin -- and the bf16 result is then
    # held to the bf16 tolerance in TRAIN mode as well, in norm and entry by entry, output and all three gradients.
    xb = qkv.to(DEV, torch.bfloat16).requires_grad_(True)
    refb_in = xb.detach().cpu().double().requires_grad_(True)
    thr_b, thr_bt = thr, thr_t
    if thr is not None:
        thr_b, margin = tie_free_threshold(_unmasked_p0(refb_in, nH, shifted), thr)
        assert margin > 2e-4, (thr, thr_b, margin)
        thr_bt = torch.tensor([thr_b], device=DEV)
    refb = _oracle_attn(refb_in, adj, nH, shifted, thr_b)
    refb.backward(do.double())
    outb = HF.window_attention(xb, bits, thr_bt, nH, shifted)
    outb.backward(do.to(DEV, torch.bfloat16))
    assert rel_err(outb.detach().float().cpu(), refb.detach()) < BF16_TOL
    assert rel_err(xb.grad.float().cpu(), refb_in.grad) < 2 * BF16_TOL
    assert _entrywise(outb.detach().float().cpu(), refb.detach()) < 2e-2
    for part in range(3):                                            # dq, dk, dv separately: each against its own scale
        assert _entrywise(xb.grad.float().cpu()[..., part * d:(part + 1) * d], refb_in.grad[..., part * d:(part + 1) * d]) < 2e-2, part
    if thr is not None and thr < 0.5:
        # the threshold has teeth here: ignoring it (eval mode) is far outside the tolerance
        assert rel_err(HF.window_attention(xb.detach(), bits, None, nH, shifted).float().cpu(), refb.detach()) > 5 * BF16_TOL


@pytest.mark.parametrize("hd,nH,nW,F,B", [(64, 2, 2, 8, 2), (128, 2, 1, 6, 2), (32, 4, 3, 4, 1)])
@pytest.mark.parametrize("shifted", [False, True])
def test_window_attention_with_attention_dropout(hd, nH, nW, F, B, shifted):
    """attn_drop_rate > 0 (reference HWGATE.py:78,112): the kernels' mask is the library's own hash over the element
    index of the reference's (B f nW, nH, 32, 32) attention tensor, so hwgat_dropout_mask_f32 hands it to the oracle;
    forward and backward (which recomputes the mask), fp32 and bf16 storage, all three backward kernels (hd 128 = the
    two-wave split form).  The threshold is set where no selector is close to a tie."""
    g = torch.Generator().manual_seed(3 * hd + nW + F)
    d, K, p_drop, seed, thr = nH * hd, nW * 16, 0.2, 0xC0FFEE, 0.9999
    qkv = torch.randn(B, F, K, 3 * d, generator=g) * 0.8
    do = torch.randn(B, F, K, d, generator=g)
    adj = O.window_adjacency(nW)
    bits = HF.mask_bits(adj).to(DEV)
    thr_t = torch.tensor([thr], device=DEV)
    keep = HF.dropout_mask((B, F // 2, nW, nH, 32, 32), seed, p_drop, DEV).cpu().double()
    kept = keep[keep > 0]
    assert float((kept - 1.0 / (1.0 - p_drop)).abs().max()) < 1e-6 and abs(kept.numel() / keep.numel() - (1 - p_drop)) < 0.02

    ref_in = qkv.double().requires_grad_(True)
    ref = _oracle_attn(ref_in, adj, nH, shifted, thr, keep)
    ref.backward(do.double())
    x = qkv.to(DEV).requires_grad_(True)
    out = HF.window_attention(x, bits, thr_t, nH, shifted, drop=(seed, p_drop))
    out.backward(do.to(DEV))
    assert rel_err(out.detach().cpu(), ref.detach()) < F32_TOL
    assert rel_err(x.grad.cpu(), ref_in.grad) < F32_TOL
    # a different seed is a different mask; p = 0 is the plain kernel, bit for bit
    other = HF.window_attention(x.detach(), bits, thr_t, nH, shifted, drop=(seed + 1, p_drop))
    assert rel_err(other.cpu(), ref.detach()) > 0.05
    assert torch.equal(HF.window_attention(x.detach(), bits, thr_t, nH, shifted, drop=(seed, 0.0)),
                       HF.window_attention(x.detach(), bits, thr_t, nH, shifted))

    xb = qkv.to(DEV, torch.bfloat16).requires_grad_(True)
    refb_in = xb.detach().cpu().double().requires_grad_(True)
    refb = _oracle_attn(refb_in, adj, nH, shifted, thr, keep)
    refb.backward(do.double())
    outb = HF.window_attention(xb, bits, thr_t, nH, shifted, drop=(seed, p_drop))
    outb.backward(do.to(DEV, torch.bfloat16))
    assert rel_err(outb.detach().float().cpu(), refb.detach()) < BF16_TOL
    assert rel_err(xb.grad.float().cpu(), refb_in.grad) < 2 * BF16_TOL

    # eval mode has no dropout: asking for it without the train-mode threshold is an error, in Python and in the C-ABI
    with pytest.raises(ValueError):
        HF.window_attention(x.detach(), bits, None, nH, shifted, drop=(seed, p_drop))
    L = hw._lib
    o = torch.empty(B, F, K, d, device=DEV)
    assert L.lib().hwgat_win_attn_fwd_drop(L.ptr(x), L.ptr(o), L.ptr(bits), None, B, F, nW, nH, hd, int(shifted), 0, seed, p_drop, None, None) < 0
    assert L.lib().hwgat_win_attn_fwd_drop(L.ptr(x), L.ptr(o), L.ptr(bits), L.ptr(thr_t), B, F, nW, nH, hd, int(shifted), 0, seed, 1.0, None, None) < 0


def test_window_attention_edge_rows():
    """exact-zero logits and fully masked rows (SURVEY 8a consequences i-iii)"""
    g = torch.Generator().manual_seed(5)
    B, F, nW, nH, hd = 1, 4, 2, 2, 64
    d, K = nH * hd, nW * 16
    qkv = torch.randn(B, F, K, 3 * d, generator=g)
    qkv[0, 0, 3, :d] = 0.0            # a query row of zeros -> all its logits are exactly 0 -> uniform 1/32
    qkv[0, 1, 5, d:2 * d] = 0.0       # a key row of zeros  -> that key's logit is exactly 0 for every query
    adj = O.window_adjacency(nW)
    bits = HF.mask_bits(adj).to(DEV)
    for shifted in (False, True):
        for thr in (None, 1e-6):      # thr ~ 0 drops everything: every row uniform over all 32 keys
            ref_in = qkv.double().requires_grad_(True)
            ref = _oracle_attn(ref_in, adj, nH, shifted, thr)
            ref.sum().backward()
            x = qkv.to(DEV).requires_grad_(True)
            t = None if thr is None else torch.tensor([thr], device=DEV)
            out = HF.window_attention(x, bits, t, nH, shifted)
            out.sum().backward()
            assert rel_err(out.detach().cpu(), ref.detach()) < F32_TOL, (shifted, thr)
            assert (x.grad.cpu() - ref_in.grad).abs().max() < 1e-5, (shifted, thr)
    # uniform row really is the mean of V over the whole window
    out = HF.window_attention(qkv.to(DEV), bits, None, nH, False).cpu()
    v = qkv[0, 0:2, 0:16, 2 * d:].reshape(32, d)
    assert torch.allclose(out[0, 0, 3], v.mean(0), atol=1e-5)


@pytest.mark.parametrize("d", [128, 256, 512, 1024])
def test_layer_norm_fwd_bwd(d):
    g = torch.Generator().manual_seed(d)
    for n in (7, 64, 1000):
        x = torch.randn(n, d, generator=g) * 2 + 0.5
        w, b = torch.randn(d, generator=g), torch.randn(d, generator=g)
        dy = torch.randn(n, d, generator=g)
        xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
        O.layer_norm(xr, wr, br).backward(dy.double())
        xg, wg, bg = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
        y = HF.layer_norm(xg, wg, bg)
        y.backward(dy.to(DEV))
        assert rel_err(y.detach().cpu(), O.layer_norm(x.double(), w.double(), b.double())) < F32_TOL
        assert rel_err(xg.grad.cpu(), xr.grad) < F32_TOL
        assert rel_err(wg.grad.cpu(), wr.grad) < F32_TOL
        assert rel_err(bg.grad.cpu(), br.grad) < F32_TOL
        xb = x.to(DEV, torch.bfloat16)
        yb = HF.layer_norm(xb, wg.detach(), bg.detach())
        assert rel_err(yb.float().cpu(), O.layer_norm(xb.cpu().double(), w.double(), b.double())) < BF16_TOL


def test_ln_bwd_residual_argument():
    g = torch.Generator().manual_seed(1)
    n, d = 50, 256
    x, dy, res = (torch.randn(n, d, generator=g).to(DEV) for _ in range(3))
    w = torch.randn(d, generator=g).to(DEV)
    mean, rstd = torch.empty(n, device=DEV), torch.empty(n, device=DEV)
    y = torch.empty_like(x)
    L = hw._lib
    L.call("hwgat_ln_fwd", L.ptr(x), L.ptr(w), L.ptr(w), L.ptr(y), L.ptr(mean), L.ptr(rstd), n, d, 0, L.stream())
    dx0, dx1 = torch.empty_like(x), torch.empty_like(x)
    dg = torch.zeros(4, d, device=DEV)
    L.call("hwgat_ln_bwd", L.ptr(dy), L.ptr(x), L.ptr(mean), L.ptr(rstd), L.ptr(w), None, L.ptr(dx0),
           L.ptr(dg[0]), L.ptr(dg[1]), n, d, 0, L.stream())
    L.call("hwgat_ln_bwd", L.ptr(dy), L.ptr(x), L.ptr(mean), L.ptr(rstd), L.ptr(w), L.ptr(res), L.ptr(dx1),
           L.ptr(dg[2]), L.ptr(dg[3]), n, d, 0, L.stream())
    assert torch.allclose(dx1, dx0 + res, atol=1e-6)
    assert torch.equal(dg[0], dg[2]) or torch.allclose(dg[0], dg[2], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("d", [128, 256, 512])
def test_masked_gradient_copies_equal_gradient_times_dropout_mask(d, dtype):
    """hwgat_ln_bwd_masked, hwgat_unmerge_masked, hwgat_lnpool_bwd_masked: the second output is the first one times the
    very mask hwgat_dropout_mask_f32 generates for (seed, element index); the first output is what the plain entry
    point writes."""
    g = torch.Generator().manual_seed(d)
    L = hw._lib
    dc = 0 if dtype == torch.float32 else 1
    seed, p = 4711, 0.1
    tol = 1e-6 if dtype == torch.float32 else 8e-3

    def masked_ok(plain, masked, shape):
        m = HF.dropout_mask(shape, seed, p, DEV).view_as(plain)
        return rel_err(masked.float().cpu(), (plain.float() * m).cpu().double()) < tol

    # --- LayerNorm backward
    n = 300
    x, dy, res = (torch.randn(n, d, generator=g).to(DEV).to(dtype) for _ in range(3))
    w = torch.randn(d, generator=g).to(DEV)
    mean, rstd = HF.ln_stats(x, w, w)
    dx0, dx1, dxm = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    dg = torch.zeros(4, d, device=DEV)
    L.call("hwgat_ln_bwd", L.ptr(dy), L.ptr(x), L.ptr(mean), L.ptr(rstd), L.ptr(w), L.ptr(res), L.ptr(dx0),
           L.ptr(dg[0]), L.ptr(dg[1]), n, d, dc, L.stream())
    L.call("hwgat_ln_bwd_masked", L.ptr(dy), L.ptr(x), L.ptr(mean), L.ptr(rstd), L.ptr(w), L.ptr(res), L.ptr(dx1),
           L.ptr(dg[2]), L.ptr(dg[3]), n, d, dc, L.ptr(dxm), seed, p, None, L.stream())
    assert torch.equal(dx0, dx1) and torch.allclose(dg[0], dg[2], rtol=1e-5, atol=1e-5)
    if dtype == torch.float32:
        assert masked_ok(dx1, dxm, (n, d))
    else:                                           # the masked copy is rounded from the fp32 value, not from the rounded dx
        m = HF.dropout_mask((n, d), seed, p, DEV)
        assert rel_err(dxm.float().cpu(), (dx1.float() * m).cpu().double()) < tol
    # --- un-merge
    B, F, K = 2, 8, 32
    merged = torch.randn(B, F // 2, K, 2 * d, generator=g).to(DEV).to(dtype)
    nat0, nat1, natm = (torch.empty(B, F, K, d, device=DEV, dtype=dtype) for _ in range(3))
    L.call("hwgat_merge", L.ptr(merged), L.ptr(nat0), B, F, K, d, 1, dc, L.stream())
    L.call("hwgat_unmerge_masked", L.ptr(merged), L.ptr(nat1), L.ptr(natm), B, F, K, d, dc, seed, p, None, L.stream())
    assert torch.equal(nat0, nat1)
    assert masked_ok(nat1, natm, (B, F, K, d))
    # --- pooled LayerNorm backward
    n_tok = 96
    x = torch.randn(B, n_tok, d, generator=g).to(DEV).to(dtype)
    gvec = torch.randn(B, d, generator=g).to(DEV)
    mean, rstd = HF.ln_stats(x.view(B * n_tok, d), w, w)
    p0, p1, pm = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    L.call("hwgat_lnpool_bwd", L.ptr(gvec), L.ptr(x), L.ptr(mean), L.ptr(rstd), L.ptr(p0), B, n_tok, d, dc, L.stream())
    L.call("hwgat_lnpool_bwd_masked", L.ptr(gvec), L.ptr(x), L.ptr(mean), L.ptr(rstd), L.ptr(p1), B, n_tok, d, dc, L.ptr(pm),
           seed, p, None, L.stream())
    assert torch.equal(p0, p1)
    if dtype == torch.float32:
        assert masked_ok(p1, pm, (B, n_tok, d))
    else:
        m = HF.dropout_mask((B, n_tok, d), seed, p, DEV)
        assert rel_err(pm.float().cpu(), (p1.float() * m).cpu().double()) < tol


@pytest.mark.parametrize("C,d0,J,nW", [(2, 128, 29, 4), (3, 256, 133, 7), (2, 128, 32, 2)])
def test_embed(C, d0, J, nW):
    g = torch.Generator().manual_seed(C * 7 + nW)
    B, T, K = 2, 8, nW * 16
    x = torch.rand(B, T, J, C, generator=g)
    bmat = torch.randn(d0 // 2, C, generator=g) * 10
    pe = O.sinusoid_table(T, d0)
    idx = hw.part_table(J, nW) if J != K else None
    xs = x[:, :, idx.long()] if idx is not None else x
    # the fp32 oracle is the comparison point here: the argument 2*pi*x.B is itself only
    # fp32-accurate (|arg| ~ 1e2 rad -> abs error ~1e-5), so compare with an absolute bound
    ref = O.fourier_embed(xs, bmat) + pe[:, :T]
    out = HF.embed(x.to(DEV), None if idx is None else idx.to(DEV), bmat.to(DEV),
                   pe.view(T, d0).to(DEV), K)
    assert (out.cpu() - ref).abs().max() < 2e-4
    ref64 = O.fourier_embed(xs.double(), bmat.double()) + pe[:, :T].double()
    assert (out.cpu().double() - ref64).abs().max() < 2e-4
    out_nope = HF.embed(x.to(DEV), None if idx is None else idx.to(DEV), bmat.to(DEV), None, K)
    assert (out_nope.cpu() - O.fourier_embed(xs, bmat)).abs().max() < 2e-4
    # bf16 output: hardware sine / cosine of the angle in revolutions (embed.hip); the error is the output rounding
    # (values up to 2: half an ulp = 3.9e-3 + the ~2e-5 of the argument)
    out_b = HF.embed(x.to(DEV), None if idx is None else idx.to(DEV), bmat.to(DEV), pe.view(T, d0).to(DEV), K,
                     out_dtype=torch.bfloat16)
    err = (out_b.cpu().double() - ref64).abs()
    assert err.max() < 4.2e-3 and err.mean() < 1.2e-3
    assert torch.equal(out_b.cpu(), ref64.to(torch.bfloat16)) or (out_b.cpu() != ref64.to(torch.bfloat16)).double().mean() < 0.02


def test_merge_roundtrip():
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 8, 32, 128, generator=g)
    ref = x.reshape(2, 4, 2, 32, 128).transpose(2, 3).reshape(2, 4, 32, 256)
    xg = x.to(DEV).requires_grad_(True)
    out = HF.temporal_merge(xg)
    assert torch.equal(out.detach().cpu(), ref)
    out.backward(out.detach())
    assert torch.equal(xg.grad.cpu(), x)
    xb = x.to(DEV, torch.bfloat16)
    assert torch.equal(HF.temporal_merge(xb).cpu(), ref.to(torch.bfloat16))


@pytest.mark.parametrize("d,n_tok,B", [(512, 640, 3), (128, 37, 2), (1024, 16, 1)])
def test_ln_mean_pool(d, n_tok, B):
    g = torch.Generator().manual_seed(d + n_tok)
    x = torch.randn(B, n_tok, d, generator=g) * 1.5
    w, b = torch.randn(d, generator=g), torch.randn(d, generator=g)
    df = torch.randn(B, d, generator=g)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    ref = O.layer_norm(xr, wr, br).mean(1)
    ref.backward(df.double())
    xg, wg, bg = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
    out = HF.ln_mean_pool(xg, wg, bg)
    out.backward(df.to(DEV))
    assert rel_err(out.detach().cpu(), ref.detach()) < F32_TOL
    assert rel_err(xg.grad.cpu(), xr.grad) < F32_TOL
    assert rel_err(wg.grad.cpu(), wr.grad) < 1e-4
    assert rel_err(bg.grad.cpu(), br.grad) < F32_TOL


def test_full_size_properties_config2():
    """BASELINE config 2 (B64 T128 K80 d128): size-independent properties of the
    attention kernel where the oracle would take minutes."""
    B, F, nW, nH, hd = 64, 128, 5, 2, 64
    d, K = nH * hd, nW * 16
    g = torch.Generator(device=DEV).manual_seed(0)
    qkv = torch.randn(B, F, K, 3 * d, device=DEV, generator=g)
    bits = HF.mask_bits(O.window_adjacency(nW)).to(DEV)
    for shifted in (False, True):
        # (1) rows of P sum to 1: V == 1 -> O == 1 everywhere
        q1 = qkv.clone()
        q1[..., 2 * d:] = 1.0
        o = HF.window_attention(q1, bits, None, nH, shifted)
        assert (o - 1).abs().max() < 1e-5
        # (2) linear in V
        a = HF.window_attention(qkv, bits, None, nH, shifted)
        q2 = qkv.clone()
        q2[..., 2 * d:] *= -2.0
        b2 = HF.window_attention(q2, bits, None, nH, shifted)
        assert (b2 + 2 * a).abs().max() < 1e-4
        # (3) clips are independent: a batch permutation permutes the output
        perm = torch.randperm(B, device=DEV)
        assert torch.equal(HF.window_attention(qkv[perm].contiguous(), bits, None, nH, shifted), a[perm])
        # (4) a sampled clip agrees with the oracle
        ref = _oracle_attn(qkv[7:8, :8].cpu().double(), O.window_adjacency(nW), nH, False, None)
        if not shifted:
            got = HF.window_attention(qkv[7:8, :8].contiguous(), bits, None, nH, False)
            assert rel_err(got.cpu(), ref) < F32_TOL
    # backward: sum over dqkv of V-part equals P^T dO column sums -> with dO == 1, dV sums to 32 per window/channel
    x = qkv.clone().requires_grad_(True)
    HF.window_attention(x, bits, None, nH, False).sum().backward()
    dv = x.grad[..., 2 * d:]
    per_window = dv.view(B, F // 2, 2, nW, 16, d).sum(dim=(2, 4))
    assert (per_window - 32).abs().max() < 1e-3
    # rows of dS sum to zero: with every key vector identical (k == 1) dq = scale * rowsum(dS) * 1 = 0
    x = qkv.clone()
    x[..., d:2 * d] = 1.0
    x.requires_grad_(True)
    g2 = torch.randn(B, F, K, d, device=DEV, generator=g)
    HF.window_attention(x, bits, None, nH, True).backward(g2)
    assert x.grad[..., :d].abs().max() < 1e-3


def test_full_size_properties_config5_head_dim_128():
    """BASELINE config 5 shape of one micro-batch (B16 T256 K112 d256, head_dim 128): the two-waves-per-unit backward
    (head-dim halves, S / dP partial sums exchanged through LDS) against size-independent properties, a sampled clip
    against the fp64 oracle (the library has no switch back to the one-wave kernel: the comparison is by value)."""
    B, F, nW, nH, hd = 16, 256, 7, 2, 128
    d, K = nH * hd, nW * 16
    g = torch.Generator(device=DEV).manual_seed(3)
    qkv = torch.randn(B, F, K, 3 * d, device=DEV, generator=g)
    bits = HF.mask_bits(O.window_adjacency(nW)).to(DEV)
    # dO == 1: dV sums to 32 per window and channel (columns of P^T sum over 32 queries of rows that sum to 1)
    x = qkv.clone().requires_grad_(True)
    HF.window_attention(x, bits, None, nH, False).sum().backward()
    per_window = x.grad[..., 2 * d:].view(B, F // 2, 2, nW, 16, d).sum(dim=(2, 4))
    assert (per_window - 32).abs().max() < 1e-3
    # rows of dS sum to zero: identical keys -> dq == 0 for any dO
    x = qkv.clone()
    x[..., d:2 * d] = 1.0
    x.requires_grad_(True)
    g2 = torch.randn(B, F, K, d, device=DEV, generator=g)
    HF.window_attention(x, bits, None, nH, True).backward(g2)
    assert x.grad[..., :d].abs().max() < 1e-3
    # a sampled clip, train-mode threshold, shifted: all three gradients vs the fp64 oracle
    for shifted, thr in ((False, None), (True, 0.07)):
        xs = qkv[5:6, :6].contiguous().requires_grad_(True)
        gs = g2[5:6, :6].contiguous()
        tt = None if thr is None else torch.tensor([thr], device=DEV)
        HF.window_attention(xs, bits, tt, nH, shifted).backward(gs)
        xr = qkv[5:6, :6].cpu().double().requires_grad_(True)
        _oracle_attn(xr, O.window_adjacency(nW), nH, shifted, thr).backward(gs.cpu().double())
        assert rel_err(xs.grad.cpu(), xr.grad) < F32_TOL, (shifted, thr)
    # bf16 storage
    xb = qkv[:2].to(torch.bfloat16).requires_grad_(True)
    HF.window_attention(xb, bits, None, nH, False).backward(g2[:2].to(torch.bfloat16))
    xr = qkv[:2, :4].to(torch.bfloat16).cpu().double().requires_grad_(True)
    _oracle_attn(xr, O.window_adjacency(nW), nH, False, None).backward(g2[:2, :4].to(torch.bfloat16).cpu().double())
    assert rel_err(xb.grad[:, :4].float().cpu(), xr.grad) < 1e-2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("d", [128, 256, 512, 1024])
def test_ln_bwd_xn_also_writes_the_layernorm_output(d, dtype):
    """hwgat_ln_bwd_xn = hwgat_ln_bwd(_masked) + xn = LN(x) (the plain operand of the weight-gradient launch of the Linear
    behind the LayerNorm, reference HWGATE.py:203 -> :86, :219 -> :131): dx / dgamma / dbeta / masked copy are what the
    other entry points write, xn is the LayerNorm forward."""
    g = torch.Generator().manual_seed(d)
    L = hw._lib
    dc = 0 if dtype == torch.float32 else 1
    n = 777
    x, dy, res = ((torch.randn(n, d, generator=g) * 1.5 + 0.3).to(DEV).to(dtype) for _ in range(3))
    w, b = (1 + 0.2 * torch.randn(d, generator=g)).to(DEV), (0.2 * torch.randn(d, generator=g)).to(DEV)
    mean, rstd = HF.ln_stats(x, w, b)
    dgr, dbr = torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
    dx0, dxm0 = HF.ln_backward(dy, x, mean, rstd, w, res, dgr, dbr, mask=(99, 0.1))
    for mask in (None, (99, 0.1)):
        dg1, db1 = torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
        out = HF.ln_backward(dy, x, mean, rstd, w, res, dg1, db1, mask=mask, beta=b)
        dx1, xn = out[0], out[-1]
        assert torch.equal(dx1, dx0)
        assert torch.allclose(dg1, dgr, rtol=1e-5, atol=1e-4) and torch.allclose(db1, dbr, rtol=1e-5, atol=1e-4)
        if mask is not None:
            assert torch.equal(out[1], dxm0)
        ref = O.layer_norm(x.cpu().double(), w.cpu().double(), b.cpu().double())
        assert rel_err(xn.float().cpu(), ref) < (F32_TOL if dtype == torch.float32 else BF16_TOL)
        y = torch.empty_like(x)
        L.call("hwgat_ln_fwd", L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y), L.ptr(mean), L.ptr(rstd), n, d, dc, L.stream())
        assert rel_err(xn.float().cpu(), y.float().cpu().double()) < (1e-6 if dtype == torch.float32 else 4e-3)
