"""One PartAttentionBlock (reference hwgat/models/HWGATE.py:189-221; likewise HGATE.py:175-213) as a single autograd
node whose forward and backward are sequences of HIP launches only.

forward  (4 GEMM launches + 1 attention; the reference runs ~40 ATen ops):
    stats1           = LN statistics of x: handed over by the PREVIOUS block's fc2 epilogue, or one hwgat_ln_fwd(y=NULL)
                       pass for the first block (and whenever the epilogue cannot produce them: bf16, ragged M)
    qkv              = LN1(x) Wqkv^T + b        LN folded into the weights and the GEMM epilogue (hwgat_ln_fold, pro 3);
                       ragged token counts: applied by the GEMM's loader (pro 1)
    o                = window attention(qkv)                           (hwgat_win_attn_fwd)
    y                = x + drop(o Wp^T + b)     bias+dropout+residual in the GEMM epilogue, which also accumulates the
                       row sums / sums of squares of y -> stats2 (hwgat_linear_nt_f32_ex + hwgat_ln_finalize)
    h1, u            = LN2(y) W1^T + b ; u = drop(gelu(h1))            GEMM epilogue; what is SAVED as h1 is gelu'(h1) * mask,
                       so the backward dX launch only multiplies by it
    out              = y + drop(u W2^T + b)     same epilogue: statistics of `out` for the next block's LN1, and at a
                       stage end the store goes straight to the TemporalMerging layout (HWGATE.py:55-63)
saved for backward: x, qkv, o, y, h1, u and the row statistics (10 E floats); LN outputs and
all dropout masks are recomputed (masks are a hash of (seed, element index)).

backward (4 dW/db GEMMs, 4 dX GEMMs, attention backward, 2 LN backward launches).
"""
import torch

from . import functional as HF

# Weight-gradient GEMMs (dW/db) are off the backward critical path: nothing downstream in the
# block consumes them.  When enabled they are issued on a side HIP stream, ordered by events after
# the kernels that produce their operands, so they fill the machine while the critical path runs
# its HBM-bound kernels (attention backward, LayerNorm backward).  Joined before backward returns.
# Measured on MI355X (config 2, fp32): 553.5 vs 552.4 clips/s -- no gain, both kinds of kernel use
# persistent full-chip grids; kept as an opt-in module switch (set block.OVERLAP_DW = True), off by default.
# Round 3 also gave the side stream its own CUs (hipExtStreamCreateWithCUMask: k CUs of every XCD, the main stream on all
# or on the rest): 32 / 64 / 96 CUs -> 206 / 328 / 396 clips/s against 627 (profiles/r03_cu_split_sweep.json, code at
# commit 183be5b): the step is throughput-bound on the matrix pipes, so a partition only lengthens the dW launches, which
# every block's backward joins before it returns.  Dropped.
OVERLAP_DW = False
_SIDE = {}


def _side_stream(device):
    key = (device.type, device.index)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device)
    return _SIDE[key]


class _DwQueue:
    """runs closures on the side stream after everything enqueued so far on the main stream"""

    def __init__(self, device, enabled):
        self.enabled = enabled and not HF.TIMERS_ACTIVE()
        if self.enabled:
            self.main = torch.cuda.current_stream(device)
            self.side = _side_stream(device)

    def run(self, fn):
        if not self.enabled:
            fn()
            return
        ev = torch.cuda.Event()
        ev.record(self.main)
        self.side.wait_event(ev)
        with torch.cuda.stream(self.side):
            fn()

    def join(self):
        if self.enabled:
            ev = torch.cuda.Event()
            ev.record(self.side)
            self.main.wait_event(ev)


class _FusedBlock(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, xc, thr, m1, r1, n1w, n1b, wqkv, bqkv, wp, bp, n2w, n2b, w1, b1, w2, b2, cfg):
        # xc: the "carrier" the producing block handed over with x (or None).  It has no data (a 1-element tensor
        # expanded to x's shape); its only purpose is that THIS block's backward can return, as its gradient, the masked
        # copy of dx that the producing block needs in front of its fc2 dropout (cfg `up` = that dropout's seed).
        bits, n_heads, shifted, p, seeds, kind, want_stats, merge_out, up, carry_out, book, deterministic, attn_p, prep, sb, _det_bwd = cfg
        ctx.set_materialize_grads(False)          # an unused carrier gradient arrives as None, not as a zero tensor
        B, F, K, d = x.shape
        dt = x.dtype                              # fp32, or bf16 activations with fp32 master weights
        cw = (lambda w: w) if dt == torch.float32 else (lambda w: w.to(dt))
        prep = prep or {}                         # derived weight copies made once per call by functional.WeightPrep
        if prep and prep["qkv_f"][0].dtype != dt:
            prep = {}
        fuse = HF.can_fuse_row_stats(x)           # the producing epilogue delivers the LayerNorm statistics
        if deterministic and fuse:
            # Bit-reproducible forward.  The 256-wide-tile kernels (N % 256 == 0) combine a tile's partial row sums in a
            # fixed order and add ONE (sum, sum of squares) per column tile -- and per frame of a merged row -- to the
            # row's global slot with fp32 atomics: two addends onto zero commute exactly, more do not.  The 128-wide-tile
            # kernels (N = 128) accumulate a tile's partials with LDS atomics in arrival order.  Eval takes the separate
            # statistics pass (and the merge pass) wherever the epilogue's result could depend on timing.
            tiles = d // 256
            if d % 256 or tiles > 2:
                fuse = False
            elif merge_out and 2 * tiles > 2:
                merge_out = False
        if m1 is None:
            m1, r1 = HF.ln_stats(x, n1w, n1b)
        qkv = HF.linear_nt_ln(x, wqkv, bqkv, (m1, r1, n1w, n1b), folded=prep.get("qkv_f"))
        o = torch.empty_like(x)
        HF.attn_fwd(kind, qkv, o, bits, thr, n_heads, shifted, (seeds[3], attn_p, sb) if attn_p > 0.0 else None)
        wp_c = prep["wp_c"] if "wp_c" in prep else cw(wp)
        w2_c = prep["w2_c"] if "w2_c" in prep else cw(w2)
        if fuse:
            y, m2, r2 = HF.linear_nt(o, wp_c, bp, epi=HF.EPI_BIAS_DROP_RES, res=x, epi_seed=seeds[0], epi_p=p, stats=True, seed_base=sb)
        else:
            y = HF.linear_nt(o, wp_c, bp, epi=HF.EPI_BIAS_DROP_RES, res=x, epi_seed=seeds[0], epi_p=p, seed_base=sb)
            m2, r2 = HF.ln_stats(y, n2w, n2b)
        # h1 here is gelu'(pre-activation) * dropout mask, the factor the backward multiplies by (EPI_BIAS_GELU_DROP_G)
        u, h1 = HF.linear_nt_ln(y, w1, b1, (m2, r2, n2w, n2b), epi=HF.EPI_BIAS_GELU_DROP_G, epi_seed=seeds[1], epi_p=p,
                                folded=prep.get("w1_f"), seed_base=sb)
        merged = bool(merge_out and fuse)
        if fuse and (want_stats or merged):
            out, mo, ro = HF.linear_nt(u, w2_c, b2, epi=HF.EPI_BIAS_DROP_RES, res=y, epi_seed=seeds[2], epi_p=p,
                                       stats=True, merge=(F, K) if merged else None, seed_base=sb)
        else:
            out = HF.linear_nt(u, w2_c, b2, epi=HF.EPI_BIAS_DROP_RES, res=y, epi_seed=seeds[2], epi_p=p, seed_base=sb)
            mo = ro = torch.empty(0, device=x.device)
        ctx.save_for_backward(x, thr, n1w, n1b, wqkv, wp, n2w, n2b, w1, w2, m1, r1, qkv, o, y, m2, r2, h1, u)
        ctx.cfg = cfg
        ctx.merged = merged
        wide = d >= HF.MASK_ONCE_MIN_D
        ctx.book = book
        ctx.send_up = xc is not None and up is not None and p > 0.0 and wide and book is not None
        carry = bool(carry_out) and not merged and p > 0.0 and HF.MASK_ONCE >= 2 and wide and book is not None
        oc = x.new_zeros(1).expand(out.shape) if carry else x.new_empty(0)
        if carry:
            ctx.mark_non_differentiable(mo, ro)
        else:
            ctx.mark_non_differentiable(mo, ro, oc)
        return out, mo, ro, oc

    @staticmethod
    def backward(ctx, dout, _dmo, _dro, doutm=None):
        x, thr, n1w, n1b, wqkv, wp, n2w, n2b, w1, w2, m1, r1, qkv, o, y, m2, r2, h1, u = ctx.saved_tensors
        bits, n_heads, shifted, p, seeds, kind = ctx.cfg[:6]
        up, attn_p, prep, sb = ctx.cfg[8], ctx.cfg[12], ctx.cfg[13] or {}, ctx.cfg[14]
        det = bool(ctx.cfg[15])                    # `deterministic_backward`: fixed-order parameter gradients (functional.linear_tn / ln_backward)
        if prep and ("w2T" not in prep or prep["w2T"].dtype != x.dtype):
            prep = {}

        def wT(name, w):                          # W^T in the activation dtype: from this call's WeightPrep, else made here
            return prep[name] if name in prep else HF.transpose(w, x.dtype)
        B, F, K, d = x.shape
        dt = x.dtype
        book = ctx.book
        if not (dout.is_contiguous() and book is not None and book.valid(dout, doutm)):
            doutm = None                          # no carrier, or dout is not (only) the dx the masked copy was made from
        dout = dout.contiguous()
        if ctx.merged:                            # the gradient arrives in the merged layout: back to (B, F, K, d)
            nat = torch.empty_like(x)
            if p > 0.0 and HF.MASK_ONCE >= 2 and d >= HF.MASK_ONCE_MIN_D:   # ... and once more multiplied by this block's fc2-dropout mask
                doutm = torch.empty_like(x)
                HF.call("hwgat_unmerge_masked", HF.ptr(dout), HF.ptr(nat), HF.ptr(doutm), B, F, K, d, HF.dtype_code(dout),
                        seeds[2] & 0xFFFFFFFF, float(p), HF.ptr(sb), HF.stream())
            else:
                doutm = None
                HF.call("hwgat_merge", HF.ptr(dout), HF.ptr(nat), B, F, K, d, 1, HF.dtype_code(dout), HF.stream())
            dout = nat
        # all 12 parameter gradients are accumulated into (split-M atomics, LN column sums): one flat
        # zero-filled buffer and views of it instead of 12 fill launches
        hid = w1.shape[0]
        sizes = [d, d, d, d, 3 * d * d, d * d, hid * d, d * hid, 3 * d, d, d, hid]
        flat = torch.zeros(sum(sizes), device=x.device, dtype=torch.float32)
        parts = torch.split(flat, sizes)
        dn1w, dn1b, dn2w, dn2b = parts[0], parts[1], parts[2], parts[3]
        dwqkv, dwp = parts[4].view(3 * d, d), parts[5].view(d, d)
        dw1, dw2 = parts[6].view(hid, d), parts[7].view(d, hid)
        dbqkv, dbp, db2, db1 = parts[8], parts[9], parts[10], parts[11]

        dwq = _DwQueue(x.device, OVERLAP_DW)
        # ---- FFN branch: out = y + drop3(u W2^T + b2), u = drop2(gelu(h1)), h1 = LN2(y) W1^T + b1
        if doutm is not None and doutm.shape == dout.shape:
            # the consumer of this block's output already wrote dropmask3 * dout (hwgat_ln_bwd_masked): no hashing here
            doutm = doutm.contiguous()
            dwq.run(lambda: HF.linear_tn(doutm, u, dw2, db2, deterministic=det))
            d_h1 = HF.linear_nt(doutm, wT("w2T", w2), None, epi=HF.EPI_MUL_AUX, aux=h1)
        else:
            dwq.run(lambda: HF.linear_tn(dout, u, dw2, db2, pro_seed=seeds[2], pro_p=p, seed_base=sb, deterministic=det))
            d_h1 = HF.linear_nt(dout, wT("w2T", w2), None, pro=HF.PRO_DROP, pro_seed=seeds[2], pro_p=p,
                                epi=HF.EPI_MUL_AUX, aux=h1, seed_base=sb)
        xn_path = HF.dw_wants_xn(x)               # LN(x) written by the LayerNorm backward for the dW launches (bf16, d % 256 == 0)
        if not xn_path:
            dwq.run(lambda: HF.linear_tn(d_h1, y, dw1, db1, ln=(m2, r2, n2w, n2b), deterministic=det))
        d_z = HF.linear_nt(d_h1, wT("w1T", w1), None, epi=HF.EPI_NONE)
        # ---- attention branch: y = x + drop1(o Wp^T + bp)
        if p > 0.0 and HF.MASK_ONCE >= 1 and d >= HF.MASK_ONCE_MIN_D:
            if xn_path:
                d_y, d_ym, yn = HF.ln_backward(d_z, y, m2, r2, n2w, dout, dn2w, dn2b, mask=(seeds[0], p), beta=n2b, seed_base=sb, deterministic=det)
                dwq.run(lambda: HF.linear_tn(d_h1, yn, dw1, db1, deterministic=det))
            else:
                d_y, d_ym = HF.ln_backward(d_z, y, m2, r2, n2w, dout, dn2w, dn2b, mask=(seeds[0], p), seed_base=sb, deterministic=det)   # + shortcut; and dropmask1 * d_y
            dwq.run(lambda: HF.linear_tn(d_ym, o, dwp, dbp, deterministic=det))
            d_o = HF.linear_nt(d_ym, wT("wpT", wp), None, epi=HF.EPI_NONE, out=d_z)
        else:
            if xn_path:
                d_y, yn = HF.ln_backward(d_z, y, m2, r2, n2w, dout, dn2w, dn2b, beta=n2b, deterministic=det)
                dwq.run(lambda: HF.linear_tn(d_h1, yn, dw1, db1, deterministic=det))
            else:
                d_y = HF.ln_backward(d_z, y, m2, r2, n2w, dout, dn2w, dn2b, deterministic=det)          # + shortcut gradient
            dwq.run(lambda: HF.linear_tn(d_y, o, dwp, dbp, pro_seed=seeds[0], pro_p=p, seed_base=sb, deterministic=det))
            d_o = HF.linear_nt(d_y, wT("wpT", wp), None, pro=HF.PRO_DROP, pro_seed=seeds[0], pro_p=p,
                               epi=HF.EPI_NONE, out=d_z, seed_base=sb)
        dqkv = torch.empty_like(qkv)
        HF.attn_bwd(kind, qkv, d_o, dqkv, bits, thr, n_heads, shifted, (seeds[3], attn_p, sb) if attn_p > 0.0 else None)
        if not xn_path:
            dwq.run(lambda: HF.linear_tn(dqkv, x, dwqkv, dbqkv, ln=(m1, r1, n1w, n1b), deterministic=det))
        d_xn = HF.linear_nt(dqkv, wT("wqkvT", wqkv), None, epi=HF.EPI_NONE, out=d_o)
        # (the first block's input comes from the parameter-free embedding: its dx is still produced because
        # dgamma / dbeta of norm1 fall out of the same LayerNorm-backward pass)
        dxm = None
        if ctx.send_up:           # the block that produced x gets dropmask3(its seed) * dx through the carrier's gradient
            if xn_path:
                dx, dxm, xn = HF.ln_backward(d_xn, x, m1, r1, n1w, d_y, dn1w, dn1b, mask=up, beta=n1b, seed_base=sb, deterministic=det)
            else:
                dx, dxm = HF.ln_backward(d_xn, x, m1, r1, n1w, d_y, dn1w, dn1b, mask=up, seed_base=sb, deterministic=det)
        elif xn_path:
            dx, xn = HF.ln_backward(d_xn, x, m1, r1, n1w, d_y, dn1w, dn1b, beta=n1b, deterministic=det)
        else:
            dx = HF.ln_backward(d_xn, x, m1, r1, n1w, d_y, dn1w, dn1b, deterministic=det)
        if xn_path:
            dwq.run(lambda: HF.linear_tn(dqkv, xn, dwqkv, dbqkv, deterministic=det))
        if dxm is not None:
            book.register(dx, dxm)
        dwq.join()        # every temporary above stays referenced until here, so the allocator cannot recycle it early
        return (dx, dxm, None, None, None, dn1w, dn1b, dwqkv, dbqkv, dwp, dbp, dn2w, dn2b, dw1, db1, dw2, db2, None)


def fused_block(x, thr, blk, bits, n_heads, shifted, p, seeds, kind="win", stats=None, want_stats=False,
                merge_out=False, return_stats=False, carrier=None, up=None, carry_out=False, return_carrier=None,
                book=None, deterministic=False, attn_p=0.0, prep=None, seed_base=None, deterministic_backward=False):
    """x (B,F,K,d) contiguous; `blk` holds norm1/attn.qkv/attn.proj/norm2/ff.fc1/ff.fc2.
    `kind`: 'win' = HWGATE part-window attention, 'blk' = HGATE block attention (thr must be None).
    `stats` = (mean, rstd) of the rows of x if the producer already has them; `want_stats`: have the fc2 epilogue produce
    the statistics of the output rows; `merge_out`: store the output in the TemporalMerging layout (B, F/2, K, 2d)
    (done only where the epilogue can: fp32, whole tiles -- check the returned shape).
    `carrier` / `up` / `carry_out` (training with dropout): the dropout mask of a block's fc2 output is applied to the
    incoming gradient ONCE, by the LayerNorm backward of the block that consumes that output, instead of in two GEMM
    loaders: `carry_out` makes this block return a data-less carrier next to `out`; the consumer passes it as `carrier`
    together with `up` = (this block's seeds[2], p) and returns the masked gradient as the carrier's gradient; both
    sides need the same `book` (functional.CarryBook of this forward call), without one no carrier is made or used.
    `attn_p`: attention dropout rate (reference HWGATE.py:78,112; 'win' only, needs `thr` and a fourth seed, seeds[3]).
    `prep`: this block's entry of the call's functional.WeightPrep (derived weight copies made by one launch per call).
    `seed_base`: 1-element device tensor added to every site seed when a kernel runs (functional.embed), or None.
    `deterministic_backward`: bit-reproducible parameter gradients (per-split / per-block partial images added in a fixed
    order instead of float atomics; token counts must be multiples of 32) -- with `deterministic` a whole train step repeats
    bit for bit, like the reference's on one device.
    `deterministic`: bit-reproducible forward (eval mode): statistics / merged store stay in the epilogue only where a
    row collects at most two atomic partials.
    Returns out, or (out, (mean, rstd) or None) with `return_stats`; with `carry_out` / `return_carrier` the carrier (or None) is appended."""
    m1, r1 = stats if stats is not None else (None, None)
    if carrier is not None and (carrier.shape != x.shape or not x.requires_grad or book is None):
        carrier = None
    out, mo, ro, oc = _FusedBlock.apply(
        x, carrier, thr, m1, r1, blk.norm1.weight, blk.norm1.bias, blk.attn.qkv.weight, blk.attn.qkv.bias,
        blk.attn.proj.weight, blk.attn.proj.bias, blk.norm2.weight, blk.norm2.bias,
        blk.ff.fc1.weight, blk.ff.fc1.bias, blk.ff.fc2.weight, blk.ff.fc2.bias,
        (bits, n_heads, shifted, float(p), tuple(int(s) for s in seeds), kind, bool(want_stats), bool(merge_out),
         (int(up[0]), float(up[1])) if (up is not None and carrier is not None) else None, bool(carry_out),
         book, bool(deterministic), float(attn_p), prep, seed_base, bool(deterministic_backward)))
    oc = oc if oc.numel() else None
    if return_carrier is None:
        return_carrier = bool(carry_out)
    if not return_stats:
        return (out, oc) if return_carrier else out
    st = (mo, ro) if mo.numel() else None
    return (out, st, oc) if return_carrier else (out, st)
