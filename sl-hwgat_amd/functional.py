"""torch.autograd bindings of the HWGAT HIP kernels (thin: pointers + sizes).

Every function here calls straight into libhwgat_hip.so through `_lib.call`;
nothing falls back to torch arithmetic.  All activations are in the natural
token order (B, F, K, d).
"""
import torch

from . import _lib
from ._lib import ptr, stream, dtype_code

# Optional live kernel timing (bench.py): when TIMERS is a dict, every launcher call
# is bracketed by HIP events recorded on the stream the kernel is launched on
# (torch's current stream) and the (start, stop) pairs are appended per entry point.
TIMERS = None
_EVENTS = []            # events created (and recorded once, which is what makes the driver allocate them) ahead of a timed region


def prime_events(n):
    """create `n` timing events now: hipEventCreate happens on an event's first record, ~15 us each -- inside a timed
    region that was 14 % of a 11 ms step (HGATE bf16), although the kernels themselves are untouched by it"""
    fresh = [torch.cuda.Event(enable_timing=True) for _ in range(max(0, n - len(_EVENTS)))]
    for e in fresh:
        e.record()
    torch.cuda.synchronize()
    _EVENTS.extend(fresh)


def call(name, *args):
    if TIMERS is None:
        return _lib.call(name, *args)
    e0 = _EVENTS.pop() if _EVENTS else torch.cuda.Event(enable_timing=True)
    e1 = _EVENTS.pop() if _EVENTS else torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.call(name, *args)
    e1.record()
    TIMERS.setdefault(name, []).append((e0, e1))


def TIMERS_ACTIVE():
    return TIMERS is not None


def timers_summary():
    """{entry point: (launches, total_ms)}; synchronises."""
    torch.cuda.synchronize()
    return {k: (len(v), sum(a.elapsed_time(b) for a, b in v)) for k, v in (TIMERS or {}).items()}


# ---------------------------------------------------------------- mask rows
def mask_bits(adj: torch.Tensor) -> torch.Tensor:
    """(nW,32,32) 0/1 adjacency (reference model_params.py:373-392) -> the
    (2,nW,32) uint32 rows `hwgat_win_attn_*` consume: [0] adjacency only,
    [1] adjacency AND the last-slot shift mask (no cross-frame attention,
    reference HWGATE.py:169-187)."""
    a = adj.detach().to("cpu", torch.float32)
    if a.dim() != 3 or a.shape[1:] != (32, 32):
        raise ValueError("adjacency must be (nW, 32, 32) (temporal_patch_size 2 x window 16)")
    if not bool(((a == 0) | (a == 1)).all()):
        raise ValueError("adjacency must be a 0/1 matrix")
    live = a != 0
    same_frame = torch.zeros(32, 32, dtype=torch.bool)
    same_frame[:16, :16] = True
    same_frame[16:, 16:] = True
    weights = (2 ** torch.arange(32, dtype=torch.int64))
    plain = (live.to(torch.int64) * weights).sum(-1)
    last = ((live & same_frame).to(torch.int64) * weights).sum(-1)
    bits = torch.stack([plain, last])                               # (2,nW,32) in [0, 2^32)
    bits = torch.where(bits >= 2 ** 31, bits - 2 ** 32, bits).to(torch.int32)
    return bits.contiguous()


def blk_mask_bits(adj: torch.Tensor, n_joints: int) -> torch.Tensor:
    """(2*KJ, 2*KJ) 0/1 block adjacency of HGATE (reference model_params.py:460-476) -> the (2,64,2)
    uint32 rows `hwgat_blk_attn_*` consume.  Query slot i = tp*32 + joint; word [s][i][kt] has bit j
    set iff key joint j of frame kt is visible: [0] adjacency only, [1] adjacency AND same frame (the
    last block of a shifted layer, reference HGATE.py:154-172).  Pad slots (joint >= KJ) stay 0."""
    KJ = int(n_joints)
    a = adj.detach().to("cpu", torch.float32)
    if not 1 <= KJ <= 32 or a.shape != (2 * KJ, 2 * KJ):
        raise ValueError("block adjacency must be (2*K, 2*K) with K <= 32 (temporal_patch_size 2)")
    if not bool(((a == 0) | (a == 1)).all()):
        raise ValueError("adjacency must be a 0/1 matrix")
    live = (a != 0).view(2, KJ, 2, KJ)                       # [tp_q][jq][tp_k][jk]
    weights = (2 ** torch.arange(KJ, dtype=torch.int64))
    plain = torch.zeros(2, 32, 2, dtype=torch.int64)
    plain[:, :KJ] = (live.to(torch.int64) * weights).sum(-1)
    last = plain.clone()
    last[0, :, 1] = 0                                        # no cross-frame pairs
    last[1, :, 0] = 0
    bits = torch.stack([plain.view(64, 2), last.view(64, 2)])
    bits = torch.where(bits >= 2 ** 31, bits - 2 ** 32, bits).to(torch.int32)
    return bits.contiguous()


def band_mask_rows(adj: torch.Tensor, frames: int) -> torch.Tensor:
    """(nW, T*16, T*16) 0/1 adjacency of WGATE (reference model_params.py:209-228) -> the (nW,16) int64
    rows `hwgat_band_attn_*` consume: bit 16*t + j of row [w][i] = key joint j of frame f-1+t visible to
    query joint i of frame f.  The kernel never forms the (T*16)^2 matrix, so the adjacency must have the
    structure the reference builds: block-tridiagonal over frames with the same three 16x16 blocks on
    every frame, and a visible key in every row of the diagonal block.  Anything else raises."""
    a = adj.detach().to("cpu")
    T = int(frames)
    if a.dim() != 3 or a.shape[1] != T * 16 or a.shape[2] != T * 16:
        raise ValueError("WGATE adjacency must be (nW, T*16, T*16)")
    nW = a.shape[0]
    if not bool(((a == 0) | (a == 1)).all()):
        raise ValueError("adjacency must be a 0/1 matrix")
    blk = (a != 0).view(nW, T, 16, T, 16).permute(0, 1, 3, 2, 4)          # [w][fq][fk][i][j]
    fq = torch.arange(T).view(T, 1)
    fk = torch.arange(T).view(1, T)
    off = (fk - fq)
    if bool(blk[:, off.abs() > 1].any()):
        raise NotImplementedError("WGATE HIP backend needs a block-tridiagonal adjacency (frames f-1, f, f+1)")
    rows = torch.zeros(nW, 16, dtype=torch.int64)
    weights = 2 ** torch.arange(16, dtype=torch.int64)
    for t, o in enumerate((-1, 0, 1)):
        sel = blk[:, off == o]                                            # (nW, n, 16, 16)
        if sel.shape[1] == 0:
            continue
        if not bool((sel == sel[:, :1]).all()):
            raise NotImplementedError("WGATE HIP backend needs the same adjacency blocks on every frame")
        rows += (sel[:, 0].to(torch.int64) * weights).sum(-1) << (16 * t)
    if not bool(((rows >> 16) & 0xFFFF).ne(0).all()):
        raise NotImplementedError("every query joint needs a visible key in its own frame")
    return rows.contiguous()


# ---------------------------------------------------------------- embedding
def embed(x, idx, bmat, pe, K, out_dtype=torch.float32, drop_p=0.0, seed=0, seed_base=None):
    """gather + Fourier features + PE (+ dropout) (no gradient: B is frozen, PE a buffer).
    `seed_base` (here and in every function below that takes a dropout seed): None, or a 1-element device tensor whose
    32-bit word the kernel adds to the site seed when it RUNS (include/hwgat_hip.h, "dropout seeds")."""
    B, T, J, C = x.shape
    d0 = bmat.shape[0] * 2
    out = torch.empty(B, T, K, d0, device=x.device, dtype=out_dtype)
    call("hwgat_embed_fwd", ptr(x), ptr(idx), ptr(bmat), ptr(pe), ptr(out),
         B, T, J, K, C, d0, dtype_code(out), seed & 0xFFFFFFFF, float(drop_p), ptr(seed_base), stream())
    return out


# ---------------------------------------------------------------- LayerNorm
class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta):
        d = x.shape[-1]
        n = x.numel() // d
        y = torch.empty_like(x)
        mean = torch.empty(n, device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        call("hwgat_ln_fwd", ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(rstd),
             n, d, dtype_code(x), stream())
        ctx.save_for_backward(x, gamma, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, rstd = ctx.saved_tensors
        d = x.shape[-1]
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dg = torch.zeros(2, d, device=x.device, dtype=torch.float32)
        call("hwgat_ln_bwd", ptr(dy), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), None, ptr(dx),
             ptr(dg[0]), ptr(dg[1]), x.numel() // d, d, dtype_code(x), stream())
        return dx, dg[0], dg[1]


def layer_norm(x, gamma, beta):
    return _LayerNorm.apply(x.contiguous(), gamma, beta)


# ---------------------------------------------------------------- attention
class _WinAttn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, bits, thr, n_heads, shifted, drop):
        B, F, K, d3 = qkv.shape
        d = d3 // 3
        o = torch.empty(B, F, K, d, device=qkv.device, dtype=qkv.dtype)
        attn_fwd("win", qkv, o, bits, thr, n_heads, shifted, drop)
        ctx.save_for_backward(qkv, bits, thr)
        ctx.cfg = (n_heads, int(shifted), drop)
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, bits, thr = ctx.saved_tensors
        n_heads, shifted, drop = ctx.cfg
        do = do.contiguous()
        dqkv = torch.empty_like(qkv)
        attn_bwd("win", qkv, do, dqkv, bits, thr, n_heads, shifted, drop)
        return dqkv, None, None, None, None, None


def _attn_drop(kind, thr, drop):
    """(seed, p, seed_base) of the attention dropout (reference HWGATE.py:78,112, HGATE.py:78,106, WGATE.py:81,103) or None"""
    if drop is None or float(drop[1]) <= 0.0:
        return None
    if kind == "win" and thr is None:
        raise ValueError("attention dropout is a train-mode operation: it needs the train-mode threshold tensor")
    return int(drop[0]) & 0xFFFFFFFF, float(drop[1]), (drop[2] if len(drop) > 2 else None)


def attn_fwd(kind, qkv, o, bits, thr, n_heads, shifted, drop=None):
    """launch the attention forward of a model family: 'win' = HWGATE part windows, 'blk' = HGATE blocks, 'band' = WGATE.
    `drop` = (seed, p) or (seed, p, seed_base): attention dropout ('win', train mode only)"""
    B, F, K, d = o.shape
    drop = _attn_drop(kind, thr, drop)
    if kind == "win" and drop is not None:
        call("hwgat_win_attn_fwd_drop", ptr(qkv), ptr(o), ptr(bits), ptr(thr), B, F, K // 16, n_heads, d // n_heads,
             int(shifted), dtype_code(qkv), drop[0], drop[1], ptr(drop[2]), stream())
    elif kind == "win":
        call("hwgat_win_attn_fwd", ptr(qkv), ptr(o), ptr(bits), ptr(thr), B, F, K // 16, n_heads, d // n_heads,
             int(shifted), dtype_code(qkv), stream())
    elif kind == "blk":
        assert thr is None, "HGATE has no train-mode threshold"
        if drop is not None:
            call("hwgat_blk_attn_fwd_drop", ptr(qkv), ptr(o), ptr(bits), B, F, K, n_heads, d // n_heads, int(shifted),
                 dtype_code(qkv), drop[0], drop[1], ptr(drop[2]), stream())
        else:
            call("hwgat_blk_attn_fwd", ptr(qkv), ptr(o), ptr(bits), B, F, K, n_heads, d // n_heads, int(shifted),
                 dtype_code(qkv), stream())
    elif kind == "band":
        assert thr is None and not shifted, "WGATE has neither threshold nor shift"
        if drop is not None:
            call("hwgat_band_attn_fwd_drop", ptr(qkv), ptr(o), ptr(bits), B, F, K // 16, n_heads, d // n_heads,
                 dtype_code(qkv), drop[0], drop[1], ptr(drop[2]), stream())
        else:
            call("hwgat_band_attn_fwd", ptr(qkv), ptr(o), ptr(bits), B, F, K // 16, n_heads, d // n_heads,
                 dtype_code(qkv), stream())
    else:
        raise ValueError(kind)


def attn_bwd(kind, qkv, do, dqkv, bits, thr, n_heads, shifted, drop=None):
    B, F, K, d = do.shape
    drop = _attn_drop(kind, thr, drop)
    if kind == "win" and drop is not None:
        call("hwgat_win_attn_bwd_drop", ptr(qkv), ptr(do), ptr(dqkv), ptr(bits), ptr(thr), B, F, K // 16, n_heads,
             d // n_heads, int(shifted), dtype_code(qkv), drop[0], drop[1], ptr(drop[2]), stream())
    elif kind == "win":
        call("hwgat_win_attn_bwd", ptr(qkv), ptr(do), ptr(dqkv), ptr(bits), ptr(thr), B, F, K // 16, n_heads,
             d // n_heads, int(shifted), dtype_code(qkv), stream())
    elif kind == "blk":
        if drop is not None:
            call("hwgat_blk_attn_bwd_drop", ptr(qkv), ptr(do), ptr(dqkv), ptr(bits), B, F, K, n_heads, d // n_heads,
                 int(shifted), dtype_code(qkv), drop[0], drop[1], ptr(drop[2]), stream())
        else:
            call("hwgat_blk_attn_bwd", ptr(qkv), ptr(do), ptr(dqkv), ptr(bits), B, F, K, n_heads, d // n_heads,
                 int(shifted), dtype_code(qkv), stream())
    elif kind == "band":
        if drop is not None:
            call("hwgat_band_attn_bwd_drop", ptr(qkv), ptr(do), ptr(dqkv), ptr(bits), B, F, K // 16, n_heads, d // n_heads,
                 dtype_code(qkv), drop[0], drop[1], ptr(drop[2]), stream())
        else:
            call("hwgat_band_attn_bwd", ptr(qkv), ptr(do), ptr(dqkv), ptr(bits), B, F, K // 16, n_heads, d // n_heads,
                 dtype_code(qkv), stream())
    else:
        raise ValueError(kind)


class _BlkAttn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, bits, n_heads, shifted, drop):
        B, F, K, d3 = qkv.shape
        o = torch.empty(B, F, K, d3 // 3, device=qkv.device, dtype=qkv.dtype)
        attn_fwd("blk", qkv, o, bits, None, n_heads, shifted, drop)
        ctx.save_for_backward(qkv, bits)
        ctx.cfg = (n_heads, int(shifted), drop)
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, bits = ctx.saved_tensors
        n_heads, shifted, drop = ctx.cfg
        do = do.contiguous()
        dqkv = torch.empty_like(qkv)
        attn_bwd("blk", qkv, do, dqkv, bits, None, n_heads, shifted, drop)
        return dqkv, None, None, None, None


class _BandAttn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, rows, n_heads, drop):
        B, F, K, d3 = qkv.shape
        o = torch.empty(B, F, K, d3 // 3, device=qkv.device, dtype=qkv.dtype)
        attn_fwd("band", qkv, o, rows, None, n_heads, False, drop)
        ctx.save_for_backward(qkv, rows)
        ctx.cfg = (n_heads, drop)
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, rows = ctx.saved_tensors
        n_heads, drop = ctx.cfg
        do = do.contiguous()
        dqkv = torch.empty_like(qkv)
        attn_bwd("band", qkv, do, dqkv, rows, None, n_heads, False, drop)
        return dqkv, None, None, None


def band_attention(qkv, rows, n_heads, drop=None):
    """WGATE: qkv (B,F,K,3d) -> o (B,F,K,d); a window = one 16-joint part window over all F frames.
    `drop` = (seed, p[, seed_base]): attention dropout (reference WGATE.py:103)."""
    return _BandAttn.apply(qkv.contiguous(), rows, n_heads, _attn_drop("band", None, drop))


def block_attention(qkv, bits, n_heads, shifted, drop=None):
    """HGATE: qkv (B,F,K,3d) -> o (B,F,K,d); a block = 2 frames x all K joints.
    `drop` = (seed, p[, seed_base]): attention dropout (reference HGATE.py:106)."""
    return _BlkAttn.apply(qkv.contiguous(), bits, n_heads, shifted, _attn_drop("blk", None, drop))


def window_attention(qkv, bits, thr, n_heads, shifted, drop=None):
    """qkv (B,F,K,3d) -> o (B,F,K,d).  `thr`: 1-element fp32 device tensor
    (train mode) or None (eval mode).  `drop` = (seed, p): attention dropout on the
    probabilities (reference HWGATE.py:112), train mode only."""
    drop = _attn_drop("win", thr, drop)
    return _WinAttn.apply(qkv.contiguous(), bits, thr, n_heads, shifted, drop)


# ---------------------------------------------------------------- merge
class _Merge(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        B, F, K, d = x.shape
        out = torch.empty(B, F // 2, K, 2 * d, device=x.device, dtype=x.dtype)
        call("hwgat_merge", ptr(x), ptr(out), B, F, K, d, 0, dtype_code(x), stream())
        return out

    @staticmethod
    def backward(ctx, dout):
        B, f, K, d2 = dout.shape
        dout = dout.contiguous()
        dx = torch.empty(B, f * 2, K, d2 // 2, device=dout.device, dtype=dout.dtype)
        call("hwgat_merge", ptr(dout), ptr(dx), B, f * 2, K, d2 // 2, 1, dtype_code(dout), stream())
        return dx


def temporal_merge(x):
    return _Merge.apply(x.contiguous())


# ---------------------------------------------------------------- LN + pool
class _LnPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, xc=None, up=None, book=None, deterministic=False, seed_base=None):
        # xc / up / book: carrier of the masked gradient for the block that produced x (block.fused_block)
        ctx.up = up if (xc is not None and up is not None and up[1] > 0.0 and book is not None) else None
        ctx.book = book
        ctx.seed_base = seed_base
        B, d = x.shape[0], x.shape[-1]
        n_tok = x.numel() // (B * d)
        mean = torch.empty(B * n_tok, device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        if deterministic:      # per-block partial sums added in index order: the same bits on every run
            rows = _lib.lib().hwgat_lnpool_partial_rows(B, n_tok)
            hat = torch.empty(B, d, device=x.device, dtype=torch.float32)
            part = torch.empty(B * rows, d, device=x.device, dtype=torch.float32)
            call("hwgat_lnpool_fwd_det", ptr(x), ptr(hat), ptr(mean), ptr(rstd), B, n_tok, d, dtype_code(x), ptr(part), stream())
        else:
            hat = torch.zeros(B, d, device=x.device, dtype=torch.float32)
            call("hwgat_lnpool_fwd", ptr(x), ptr(hat), ptr(mean), ptr(rstd), B, n_tok, d, dtype_code(x), stream())
        hat_mean = hat / n_tok
        ctx.save_for_backward(x, gamma, mean, rstd, hat_mean)
        return hat_mean * gamma + beta

    @staticmethod
    def backward(ctx, dfeat):
        x, gamma, mean, rstd, hat_mean = ctx.saved_tensors
        B, d = x.shape[0], x.shape[-1]
        n_tok = x.numel() // (B * d)
        dfeat = dfeat.float()
        g = (dfeat * gamma / n_tok).contiguous()
        dx = torch.empty_like(x)
        dxm = torch.empty_like(x) if ctx.up is not None else None
        call("hwgat_lnpool_bwd_masked", ptr(g), ptr(x), ptr(mean), ptr(rstd), ptr(dx), B, n_tok, d,
             dtype_code(x), ptr(dxm), (ctx.up[0] if ctx.up else 0) & 0xFFFFFFFF, float(ctx.up[1]) if ctx.up else 0.0,
             ptr(ctx.seed_base), stream())
        if dxm is not None:
            ctx.book.register(dx, dxm)
        return dx, (dfeat * hat_mean).sum(0), dfeat.sum(0), dxm, None, None, None, None


def ln_mean_pool(x, gamma, beta, carrier=None, up=None, book=None, deterministic=False, seed_base=None):
    """final LayerNorm + mean over all tokens -> (B, d) fp32.  carrier / up / book: see block.fused_block (the last
    block's fc2-dropout mask is applied to its incoming gradient here, once).  `deterministic`: fixed summation order
    (two launches, no atomics), what eval() uses so that two forwards are bit-identical like the reference's."""
    xcont = x.contiguous()
    if carrier is not None and (xcont is not x or carrier.shape != x.shape or book is None):
        carrier = None
    return _LnPool.apply(xcont, gamma, beta, carrier, up, book, bool(deterministic), seed_base)


# ---------------------------------------------------------------- fp32 MFMA linears
PRO_NONE, PRO_LN, PRO_DROP, PRO_LN_FOLD = 0, 1, 2, 3
# Formulation switches.  Plain module constants: the product reads no environment variables; the parity tests flip them
# (monkeypatch) to compare the formulations against each other.
# LayerNorm -> Linear pairs (norm1 -> qkv, norm2 -> fc1) run with the normalisation folded into the weights and the
# GEMM epilogue (hwgat_ln_fold + pro 3) when the token count is whole tiles; False keeps the normalising loader (pro 1).
LN_FOLD = True
# Dropout masks in the backward pass: 0 = hashed in every GEMM loader that needs the masked gradient; 1 = the
# LayerNorm backward that produces a gradient also writes its masked copy once (hwgat_ln_bwd_masked) for the block's own
# projection dropout; 2 = also across blocks, for the fc2 dropout of the block that produced this block's input.
MASK_ONCE = 2
# ... for blocks at least this wide: at d = 128 the extra E-sized write costs what the mask loaders cost there
# (measured: WGATE, 8 blocks of d = 128, 1 468 -> 1 441 clips/s with masked copies everywhere)
MASK_ONCE_MIN_D = 256


class CarryBook:
    """Validity records of the dropout-masked gradient copies of ONE forward call.

    A masked copy handed to the producing block through a carrier is only valid if the gradient that block receives IS
    the dx it was made from.  If the tensor has another consumer (an auxiliary loss on a block output, say) autograd adds
    that gradient to dx -- in a new tensor, or in place -- and the copy would miss it.  The consumer therefore registers
    (address, version counter, shape) of its dx under the address of the masked copy; the producer uses the copy only if
    the gradient it got still has exactly that address and version, and otherwise masks the real gradient in its
    loaders.  One book per forward call (held by the HandOver, referenced by the autograd nodes of that call): nothing is
    shared between models or between two forwards of one model."""

    def __init__(self):
        self._rec = {}

    def register(self, dx, dxm):
        if dxm is not None:
            self._rec[dxm.data_ptr()] = (dx.data_ptr(), dx._version, tuple(dx.shape))

    def valid(self, dout, doutm):
        """True iff `doutm` is the registered masked copy of exactly this `dout`"""
        if doutm is None:
            return False
        rec = self._rec.pop(doutm.data_ptr(), None)
        return rec is not None and rec == (dout.data_ptr(), dout._version, tuple(dout.shape))


class WeightPrep:
    """Every derived copy of the block weights a forward (+ backward) call consumes, made by ONE launch
    (hwgat_weight_prep): per block the LayerNorm-folded qkv / fc1 weights with their row sums (hwgat_ln_fold), the
    activation-dtype copies of proj / fc2 (bf16 activations only) and, when a backward will follow, the four transposed
    copies for the dX launches.  The output buffers and the device table are built once per (model, dtype, train) and
    reused: the copies are rewritten by every forward call from the current master weights, and a backward reads the
    copies of its own forward (the weights do not change in between).  `per_block[k]` maps names to tensors."""

    OP_COPY, OP_T, OP_FOLD = 0, 1, 2

    def __init__(self, blocks, dtype, with_transposes):
        import numpy as np
        self.dtype = dtype
        dev = blocks[0].attn.qkv.weight.device
        ent, self.per_block, self._keep = [], [], []
        first = 0

        def add(op, W, out, bias=None, gamma=None, beta=None, s=None, c=None):
            nonlocal first
            N, K = W.shape
            ent.append((W.data_ptr(), 0 if bias is None else bias.data_ptr(), 0 if gamma is None else gamma.data_ptr(),
                        0 if beta is None else beta.data_ptr(), out.data_ptr(), 0 if s is None else s.data_ptr(),
                        0 if c is None else c.data_ptr(), N, K, op, first))
            first += -(-N // 4) if op == self.OP_FOLD else (-(-N // 32)) * (-(-K // 32))

        def fold(lin, norm):
            N, K = lin.weight.shape
            Wf = torch.empty(N, K, device=dev, dtype=dtype)
            sc = torch.empty(2, N, device=dev, dtype=torch.float32)
            add(self.OP_FOLD, lin.weight, Wf, lin.bias, norm.weight, norm.bias, sc[0], sc[1])
            return Wf, sc[0], sc[1]

        def copy(W, transposed):
            N, K = W.shape
            out = torch.empty((K, N) if transposed else (N, K), device=dev, dtype=dtype)
            add(self.OP_T if transposed else self.OP_COPY, W, out)
            return out

        srcs = []
        for blk in blocks:
            lins = (blk.attn.qkv, blk.attn.proj, blk.ff.fc1, blk.ff.fc2)
            srcs += [t for lin in lins for t in (lin.weight, lin.bias)] + [blk.norm1.weight, blk.norm1.bias, blk.norm2.weight, blk.norm2.bias]
            d = {"qkv_f": fold(blk.attn.qkv, blk.norm1), "w1_f": fold(blk.ff.fc1, blk.norm2)}
            if dtype != torch.float32:
                d["wp_c"], d["w2_c"] = copy(blk.attn.proj.weight, False), copy(blk.ff.fc2.weight, False)
            if with_transposes:
                d["wqkvT"], d["wpT"] = copy(blk.attn.qkv.weight, True), copy(blk.attn.proj.weight, True)
                d["w1T"], d["w2T"] = copy(blk.ff.fc1.weight, True), copy(blk.ff.fc2.weight, True)
            self.per_block.append(d)
        self.key = self.signature(blocks, dtype, with_transposes)
        rec = np.dtype([("p", np.uint64, 7), ("i", np.int32, 4)])           # hwgat_prep_entry: 7 pointers, N, K, op, first_block
        assert rec.itemsize == 72
        host = np.zeros(len(ent), dtype=rec)
        for i, e in enumerate(ent):
            host[i]["p"] = e[:7]
            host[i]["i"] = e[7:]
        self.table = torch.from_numpy(host.view(np.uint8).copy()).to(dev)
        self.n, self.total = len(ent), first

    @staticmethod
    def signature(blocks, dtype, with_transposes):
        """what the cached buffers and table were built for: the addresses of every source parameter"""
        ptrs = []
        for blk in blocks:
            for t in (blk.attn.qkv.weight, blk.attn.qkv.bias, blk.attn.proj.weight, blk.ff.fc1.weight, blk.ff.fc1.bias,
                      blk.ff.fc2.weight, blk.norm1.weight, blk.norm1.bias, blk.norm2.weight, blk.norm2.bias):
                ptrs.append(t.data_ptr())
        return (dtype, bool(with_transposes), tuple(ptrs))

    def run(self):
        call("hwgat_weight_prep", ptr(self.table), self.n, self.total, 0 if self.dtype == torch.float32 else 1, stream())
        return self


def weight_prep(owner, blocks, dtype, with_transposes):
    """the (cached) WeightPrep of `owner` (a model) for this dtype / mode, run for the current weights; None where the
    masters are not plain fp32 parameters (the per-call kernels then do the work, as before)"""
    ws = [blk.attn.qkv.weight for blk in blocks]
    if not ws or any(w.dtype != torch.float32 or not w.is_cuda for w in ws):
        return None
    key = WeightPrep.signature(blocks, dtype, with_transposes)
    cache = owner.__dict__.setdefault("_weight_prep_cache", {})
    wp = cache.get(key[:2])
    if wp is None or wp.key != key:
        wp = cache[key[:2]] = WeightPrep(blocks, dtype, with_transposes)
    return wp.run()


class HandOver:
    """What one block's epilogues produced for the next block of the same forward call: `of` = the tensor the values
    belong to, `stats` = (mean, rstd) of its rows (from the fc2 epilogue), `carrier` / `up` = the data-less carrier of
    the masked gradient and the (seed, p) of the dropout it masks, `plan[k]` = (produce output statistics, store merged)
    for block k, `book` = the CarryBook of the call.  A local of Model.forward_features; never stored on the module."""

    def __init__(self, last_block=-1, deterministic=False):
        self.of = self.stats = self.carrier = self.up = None
        self.plan = {}
        self.last_block = last_block
        self.deterministic = bool(deterministic)
        self.book = CarryBook()
        self.prep = None            # WeightPrep of this call (per_block[k] = the derived weight copies of block k)
        self.seed_base = None       # 1-element device tensor: the base seed of this call's dropout masks (or None)


EPI_BIAS, EPI_BIAS_DROP_RES, EPI_BIAS_GELU_DROP, EPI_GELU_BWD, EPI_NONE, EPI_BIAS_GELU_DROP_G, EPI_MUL_AUX = 0, 1, 2, 3, 4, 5, 6


def can_fuse_row_stats(x):
    """the producing linear's epilogue can deliver the LayerNorm statistics of `x`-shaped output (whole tiles)"""
    return (x.numel() // x.shape[-1]) % 256 == 0


def linear_nt(A, W, bias=None, *, pro=PRO_NONE, ln=None, pro_seed=0, pro_p=0.0, epi=EPI_BIAS,
              res=None, aux=None, epi_seed=0, epi_p=0.0, out=None, stats=False, merge=None, seed_base=None):
    """C[M,N] = pro(A)[M,K] . W[N,K]^T with fused epilogue (see include/hwgat_hip.h).
    Returns C, or (C, C2) for EPI_BIAS_GELU_DROP (C2 = pre-activation) / EPI_BIAS_GELU_DROP_G (C2 = gelu' * mask).
    EPI_BIAS_DROP_RES: `stats=True` also returns (mean, rstd) of the OUTPUT rows, produced by the epilogue
    (no separate pass over C); `merge=(F, K_tok)` stores C in the TemporalMerging layout (B, F/2, K_tok, 2N)
    (HWGATE.py:55-63), statistics then per merged row.  Returns (C, mean, rstd)."""
    K = A.shape[-1]
    M = A.numel() // K
    N = W.shape[0]
    if W.dtype != A.dtype:
        raise TypeError("weight must already be in the activation dtype (cast once per step)")
    if stats or merge is not None:
        if epi != EPI_BIAS_DROP_RES or pro != PRO_NONE or M % 256 or out is not None:
            raise ValueError("row statistics / merged store: EPI_BIAS_DROP_RES, no prologue, M % 256 == 0")
        if merge is not None:
            F, Kt = merge
            C = torch.empty(M // (F * Kt), F // 2, Kt, 2 * N, device=A.device, dtype=A.dtype)
            rows, width = M // 2, 2 * N
        else:
            C = torch.empty(*A.shape[:-1], N, device=A.device, dtype=A.dtype)
            rows, width = M, N
        st = torch.zeros(2, rows, device=A.device, dtype=torch.float32)
        call("hwgat_linear_nt_f32_ex" if A.dtype == torch.float32 else "hwgat_linear_nt_bf16_ex",
             ptr(A), ptr(W), ptr(bias), ptr(C), M, N, K, pro, None, None, None, None,
             pro_seed & 0xFFFFFFFF, float(pro_p), epi, ptr(res), None, None, epi_seed & 0xFFFFFFFF, float(epi_p),
             ptr(st[0]), ptr(st[1]), merge[0] if merge else 0, merge[1] if merge else 0, ptr(seed_base), stream())
        call("hwgat_ln_finalize", ptr(st[0]), ptr(st[1]), rows, width, stream())
        return C, st[0], st[1]
    C = out if out is not None else torch.empty(*A.shape[:-1], N, device=A.device, dtype=A.dtype)
    C2 = torch.empty_like(C) if epi in (EPI_BIAS_GELU_DROP, EPI_BIAS_GELU_DROP_G) else None
    mean = rstd = gamma = beta = None
    if pro in (PRO_LN, PRO_LN_FOLD):
        mean, rstd, gamma, beta = ln
    call("hwgat_linear_nt_f32" if A.dtype == torch.float32 else "hwgat_linear_nt_bf16", ptr(A), ptr(W), ptr(bias), ptr(C), M, N, K, pro, ptr(mean), ptr(rstd),
         ptr(gamma), ptr(beta), pro_seed & 0xFFFFFFFF, float(pro_p), epi, ptr(res), ptr(C2), ptr(aux),
         epi_seed & 0xFFFFFFFF, float(epi_p), ptr(seed_base), stream())
    return (C, C2) if C2 is not None else C


def ln_fold(W, bias, gamma, beta, dtype):
    """(W o gamma in `dtype`, s, c) for pro = PRO_LN_FOLD from the fp32 master weights (hwgat_ln_fold)"""
    N, K = W.shape
    Wf = torch.empty(N, K, device=W.device, dtype=dtype)
    sc = torch.empty(2, N, device=W.device, dtype=torch.float32)
    call("hwgat_ln_fold", ptr(W), ptr(bias), ptr(gamma), ptr(beta), N, K, ptr(Wf), ptr(sc[0]), ptr(sc[1]),
         0 if dtype == torch.float32 else 1, stream())
    return Wf, sc[0], sc[1]


def linear_nt_ln(A, W, bias, ln, *, epi=EPI_BIAS, epi_seed=0, epi_p=0.0, out=None, folded=None, seed_base=None):
    """LN(A) . W^T + bias with the epilogue `epi` (EPI_BIAS or EPI_BIAS_GELU_DROP); W, bias are the fp32 master
    parameters, ln = (mean, rstd, gamma, beta).  Whole-tile token counts take the folded form (no per-element
    normalisation in the GEMM's load path), anything else the normalising loader."""
    mean, rstd, gamma, beta = ln
    M = A.numel() // A.shape[-1]
    if LN_FOLD and M % 128 == 0 and W.dtype == torch.float32 and gamma.dtype == torch.float32:   # hwgat_ln_fold reads fp32 masters
        Wf, s, c = folded if folded is not None else ln_fold(W, bias, gamma, beta, A.dtype)   # `folded`: made by WeightPrep
        return linear_nt(A, Wf, None, pro=PRO_LN_FOLD, ln=(mean, rstd, s, c), epi=epi, epi_seed=epi_seed, epi_p=epi_p, out=out,
                         seed_base=seed_base)
    Wc = W if W.dtype == A.dtype else W.to(A.dtype)
    return linear_nt(A, Wc, bias, pro=PRO_LN, ln=ln, epi=epi, epi_seed=epi_seed, epi_p=epi_p, out=out, seed_base=seed_base)


def linear_tn(A, Bm, dW, db=None, *, pro_seed=0, pro_p=0.0, ln=None, seed_base=None, deterministic=False):
    """dW[N,K] += dropmask(A)[M,N]^T . ln(Bm)[M,K]; db[N] += colsum(dropmask(A)).
    ln = (mean, rstd, gamma, beta) normalises Bm on the fly.
    `deterministic`: the bit-reproducible form (hwgat_linear_tn_*_det: per-split partial images in a zero-filled
    workspace, added in split order; no float atomics)."""
    N, K = dW.shape
    M = A.numel() // N
    if deterministic:
        need = _lib.lib().hwgat_linear_tn_det_bytes(M, N, K)
        if M % 32 or need <= 0:
            raise NotImplementedError("deterministic weight gradients need a token count that is a multiple of 32 and "
                                      "N, K multiples of 128")
        mean, rstd, gamma, beta = ln if ln is not None else (None, None, None, None)
        ws = torch.zeros(need // 4, device=A.device, dtype=torch.float32)
        call("hwgat_linear_tn_f32_det" if A.dtype == torch.float32 else "hwgat_linear_tn_bf16_det", ptr(A), ptr(Bm), ptr(dW),
             ptr(db), M, N, K, pro_seed & 0xFFFFFFFF, float(pro_p), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(seed_base),
             ptr(ws), need, stream())
        return
    if A.dtype == torch.bfloat16 and ln is None and pro_p == 0.0:
        # plain bf16 operands: M-split partial tiles through a workspace + fixed-order reduction instead of global atomics
        need = _lib.lib().hwgat_linear_tn_bf16_ws_bytes(M, N, K)
        if need > 0:
            ws = torch.empty(need // 4, device=A.device, dtype=torch.float32)
            call("hwgat_linear_tn_bf16_ws", ptr(A), ptr(Bm), ptr(dW), ptr(db), M, N, K, ptr(ws), need, stream())
            return
    mean, rstd, gamma, beta = ln if ln is not None else (None, None, None, None)
    if A.dtype == torch.float32:
        # fp32, 256-aligned multi-tile outputs: slabs + fixed-order reduction as well (any prologue)
        need = _lib.lib().hwgat_linear_tn_f32_ws_bytes(M, N, K)
        if need > 0:
            ws = torch.empty(need // 4, device=A.device, dtype=torch.float32)
            call("hwgat_linear_tn_f32_ws", ptr(A), ptr(Bm), ptr(dW), ptr(db), M, N, K, pro_seed & 0xFFFFFFFF, float(pro_p),
                 ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(ws), need, ptr(seed_base), stream())
            return
    call("hwgat_linear_tn_f32" if A.dtype == torch.float32 else "hwgat_linear_tn_bf16", ptr(A), ptr(Bm), ptr(dW), ptr(db), M, N, K, pro_seed & 0xFFFFFFFF,
         float(pro_p), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(seed_base), stream())


def ln_stats(x, gamma, beta):
    """per-row mean / rstd only (consumers normalise on the fly)"""
    d = x.shape[-1]
    n = x.numel() // d
    mean = torch.empty(n, device=x.device, dtype=torch.float32)
    rstd = torch.empty_like(mean)
    call("hwgat_ln_fwd", ptr(x), ptr(gamma), ptr(beta), None, ptr(mean), ptr(rstd), n, d, dtype_code(x), stream())
    return mean, rstd


def ln_backward(dy, x, mean, rstd, gamma, dres, dgamma, dbeta, mask=None, beta=None, seed_base=None, deterministic=False):
    """dx = dLN(dy) (+ dres); dgamma/dbeta accumulated in place.  mask = (seed, p): also returns dx * dropout-mask
    (the gradient in front of the dropout that produced this tensor) -> (dx, dx_masked).  `beta` given: ALSO returns
    xn = LN(x) (appended), for the weight-gradient launch of the Linear behind this LayerNorm (hwgat_ln_bwd_xn).
    `deterministic`: dgamma / dbeta through per-block images added in a fixed order (hwgat_ln_bwd_det)."""
    d = x.shape[-1]
    dx = torch.empty_like(x)
    if deterministic:
        dxm = torch.empty_like(x) if mask is not None else None
        xn = torch.empty_like(x) if beta is not None else None
        need = _lib.lib().hwgat_ln_bwd_det_bytes(d)
        ws = torch.empty(need // 4, device=x.device, dtype=torch.float32)
        call("hwgat_ln_bwd_det", ptr(dy), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(dres), ptr(dx), ptr(dgamma),
             ptr(dbeta), x.numel() // d, d, dtype_code(x), ptr(dxm), (mask[0] if mask else 0) & 0xFFFFFFFF,
             float(mask[1]) if mask else 0.0, ptr(xn), ptr(seed_base), ptr(ws), need, stream())
        out = (dx,) + ((dxm,) if mask is not None else ()) + ((xn,) if beta is not None else ())
        return out if len(out) > 1 else dx
    if beta is not None:
        dxm = torch.empty_like(x) if mask is not None else None
        xn = torch.empty_like(x)
        call("hwgat_ln_bwd_xn", ptr(dy), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(dres), ptr(dx),
             ptr(dgamma), ptr(dbeta), x.numel() // d, d, dtype_code(x), ptr(dxm), (mask[0] if mask else 0) & 0xFFFFFFFF,
             float(mask[1]) if mask else 0.0, ptr(xn), ptr(seed_base), stream())
        return (dx, dxm, xn) if mask is not None else (dx, xn)
    if mask is not None:
        dxm = torch.empty_like(x)
        call("hwgat_ln_bwd_masked", ptr(dy), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(dres), ptr(dx),
             ptr(dgamma), ptr(dbeta), x.numel() // d, d, dtype_code(x), ptr(dxm), mask[0] & 0xFFFFFFFF, float(mask[1]),
             ptr(seed_base), stream())
        return dx, dxm
    call("hwgat_ln_bwd", ptr(dy), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(dres), ptr(dx),
         ptr(dgamma), ptr(dbeta), x.numel() // d, d, dtype_code(x), stream())
    return dx


def dw_wants_xn(x):
    """True where the weight-gradient launch of a LayerNorm -> Linear pair takes LN(x) as a plain operand written by the
    LayerNorm backward (bf16, whole 256-wide tiles: the LDS-DMA kernel gemm_bf16_tn8w.hip) instead of normalising x in its
    loaders."""
    d = x.shape[-1]
    return x.dtype == torch.bfloat16 and d % 256 == 0 and (x.numel() // d) % 128 == 0


def transpose(W, dtype=torch.float32):
    """W^T of a (small) weight, optionally cast to the activation dtype"""
    R, C = W.shape
    out = torch.empty(C, R, device=W.device, dtype=torch.float32)
    call("hwgat_transpose_f32", ptr(W), ptr(out), R, C, stream())
    return out if dtype == torch.float32 else out.to(dtype)


def dropout_mask(shape, seed, p, device, seed_base=None):
    out = torch.empty(shape, device=device, dtype=torch.float32)
    call("hwgat_dropout_mask_f32", ptr(out), out.numel(), seed & 0xFFFFFFFF, float(p), ptr(seed_base), stream())
    return out


# ---------------------------------------------------------------- device-resident dropout seed
SEED_C1, SEED_C2, SEED_C3, SEED_SITE = 0x9E3779B1, 0x85EBCA77, 0x27D4EB2F, 0xC2B2AE35


def seed_base_value(initial, counter, salt):
    """host mirror of what hwgat_seed_set / hwgat_seed_advance leave in state[1]"""
    return (initial * SEED_C1 + counter * SEED_C2 + salt * SEED_C3) & 0xFFFFFFFF


def seed_set(state, counter, initial, salt):
    """state (4 int32 device words) <- {counter, base(counter), initial, salt}: the eager path, all four from host
    integers passed as kernel arguments (no H2D copy, no sync)"""
    call("hwgat_seed_set", ptr(state), counter & 0xFFFFFFFF, initial & 0xFFFFFFFF, salt & 0xFFFFFFFF, stream())


def seed_advance(state):
    """counter += 1 and the new base, on the device: the form a captured train step replays"""
    call("hwgat_seed_advance", ptr(state), stream())
