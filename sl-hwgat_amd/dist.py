"""Data-parallel gradient exchange: one process per GPU, RCCL over xGMI.

Clips shard along batch (every op of the model is per-clip), so
the only exchange is an all-reduce(mean) of the parameter gradients.  The
reference has no distributed code at all; this is new.

Design for xGMI (point-to-point links, ring collectives are per-link bound):
gradients live permanently inside a few large flat buckets (parameters' `.grad`
are views, so nothing is packed or copied per step); buckets follow backward
order (head + last stage first) and each bucket's all-reduce is launched
asynchronously from a post-accumulate hook the moment its last gradient is
written, overlapping RCCL with the rest of backward.  ~10.8 M fp32 gradients
-> 3 buckets of <= 16 MiB by default.
"""
from typing import List

import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, params, bucket_bytes: int = 16 << 20, group=None, always_reduce: bool = False):
        self.group = group
        self.always_reduce = always_reduce          # issue the collectives even at world size 1 (rehearsal)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        params = [p for p in params if p.requires_grad]
        order = list(reversed(params))                       # backward visits parameters roughly in reverse
        self.buckets: List[dict] = []
        cur, cur_bytes = [], 0
        for p in order:
            nbytes = p.numel() * p.element_size()
            if cur and (cur_bytes + nbytes > bucket_bytes or p.dtype != cur[0].dtype):
                self._close(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self._close(cur)
        self._handles = []
        self._in_hook = False
        self.trace = None          # tests: a list receives ("launch", bucket index, "hook" | "finish") per collective issued
        self._use_avg = dist.is_initialized() and dist.get_backend(group) == "nccl"
        for b in self.buckets:
            for p in b["params"]:
                p.register_post_accumulate_grad_hook(self._make_hook(b))

    def _close(self, plist):
        total = sum(p.numel() for p in plist)
        flat = torch.zeros(total, dtype=plist[0].dtype, device=plist[0].device)
        off = 0
        for p in plist:
            p.grad = flat[off:off + p.numel()].view_as(p)    # .grad is a view into the bucket
            off += p.numel()
        self.buckets.append({"flat": flat, "params": plist, "pending": len(plist), "index": len(self.buckets)})

    def _make_hook(self, bucket):
        def hook(_p):
            bucket["pending"] -= 1
            if bucket["pending"] == 0:
                self._in_hook = True          # we are inside autograd's backward: the rest of it overlaps the collective
                try:
                    self._launch(bucket)
                finally:
                    self._in_hook = False
        return hook

    def _launch(self, bucket):
        if self.trace is not None:
            self.trace.append(("launch", bucket["index"], "hook" if self._in_hook else "finish"))
        if self.world > 1 or (self.always_reduce and dist.is_initialized()):
            if self._use_avg:
                h = dist.all_reduce(bucket["flat"], op=dist.ReduceOp.AVG, group=self.group, async_op=True)
            else:
                bucket["flat"].div_(self.world)
                h = dist.all_reduce(bucket["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._handles.append(h)

    def zero_grad(self, n_accum: int = 1):
        """call instead of optimizer.zero_grad(): keeps the bucket views alive.
        `n_accum`: number of backward passes (micro-batches) that accumulate into the buckets
        before they are reduced; a bucket is launched when its last gradient of the LAST pass lands."""
        self._n_accum = n_accum
        for b in self.buckets:
            b["flat"].zero_()
            b["pending"] = len(b["params"]) * n_accum
            for p in b["params"]:                            # re-attach if something replaced .grad
                if p.grad is None or p.grad.data_ptr() < b["flat"].data_ptr() or \
                        p.grad.data_ptr() >= b["flat"].data_ptr() + b["flat"].numel() * b["flat"].element_size():
                    raise RuntimeError("a parameter's .grad was detached from its bucket; "
                                       "use GradReducer.zero_grad(), not set_to_none=True")

    def finish(self):
        """wait for every bucket (call after backward, before optimizer.step)."""
        for b in self.buckets:
            if b["pending"] != 0 and b["pending"] != len(b["params"]) * getattr(self, "_n_accum", 1):
                # some parameters of this bucket got no gradient this step: reduce what is there
                self._launch(b)
        for h in self._handles:
            h.wait()
        self._handles.clear()


def broadcast_parameters(module, src: int = 0, group=None):
    """make every rank start from rank `src`'s parameters and buffers"""
    if not dist.is_initialized():
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)
    # every rank draws its own dropout masks (the models hash (seed, rank_salt, call, site, element))
    module.rank_salt = dist.get_rank(group)
