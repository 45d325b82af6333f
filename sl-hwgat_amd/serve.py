"""Evaluation forward captured in a HIP graph.

The reference evaluates with `model.eval(); model(x)` (hwgat/utils.py:119-134): ~280 kernel launches per forward here.  At
the training batch that is hidden behind the GPU time; at serving batches (B = 1 .. 8) the forward is launch-bound -- the host
issues launches slower than the GPU retires them.  `GraphedEval` records the eval forward of a model ONCE for a fixed input
shape (HIP stream capture through torch.cuda.graphs: every launch of the forward goes to torch's current stream, so the C-ABI
launches are captured like torch's own) and replays it per call: one graph launch instead of ~280 kernel launches.  The
eval forward is bit-reproducible, so a replay returns exactly the eager result.

    hw = importlib.import_module("sl-hwgat_amd")
    serve = importlib.import_module("sl-hwgat_amd.serve")
    fast = serve.GraphedEval(model, torch.empty(1, 128, 80, 2, device=dev))
    logits = fast(x)          # x: same shape / dtype / device as the example

The weights are read at replay time (the graph holds the launches, not the values): an optimizer step or load_state_dict
between calls is seen by the next replay, as long as no parameter is reallocated.  The graph holds raw ADDRESSES -- of the
parameters and of the derived weight copies (functional.WeightPrep) -- so a reallocated parameter (`model.to()`,
`load_state_dict(assign=True)`) would make a replay read stale or recycled memory: `GraphedEval` keeps the WeightPrep
objects it captured alive, compares the addresses of every source parameter with the captured ones on each call and
raises "capture again" on a mismatch.
"""
import torch

from . import functional as HF


class GraphedEval:
    def __init__(self, model, example, warmup=2):
        if model.training:
            raise ValueError("GraphedEval captures the eval() forward: call model.eval() first")
        if not example.is_cuda:
            raise ValueError("the example input must live on the GPU")
        self.model = model
        self.static_in = example.detach().clone()
        side = torch.cuda.Stream(device=example.device)
        side.wait_stream(torch.cuda.current_stream(example.device))
        with torch.no_grad(), torch.cuda.stream(side):        # warm-up off the default stream, as torch.cuda.graphs asks
            for _ in range(max(1, warmup)):
                model(self.static_in)
        torch.cuda.current_stream(example.device).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.static_out = model(self.static_in)
        # the derived weight copies the captured launches read and rewrite: referenced here so that a later eager forward,
        # which may replace the model's cache entry, cannot hand their memory back to the allocator under the graph
        self._preps = dict(model.__dict__.get("_weight_prep_cache", {}))
        self._captured = self._addresses()

    def _addresses(self):
        """addresses of everything the captured launches take by pointer from the model"""
        m = self.model
        sig = HF.WeightPrep.signature(m.block_list(), m.activation_dtype, False)[2] if hasattr(m, "block_list") else ()
        rest = tuple(t.data_ptr() for t in list(m.parameters()) + list(m.buffers()))
        return sig, rest

    def __call__(self, x):
        """logits for `x` (shape / dtype of the captured example); the result is a fresh tensor"""
        if x.shape != self.static_in.shape or x.dtype != self.static_in.dtype:
            raise ValueError(f"captured for {tuple(self.static_in.shape)} {self.static_in.dtype}, got {tuple(x.shape)} {x.dtype}")
        if self._addresses() != self._captured:
            raise RuntimeError("a parameter or buffer of the model was reallocated after the capture (model.to(), "
                               "load_state_dict(assign=True), ...): the graph holds the old addresses -- capture again")
        if self.model.training:
            raise RuntimeError("the model was switched to train() after the capture; GraphedEval replays the eval() forward")
        self.static_in.copy_(x, non_blocking=True)
        self.graph.replay()
        return self.static_out.clone()
