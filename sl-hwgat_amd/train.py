"""Loss and the fwd+bwd step harness of the hot path (reference hwgat/utils.py:93-116
and hwgat/losses/SmoothCrossEntropy.py), without the per-step host syncs."""
import torch
import torch.nn.functional as tF


class SmoothedCrossEntropyLoss(torch.nn.Module):
    """(1-eps)*NLL + eps*(-mean log p), batch mean; eps 0.01 as in the reference."""

    def __init__(self, smooth_factor: float = 0.01):
        super().__init__()
        self.smooth_factor = smooth_factor

    def forward(self, input, target):
        lp = tF.log_softmax(input.float(), dim=-1)
        nll = -lp.gather(-1, target.unsqueeze(1)).squeeze(1)
        return ((1.0 - self.smooth_factor) * nll + self.smooth_factor * (-lp.mean(-1))).mean()


class TrainStep:
    """zero_grad -> forward -> loss -> backward (-> bucketed all-reduce) -> optimizer.

    Unlike utils.train (utils.py:109,114) nothing here reads a value back to the
    host: loss and correct-count stay on the device until the caller asks.

    `micro_batch`: process the batch in slices of that many clips and accumulate gradients
    (same mean-loss gradient).  Activations saved for backward are 10*E floats per block
    (DESIGN.md section 3); BASELINE config 5 (B=256, T=256, K=112, d0=256) would need ~600 GB
    in one piece, 16-clip slices need ~38 GB."""

    def __init__(self, model, optimizer=None, reducer=None, criterion=None, micro_batch=None):
        self.model, self.opt, self.reducer = model, optimizer, reducer
        self.criterion = criterion or SmoothedCrossEntropyLoss()
        self.micro_batch = micro_batch
        self.loss = None
        self.correct = None

    def __call__(self, x, y):
        n = x.shape[0]
        mb = self.micro_batch if self.micro_batch and self.micro_batch < n else n
        n_acc = (n + mb - 1) // mb
        if self.reducer is not None:
            self.reducer.zero_grad(n_acc)
        elif self.opt is not None:
            self.opt.zero_grad(set_to_none=True)
        total = None
        correct = None
        for i in range(0, n, mb):
            xs, ys = x[i:i + mb], y[i:i + mb]
            out = self.model(xs)
            loss = self.criterion(out, ys) * (xs.shape[0] / n)
            loss.backward()
            total = loss.detach() if total is None else total + loss.detach()
            c = (out.detach().argmax(-1) == ys).sum()
            correct = c if correct is None else correct + c
        if self.reducer is not None:
            self.reducer.finish()
        if self.opt is not None:
            self.opt.step()
        self.loss, self.correct = total, correct
        return self.loss


class GraphedTrainStep:
    """The whole train step -- seed advance, forward, loss, backward, fused AdamW -- captured ONCE in a HIP graph and
    replayed per call: one graph launch instead of ~400 kernel launches issued one by one from Python (reference loop:
    hwgat/utils.py:93-116, which issues its ~40 ATen ops per block the same way).  Why it matters here: the bf16 steps
    of the sibling models are 10-17 ms, within 10 % of what one Python thread can issue, and a node runs 8 such ranks
    (SURVEY 8e); a replay needs no host work between kernels.

    What makes the capture possible (round 4): nothing in a step depends on a host value that changes between steps.
    The dropout seed lives on the device (`model._seed_state`; every seeded kernel adds the step's base seed, which
    hwgat_seed_advance -- the first node of the graph -- rewrites), the train-mode thresholds of HWGATE.py:96 are drawn
    by torch's graph-safe device generator, the optimizer is AdamW(fused=True, capturable=True) (its step counter is a
    device tensor), and every C-ABI launch goes to torch's current stream, so HIP stream capture records it like
    torch's own kernels.  Capture follows the torch.cuda.graphs recipe (warm-up on a side stream, then
    `torch.cuda.graph`), in THIS process: nothing is re-launched or exec'ed.

    Inputs are copied into static buffers; `loss` / `correct` are static device tensors rewritten by every replay.
    Shapes are fixed at capture.  Parameters must not be reallocated afterwards (same rule as serve.GraphedEval).
    With `model._drop_calls = c` before construction, replay k (1-based) draws exactly the masks the eager TrainStep
    draws in its k-th step from the same `c` (tests/test_gpu_graph.py)."""

    def __init__(self, model, optimizer, x, y, criterion=None, warmup=2, reducer=None):
        import importlib
        HF = importlib.import_module(__package__ + ".functional")
        if not x.is_cuda:
            raise ValueError("GraphedTrainStep needs device inputs")
        if not model.training:
            raise ValueError("GraphedTrainStep captures the train() step: call model.train() first")
        for grp in optimizer.param_groups:
            if not grp.get("capturable", False):
                raise ValueError("the optimizer must be built with capturable=True (its step count then lives on the device)")
        self.model, self.opt, self.reducer = model, optimizer, reducer
        self.criterion = criterion or SmoothedCrossEntropyLoss()
        self.x, self.y = x.detach().clone(), y.detach().clone()
        start = int(model._drop_calls)
        dev = x.device
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        # warm-up (allocator, weight-prep caches, optimizer state) on a side stream with the weights and the optimizer
        # state put back afterwards, so that capture changes nothing the caller can observe
        keep_p = [p.detach().clone() for p in model.parameters()]
        keep_state = {p: {k: v.clone() for k, v in st.items() if torch.is_tensor(v)} for p, st in optimizer.state.items()}
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self._one()
        torch.cuda.current_stream(dev).wait_stream(side)
        with torch.no_grad():
            for p, q in zip(model.parameters(), keep_p):
                p.copy_(q)
            for p, st in optimizer.state.items():        # moments / step counters: back to what they were (zero if new)
                for k, v in st.items():
                    if torch.is_tensor(v):
                        if p in keep_state and k in keep_state[p]:
                            v.copy_(keep_state[p][k])
                        else:
                            v.zero_()
        del keep_p, keep_state
        model.device_seed_counter = True                 # from here on the device counts the steps itself
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)
        with torch.cuda.graph(self.graph):
            self.loss, self.correct = self._one()
        # capture executed nothing: put the device counter where the host mirror says the caller left it
        model._drop_calls = start
        HF.seed_set(model._seed_state, start, torch.initial_seed(), getattr(model, "rank_salt", 0))
        self._addr = self._addresses()

    def _one(self):
        self.opt.zero_grad(set_to_none=True)
        out = self.model(self.x)
        loss = self.criterion(out, self.y)
        loss.backward()
        if self.reducer is not None:
            self.reducer.finish()
        self.opt.step()
        return loss.detach(), (out.detach().argmax(-1) == self.y).sum()

    def _addresses(self):
        return tuple(t.data_ptr() for t in list(self.model.parameters()) + list(self.model.buffers()))

    def __call__(self, x, y):
        if x.shape != self.x.shape or y.shape != self.y.shape:
            raise ValueError(f"captured for {tuple(self.x.shape)} / {tuple(self.y.shape)}, got {tuple(x.shape)} / {tuple(y.shape)}")
        if self._addresses() != self._addr:
            raise RuntimeError("a parameter or buffer of the model was reallocated after the capture: capture again")
        if x.data_ptr() != self.x.data_ptr():
            self.x.copy_(x, non_blocking=True)
        if y.data_ptr() != self.y.data_ptr():
            self.y.copy_(y, non_blocking=True)
        self.graph.replay()
        self.model._drop_calls += 1                      # host mirror of the device counter (model._seeds() in tests)
        return self.loss
