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
