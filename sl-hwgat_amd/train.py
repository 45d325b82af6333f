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
    host: loss and correct-count stay on the device until the caller asks."""

    def __init__(self, model, optimizer=None, reducer=None, criterion=None):
        self.model, self.opt, self.reducer = model, optimizer, reducer
        self.criterion = criterion or SmoothedCrossEntropyLoss()
        self.loss = None
        self.correct = None

    def __call__(self, x, y):
        if self.reducer is not None:
            self.reducer.zero_grad()
        elif self.opt is not None:
            self.opt.zero_grad(set_to_none=True)
        out = self.model(x)
        loss = self.criterion(out, y)
        loss.backward()
        if self.reducer is not None:
            self.reducer.finish()
        if self.opt is not None:
            self.opt.step()
        self.loss = loss.detach()
        self.correct = (out.detach().argmax(-1) == y).sum()
        return self.loss
