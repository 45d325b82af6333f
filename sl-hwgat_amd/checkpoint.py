"""Checkpoint files, optimizer and LR schedule in the reference's formats (reference hwgat/utils.py:71-88, 164-237).

The reference's `utils.py` cannot be imported without its dataset stack (decord, matplotlib); what `main.py` needs from
it for THIS backend's model is small and restated here with the same names, argument meaning and file layout, so a
checkpoint written by either side loads on the other:

  save_checkpoint               utils.py:164-176   torch.save of one dict: 'model_state_dict', 'optimizer_state_dict',
                                                   'train_loss_list', 'val_loss_list', 'train_acc_list', 'val_acc_list',
                                                   'epoch', 'learning_rate', 'scheduler'
  load_weights_from_pretrained  utils.py:185-214   strips a "model." prefix from every key, keeps a tensor only when the
                                                   key exists in the model AND the shapes agree, reports the rest
  load_checkpoint               utils.py:216-237   weights as above + optimizer / scheduler state, start epoch, 4 lists
  get_optimizer / get_scheduler utils.py:71-88     AdamW(lr) by default; CosineAnnealingLR(T_max=20, last_epoch=-1)
"""
from typing import Optional

import torch

_LISTS = ("train_loss_list", "val_loss_list", "train_acc_list", "val_acc_list")


def get_optimizer(model, lr: float = 5e-4, optimizer_type: str = "adamw", fused: Optional[bool] = None):
    """utils.py:71-82 (configs.py:84: lr 5e-4, 'adamw').  `fused=True` selects torch's single-launch AdamW on a GPU."""
    # ALL parameters, as the reference passes them (utils.py:76: model.parameters()): the frozen Fourier matrix `B`
    # is an nn.Parameter and therefore entry 0 of the param group, so optimizer_state_dict files interchange with the
    # reference's AdamW (param-group sizes must match).  AdamW skips parameters whose grad is None: B gets no update.
    params = list(model.parameters())
    kinds = {"adamw": torch.optim.AdamW, "adam": torch.optim.Adam, "nadam": torch.optim.NAdam, "sgd": torch.optim.SGD}
    if optimizer_type not in kinds:
        raise ValueError(f"optimizer_type {optimizer_type!r}: one of {sorted(kinds)}")
    kw = {"fused": fused} if fused is not None and optimizer_type in ("adamw", "adam") else {}
    return kinds[optimizer_type](params, lr=lr, **kw)


def get_scheduler(optimizer, scheduler: Optional[str] = "CosineAnnealingLR"):
    """utils.py:84-89: cosine annealing over 20 scheduler steps (one per epoch in run_epochs, utils.py:263), else None"""
    if scheduler == "CosineAnnealingLR":
        return torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=20, last_epoch=-1)
    return None


def save_checkpoint(path, model, optimizer, scheduler, train_acc_list, train_loss_list, val_acc_list, val_loss_list,
                    epoch, lr):
    """same argument order and the same dict layout as utils.py:164-176"""
    torch.save({
        "model_state_dict": model.state_dict(),
        "optimizer_state_dict": optimizer.state_dict(),
        "train_loss_list": train_loss_list,
        "val_loss_list": val_loss_list,
        "train_acc_list": train_acc_list,
        "val_acc_list": val_acc_list,
        "epoch": epoch,
        "learning_rate": lr,
        "scheduler": scheduler.state_dict(),
    }, path)


def _read(path, device):
    # a checkpoint is tensors + python scalars / lists: the safe loader suffices (and is required for files that
    # were not written by this process)
    return torch.load(path, map_location=device, weights_only=True)


def load_weights_from_pretrained(model, pretrained_model_path, device="cpu", report=None):
    """utils.py:185-214.  Returns the model; `report` (a dict, optional) receives the three lists the reference
    prints: keys skipped because absent from the model, keys skipped because of a shape mismatch, model keys the
    file did not provide."""
    ckpt = _read(pretrained_model_path, device)["model_state_dict"]
    pretrained = {k.replace("model.", ""): v for k, v in ckpt.items()}
    own = model.state_dict()
    unknown = [k for k in pretrained if k not in own]
    mismatched = [k for k in pretrained if k in own and pretrained[k].shape != own[k].shape]
    missing = [k for k in own if k not in pretrained]
    for k, v in pretrained.items():
        if k in own and v.shape == own[k].shape:
            own[k] = v
    model.load_state_dict(own)
    model.to(dtype=torch.float)
    if report is not None:
        report.update(unknown=unknown, mismatched=mismatched, missing=missing)
    return model


def load_checkpoint(path, model, optimizer, scheduler, device="cpu", model_weights=None):
    """utils.py:216-237 with the model / optimizer / scheduler passed in (the reference builds them from its cfg).
    Returns (model, optimizer, scheduler, [train_loss, val_loss, train_acc, val_acc], start_epoch)."""
    if model_weights is not None:                         # fine-tuning: weights only, fresh optimizer state
        return load_weights_from_pretrained(model, model_weights, device), optimizer, scheduler, [[], [], [], []], 0
    ckpt = _read(path, device)
    load_weights_from_pretrained(model, path, device)
    optimizer.load_state_dict(ckpt["optimizer_state_dict"])
    scheduler.load_state_dict(ckpt["scheduler"])
    return model, optimizer, scheduler, [ckpt[k] for k in _LISTS], ckpt["epoch"] + 1
