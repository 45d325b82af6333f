"""Which stage of the eval forward differs between two runs?  (GPU box; prints the first tap that is not bit-equal.)"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
hw = importlib.import_module("sl-hwgat_amd")
HF = hw.functional
dev = torch.device("cuda:0")
dtype = torch.bfloat16 if "bf16" in sys.argv else torch.float32
torch.manual_seed(1001)
hp = hw.HWGATEParams({"src_len": 128, "num_class": 2002}, 2, dev, num_kps=80, embed_dim=128)
model = hw.Model(*hp.get_model_params()).to(dev).set_activation_dtype(dtype).eval()
for p in model.parameters():
    if p.requires_grad and p.dim() > 1:
        p.data.normal_(0, 0.08)
x = torch.rand(64, 128, 80, 2, device=dev)
runs = []
for r in range(3):
    taps = {}
    orig = model._block
    k = [0]

    def tapped(*a):
        out = orig(*a)
        hand = a[-1]
        taps[f"block{k[0]}"] = out.detach().clone()
        if hand.stats is not None:
            taps[f"block{k[0]}.mean"] = hand.stats[0].detach().clone()
            taps[f"block{k[0]}.rstd"] = hand.stats[1].detach().clone()
        k[0] += 1
        return out
    model._block = tapped
    with torch.no_grad():
        taps["embed"] = model._embed(x).clone()
        feat = model.forward_features(x)
        taps["feat"] = feat.clone()
        taps["logits"] = model.head(feat).clone()
        taps["logits_again_same_feat"] = model.head(feat).clone()
    model._block = orig
    runs.append(taps)
    torch.randn(1 << 24, device=dev).sum().item()
for name in runs[0]:
    same = [torch.equal(runs[0][name], runs[i][name]) for i in (1, 2)]
    diff = max(float((runs[0][name].float() - runs[i][name].float()).abs().max()) for i in (1, 2))
    print(f"{name:28s} equal={same} maxdiff={diff:.3e}")
print("head twice on the same features equal:", torch.equal(runs[0]["logits"], runs[0]["logits_again_same_feat"]))
