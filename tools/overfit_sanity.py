"""training sanity: each model family overfits 8 clips (loss must fall below a quarter of its start in 200 AdamW steps)"""
import importlib, sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hw = importlib.import_module("sl-hwgat_amd")
train = importlib.import_module("sl-hwgat_amd.train")
dev = torch.device("cuda:0")
torch.manual_seed(0)
for name, mk in (("HWGATE", lambda: hw.Model(*hw.HWGATEParams({"src_len": 32, "num_class": 5}, 2, dev, num_kps=32).get_model_params())),
                 ("HGATE", lambda: hw.HGATEModel(*hw.HGATEParams({"src_len": 32, "num_class": 5}, 2, dev).get_model_params())),
                 ("WGATE", lambda: hw.WGATEModel(*hw.WGATEParams({"src_len": 32, "num_class": 5}, 2, dev, num_kps=32).get_model_params()))):
    model = mk().train()
    K = model.num_kps
    x = torch.rand(8, 32, K, 2, device=dev)
    y = torch.randint(0, 5, (8,), device=dev)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, fused=True)
    step = train.TrainStep(model, opt, None)
    losses = []
    for i in range(200):
        step(x, y)
        if i % 40 == 0 or i == 199:
            losses.append(round(float(step.loss), 4))
    print(name, losses)
    assert losses[-1] < 0.25 * losses[0], name
print("overfit ok")
