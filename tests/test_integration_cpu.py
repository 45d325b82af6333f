"""CPU: the shim files under integration/models/ resolve exactly the way the reference resolves a model.

The reference finds a backend by FILE NAME: `importlib.import_module('models.' + cfg.model_type)` -> attribute `Model`,
built as `Model(*cfg.model_params.get_model_params())` and moved with `.to(cfg.device)` (hwgat/utils.py:55-59); the
hyper-parameters come from `getattr(importlib.import_module('models.model_params'), cfg.model_type + 'Params')
(dataset_params[ds], input_dim, device)` (hwgat/configs.py:80-82).  This test copies the shipped shims into a scratch
`models/` package the way a maintainer copies them into hwgat/models/ (the one-line edit of model_params.py included)
and runs those statements verbatim -- nothing of the reference tree is needed for it."""
import importlib
import importlib.util
import os
import shutil
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIMS = os.path.join(ROOT, "integration", "models")
dataset_params = {"INCLUDE": {"num_class": 262, "src_len": 64}, "FDMSE-ISL": {"num_class": 2002, "src_len": 192}}   # constants.py:3-13
input_dim = {"kp2D": 2, "kp3D": 3}                                                                                  # constants.py:17


@pytest.fixture()
def scratch_tree(tmp_path, monkeypatch):
    pkg = tmp_path / "hwgat" / "models"
    pkg.mkdir(parents=True)
    (pkg / "__init__.py").write_text("")                          # the reference's models/__init__.py is empty too
    for name in os.listdir(SHIMS):
        if name.endswith(".py"):
            shutil.copy(os.path.join(SHIMS, name), pkg / name)
    # the maintainer's one-line edit, on a stand-in for the reference's model_params.py
    (pkg / "model_params.py").write_text("class STGCNParams:\n    pass\n\n\nfrom models.model_params_amd import *\n")
    monkeypatch.setenv("HWGAT_AMD_ROOT", ROOT)                    # copied files no longer sit inside the checkout
    monkeypatch.syspath_prepend(str(tmp_path / "hwgat"))          # main.py runs with hwgat/ as the working directory
    for k in [k for k in sys.modules if k == "models" or k.startswith("models.")]:
        monkeypatch.delitem(sys.modules, k)
    yield
    for k in [k for k in sys.modules if k == "models" or k.startswith("models.")]:
        sys.modules.pop(k, None)


class _Cfg:
    """the four attributes of configs.py's runCFG that model construction reads"""

    def __init__(self, model_type, dataset_name, mode="kp2D"):
        self.model_type, self.dataset_name = model_type, dataset_name
        self.input_dim = input_dim[mode]
        self.device = torch.device("cpu")
        module = importlib.import_module('models.model_params')                                        # configs.py:80
        self.model_params = getattr(module, self.model_type + 'Params')(dataset_params[self.dataset_name],
                                                                        self.input_dim, self.device)   # configs.py:81-82


def load_model(cfg):                                                                                   # utils.py:55-59
    module = importlib.import_module('models.' + cfg.model_type)
    model = getattr(module, 'Model')(*cfg.model_params.get_model_params())
    model.to(cfg.device)
    return model


@pytest.mark.parametrize("name,cls", [("HWGATE_AMD", "Model"), ("HGATE_AMD", "HGATEModel"), ("WGATE_AMD", "WGATEModel")])
def test_backend_resolves_by_file_name_like_the_reference(scratch_tree, name, cls):
    hw = importlib.import_module("sl-hwgat_amd")
    cfg = _Cfg(name, "INCLUDE")
    assert type(cfg.model_params) is getattr(hw, cls.replace("Model", "") + "Params" if cls != "Model" else "HWGATEParams")
    model = load_model(cfg)
    assert type(model) is getattr(hw, cls)
    assert model.num_classes == 262 and model.temporal_dim == 64
    assert isinstance(model, torch.nn.Module) and sum(p.numel() for p in model.parameters() if p.requires_grad) > 1e6
    # what utils.train needs next: parameters() for AdamW (utils.py:76), state_dict() for checkpoints (utils.py:166)
    assert "B" in model.state_dict() and any(k.endswith("attn.qkv.weight") for k in model.state_dict())


def test_shims_work_in_place_without_the_environment_variable(monkeypatch):
    monkeypatch.delenv("HWGAT_AMD_ROOT", raising=False)
    spec = importlib.util.spec_from_file_location("_shim_in_place", os.path.join(SHIMS, "HWGATE_AMD.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.Model is importlib.import_module("sl-hwgat_amd").Model
