"""The three `<model_type>Params` names hwgat/configs.py:80-82 looks up in `models.model_params`
(`getattr(module, self.model_type + 'Params')(dataset_params[ds], input_dim, device)`).

Copy this file next to the reference's hwgat/models/model_params.py and add ONE line at the end of that file:

    from models.model_params_amd import *        # HWGATE_AMDParams, HGATE_AMDParams, WGATE_AMDParams

The classes keep the attribute names, defaults, adjacency builders and `get_model_params()` tuples of the reference's
HWGATEParams / HGATEParams / WGATEParams (hwgat/models/model_params.py:243-403, 5-240, 405-605).  The checkout is found
as in HWGATE_AMD.py (HWGAT_AMD_ROOT, or in place)."""
import importlib
import os
import sys

_root = os.environ.get("HWGAT_AMD_ROOT") or os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.path.isdir(os.path.join(_root, "sl-hwgat_amd")) and _root not in sys.path:
    sys.path.insert(0, _root)
_hw = importlib.import_module("sl-hwgat_amd")

HWGATE_AMDParams = _hw.HWGATEParams
HGATE_AMDParams = _hw.HGATEParams
WGATE_AMDParams = _hw.WGATEParams

__all__ = ["HWGATE_AMDParams", "HGATE_AMDParams", "WGATE_AMDParams"]
