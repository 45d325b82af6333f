"""sl-hwgat_amd: MI355X (gfx950) backend for the HWGAT hot path.

The directory name carries a hyphen (it mirrors the upstream project name), so
import it with importlib:

    import importlib
    hw = importlib.import_module("sl-hwgat_amd")
    model = hw.Model(*hw.HWGATEParams({'src_len': 128, 'num_class': 2002}, 2, dev).get_model_params())
"""
from . import _lib
from . import functional
from . import checkpoint
from .parts import part_table
from .models.HWGATE import Model
from .models.HGATE import Model as HGATEModel
from .models.WGATE import Model as WGATEModel
from .models.model_params import HWGATEParams, HGATEParams, WGATEParams

__all__ = ["Model", "HWGATEParams", "HGATEModel", "HGATEParams", "WGATEModel", "WGATEParams", "functional",
           "part_table", "_lib", "checkpoint"]
