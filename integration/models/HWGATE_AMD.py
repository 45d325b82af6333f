"""Copy this file into the reference tree as hwgat/models/HWGATE_AMD.py and select it with `-model HWGATE_AMD`.

The reference resolves a model by file name -- importlib.import_module('models.' + cfg.model_type), attribute `Model`
(hwgat/utils.py:55-59) -- so all this file does is expose the MI355X backend's class under that name.  The class takes
the same positional tuple as hwgat/models/HWGATE.py::Model and has the same forward(x) and state_dict() keys.

The checkout that contains `sl-hwgat_amd/` is found through the HWGAT_AMD_ROOT environment variable; in place (this
file still under <checkout>/integration/models/) it is found without it."""
import importlib
import os
import sys

_root = os.environ.get("HWGAT_AMD_ROOT") or os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.path.isdir(os.path.join(_root, "sl-hwgat_amd")) and _root not in sys.path:
    sys.path.insert(0, _root)

Model = importlib.import_module("sl-hwgat_amd").Model
