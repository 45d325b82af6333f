"""CPU: host logic of the backend -- C-ABI exports, drop-in boundary, mask rows."""
import ctypes
import importlib

import numpy as np
import pytest
import torch

from oracle import hwgat_oracle as O
from helpers import load_fixture

hw = importlib.import_module("sl-hwgat_amd")


def test_library_loads_and_exports_every_declared_symbol():
    names = hw._lib.declared_symbols()
    assert len(names) >= 10
    handle = hw._lib.lib()
    for n in names:
        assert getattr(handle, n) is not None, n
    assert set(names) == set(hw._lib._SIGS), set(names) ^ set(hw._lib._SIGS)
    assert handle.hwgat_abi_version() >= 1000


def test_state_dict_contract_matches_reference_keys():
    fx = load_fixture("cfg1.npz")
    hp = hw.HWGATEParams({"src_len": 32, "num_class": 10}, 2, torch.device("cpu"), num_kps=32)
    assert np.array_equal(hp.adj_mat.numpy(), fx["adj"])          # reference get_adj_mat()
    model = hw.Model(*hp.get_model_params())
    sd = model.state_dict()
    want = dict(O.param_shapes(kp_dim=2, temporal_dim=32, num_classes=10, num_kps=32))
    for k, shape in want.items():
        assert tuple(sd[k].shape) == tuple(shape), k
    extra = set(sd) - set(want)
    assert extra == {k[5:] for k in fx if k.startswith("mask.")}  # attn_mask of odd blocks only
    for k in extra:
        assert np.array_equal(sd[k].numpy(), fx["mask." + k]), k
    assert not model.B.requires_grad and torch.equal(sd["pos_encoder.pe"], O.sinusoid_table(32, 128))
    # reference-shaped checkpoints load
    model.load_state_dict(O.synth_params(11, kp_dim=2, temporal_dim=32, num_classes=10, num_kps=32), strict=False)
    # default tuple of the reference (K=64) also builds
    hp64 = hw.HWGATEParams({"src_len": 64, "num_class": 262}, 2, None)
    assert hp64.adj_mat.shape == (4, 32, 32) and len(hp64.get_model_params()) == 16


def test_forward_without_gpu_fails_loudly():
    hp = hw.HWGATEParams({"src_len": 16, "num_class": 5}, 2, None, num_kps=32)
    model = hw.Model(*hp.get_model_params())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model(torch.rand(1, 16, 32, 2))


def test_mask_rows_equal_reference_masks():
    for nW in (2, 5):
        adj = O.window_adjacency(nW)
        bits = hw.functional.mask_bits(adj).numpy().astype(np.int64) & 0xFFFFFFFF
        key = np.arange(32)
        plain = (bits[0][:, :, None] >> key) & 1
        last = (bits[1][:, :, None] >> key) & 1
        assert np.array_equal(plain, adj.numpy().astype(np.int64))
        sm = O.shift_mask(8, nW).view(4, nW, 32, 32)
        assert np.array_equal(last, (adj * sm[-1]).numpy().astype(np.int64))
        assert np.array_equal(plain, (adj * sm[0]).numpy().astype(np.int64))
    with pytest.raises(ValueError):
        hw.functional.mask_bits(torch.full((1, 32, 32), 0.5))


def test_part_tables():
    fx = load_fixture("window_create.npz")
    idx = hw.part_table(29).numpy()
    assert np.array_equal(fx["raw"][:, idx].astype(np.float32), fx["out"])   # == WindowCreate
    for J, nW in ((27, 2), (67, 5), (133, 7)):
        t = hw.part_table(J).numpy()
        assert t.shape == (nW * 16,) and t.min() >= 0 and t.max() < J
        assert all((t[w * 16:w * 16 + 3] == [0, 1, 2]).all() for w in range(nW))
