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
    assert handle.hwgat_abi_version() == hw._lib.header_abi_version() >= 3000


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


def test_c_abi_rejects_bad_arguments_without_launching():
    """argument validation happens before any HIP call, so it is testable without a GPU"""
    L = hw._lib.lib()
    import ctypes
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert L.hwgat_win_attn_fwd(None, p, p, None, 1, 4, 1, 2, 64, 0, 0, None) == -1          # HWGAT_EINVAL
    assert L.hwgat_win_attn_fwd(p, p, p, None, 1, 3, 1, 2, 64, 0, 0, None) == -2             # odd F: HWGAT_ESHAPE
    assert L.hwgat_win_attn_fwd(p, p, p, None, 1, 4, 1, 2, 48, 0, 0, None) == -2             # head_dim 48
    assert L.hwgat_win_attn_bwd(p, p, p, p, None, 1, 4, 1, 2, 64, 0, 7, None) == -3          # HWGAT_EDTYPE
    assert L.hwgat_ln_fwd(p, p, p, p, p, p, 8, 100, 0, None) == -2                           # width 100
    assert L.hwgat_linear_nt_f32(p, p, None, p, 128, 100, 128, 0, None, None, None, None, 0, 0.0, 0,
                                 None, None, None, 0, 0.0, None, None) == -2                        # N % 128 (any M is fine)
    assert L.hwgat_linear_nt_f32(p, p, None, p, 0, 128, 128, 0, None, None, None, None, 0, 0.0, 0,
                                 None, None, None, 0, 0.0, None, None) == -1                        # M <= 0
    assert L.hwgat_linear_nt_f32(p, p, None, p, 128, 128, 128, 1, None, None, None, None, 0, 0.0, 0,
                                 None, None, None, 0, 0.0, None, None) == -1                        # LN prologue w/o stats
    assert L.hwgat_linear_nt_f32(p, p, None, p, 128, 128, 128, 0, None, None, None, None, 0, 0.0, 1,
                                 None, None, None, 0, 1.5, None, None) == -1                        # residual missing / p >= 1
    assert L.hwgat_linear_tn_f32(p, p, p, None, 64, 128, 100, 0, 0.0, None, None, None, None, None, None) == -2
    assert L.hwgat_embed_fwd(p, None, p, None, p, 1, 4, 29, 64, 2, 128, 0, 0, 0.0, None, None) == -2   # J != K without a table
    assert L.hwgat_merge(p, p, 1, 3, 16, 128, 0, 0, None) == -2


@pytest.mark.skipif(not __import__("os").path.exists("/root/reference/hwgat/models/HWGATE.py"),
                    reason="reference tree only exists in the development container")
@pytest.mark.parametrize("family", ["HWGATE", "HGATE", "WGATE"])
def test_checkpoint_interchange_with_the_reference_class(family):
    """state_dict of this backend loads STRICTLY into the reference Model and vice versa
    (SURVEY 8b / 8f-4), for the headline model and the sibling HGATE (8f-3).
    Development container only; nothing here runs on the GPU box."""
    import sys, types
    for name in ("timm", "timm.models", "timm.models.layers"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["timm.models.layers"].trunc_normal_ = torch.nn.init.trunc_normal_   # init-only import, HWGATE.py:4
    sys.path.insert(0, "/root/reference/hwgat")
    try:
        ref_mod = importlib.import_module("models." + family)
        ref_par = importlib.import_module("models.model_params")
    finally:
        sys.path.remove("/root/reference/hwgat")
    rp = getattr(ref_par, family + "Params")({"src_len": 64, "num_class": 11}, 2, torch.device("cpu"))
    ref = ref_mod.Model(*rp.get_model_params())
    hp = getattr(hw, family + "Params")({"src_len": 64, "num_class": 11}, 2, torch.device("cpu"))
    assert [type(a) for a in hp.get_model_params()] == [type(a) for a in rp.get_model_params()]
    assert torch.equal(hp.adj_mat, rp.adj_mat)
    mine = {"HWGATE": hw.Model, "HGATE": hw.HGATEModel, "WGATE": hw.WGATEModel}[family](*hp.get_model_params())
    sd_ref, sd_mine = ref.state_dict(), mine.state_dict()
    assert list(sd_ref.keys()) == list(sd_mine.keys())                       # same keys, same order
    assert all(sd_ref[k].shape == sd_mine[k].shape and sd_ref[k].dtype == sd_mine[k].dtype for k in sd_ref)
    mine.load_state_dict(sd_ref, strict=True)
    ref.load_state_dict(sd_mine, strict=True)
    for k in sd_ref:
        if k.endswith("attn_mask") or k in ("pos_encoder.pe", "adj_mask"):
            assert torch.equal(sd_ref[k], sd_mine[k]), k                      # derived buffers are identical
    assert not mine.B.requires_grad and not ref.B.requires_grad
    assert sum(p.numel() for p in mine.parameters() if p.requires_grad) == \
        sum(p.numel() for p in ref.parameters() if p.requires_grad)


def test_sibling_mask_rows_on_cpu():
    """host-side mask compaction of the sibling models (no GPU needed): HGATE block bits decode back to the
    adjacency; WGATE band rows decode back to the three 16x16 blocks and malformed adjacencies are rejected"""
    from oracle import hgat_oracle as OH, wgat_oracle as OW
    HF = hw.functional
    adj = OH.block_adjacency()
    bits = HF.blk_mask_bits(adj, 29).numpy().astype(np.int64) & 0xFFFFFFFF
    assert bits.shape == (2, 64, 2)
    for tq in range(2):
        for jq in range(29):
            for tk in range(2):
                row = [(int(bits[0, tq * 32 + jq, tk]) >> j) & 1 for j in range(32)]
                assert row[:29] == [int(v) for v in adj[tq * 29 + jq, tk * 29:tk * 29 + 29]] and not any(row[29:])
                last = int(bits[1, tq * 32 + jq, tk])
                assert last == (int(bits[0, tq * 32 + jq, tk]) if tq == tk else 0)
    assert not bits[:, 29:32].any() and not bits[:, 61:64].any()              # pad query slots
    with pytest.raises(ValueError):
        HF.blk_mask_bits(adj * 0.5, 29)
    with pytest.raises(ValueError):
        HF.blk_mask_bits(torch.ones(70, 70), 35)

    T, nW = 5, 2
    band = OW.band_adjacency(T, nW)
    rows = HF.band_mask_rows(band, T)
    assert rows.shape == (nW, 16) and rows.dtype == torch.int64
    pa = OW.part_adjacency()
    for i in range(16):
        r = int(rows[1, i])
        assert [(r >> j) & 1 for j in range(16)] == [int(i == j) for j in range(16)]                 # previous frame
        assert [(r >> (16 + j)) & 1 for j in range(16)] == [int(v) for v in pa[i]]                   # same frame
        assert [(r >> (32 + j)) & 1 for j in range(16)] == [int(i == j) for j in range(16)]          # next frame
    assert HF.band_mask_rows(OW.band_adjacency(1, 1), 1).shape == (1, 16)                            # T = 1: no neighbours
    far = band.clone()
    far[0, 0, 3 * 16] = 1
    with pytest.raises(NotImplementedError):
        HF.band_mask_rows(far, T)
    no_diag = band.clone()
    no_diag[1, 2 * 16 + 4, 2 * 16:3 * 16] = 0
    with pytest.raises(NotImplementedError):
        HF.band_mask_rows(no_diag, T)
    with pytest.raises(ValueError):
        HF.band_mask_rows(band, T + 1)


def test_attention_dropout_is_a_constructor_hyper_parameter():
    """reference HWGATE.py:273 `attn_drop_rate`: accepted by the HWGATE backend (stored, used in train mode only), range
    checked; four site seeds per block (proj, fc1, fc2, attention); the sibling backends take it too (HGATE.py:232,
    WGATE.py:165) with the same range check"""
    hp = hw.HWGATEParams({"src_len": 16, "num_class": 5}, 2, "cpu", num_kps=32)
    hp.attn_drop_rate = 0.1
    m = hw.Model(*hp.get_model_params())
    assert m.attn_drop_rate == 0.1
    assert len(m._seeds(0)) == 4 and len(set(m._seeds(0) + m._seeds(1))) == 8
    hp.attn_drop_rate = 1.0
    with pytest.raises(ValueError):
        hw.Model(*hp.get_model_params())
    with pytest.raises(ValueError):
        hw.functional.window_attention(torch.zeros(1, 2, 16, 3 * 64), torch.zeros(2, 1, 32, dtype=torch.int32), None, 1, False,
                                       drop=(1, 0.1))         # eval mode (no threshold) has no dropout
    for params, cls in ((hw.HGATEParams({"src_len": 16, "num_class": 5}, 2, "cpu"), hw.HGATEModel),
                        (hw.WGATEParams({"src_len": 16, "num_class": 5}, 2, "cpu", num_kps=32), hw.WGATEModel)):
        params.attn_drop_rate = 0.2
        assert cls(*params.get_model_params()).attn_drop_rate == 0.2
        params.attn_drop_rate = -0.1
        with pytest.raises(ValueError):
            cls(*params.get_model_params())


def test_dropout_seeds_differ_per_rank_and_per_call():
    """data-parallel ranks share torch's seed but not dropout masks: rank_salt enters the site seeds"""
    hp = hw.HWGATEParams({"src_len": 32, "num_class": 5}, 2, torch.device("cpu"), num_kps=32)
    torch.manual_seed(1001)
    m = hw.Model(*hp.get_model_params())
    base = m._seeds(3)
    assert len(set(base)) == 4 and m._seeds(3) == base and m._seeds(4) != base
    m.rank_salt = 1
    assert m._seeds(3) != base and len(set(m._seeds(3)) & set(base)) == 0
    m.rank_salt = 0
    m._drop_calls += 1
    assert m._seeds(3) != base
