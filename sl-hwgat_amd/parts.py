"""Body-part window tables: which raw joint feeds each of the nW*16 model slots.

A part window has 16 slots laid out [3 head | 3 arm | 10 hand], the order the
25-edge part graph of `HWGATEParams` assumes (reference
hwgat/models/model_params.py:261-287).  `reference_29()` reproduces the
reference's host-side `WindowCreate` gather exactly (hwgat/dataTransform.py:
428-455); on the device the gather is fused into the embedding kernel
(`hwgat_embed_fwd`), so no (T,64,C) copy is ever materialised.

The other tables map the joint counts named in BASELINE.json (27, 67, 133) to
nW = 2, 5, 7 windows.  The reference ships no tables for those (it only ever
runs 29 -> 4 windows), so these are this project's own, documented, synthetic
choices: head = joints 0..2, the remaining joints are split into `arm` and
`hand` pools and dealt to the windows cyclically.
"""
import torch

HEAD = [0, 1, 2]


def reference_29():
    l_arm, r_arm = [3, 5, 7], [4, 6, 8]
    l_hand, r_hand = list(range(9, 19)), list(range(19, 29))
    return HEAD + l_arm + l_hand + HEAD + r_arm + r_hand + HEAD + l_arm + r_hand + HEAD + r_arm + l_hand


def synthetic(num_joints: int, n_windows: int):
    if num_joints < 16:
        raise ValueError("need at least 16 joints for one part window")
    rest = list(range(3, num_joints))
    n_arm = max(3, min(len(rest) // 4, 3 * n_windows))
    arm, hand = rest[:n_arm], rest[n_arm:]
    out = []
    for w in range(n_windows):
        a = [arm[(3 * w + i) % len(arm)] for i in range(3)]
        h = [hand[(10 * w + i) % len(hand)] for i in range(10)]
        out += HEAD + a + h
    return out


def part_table(num_joints: int, n_windows: int = None) -> torch.Tensor:
    """int32 (nW*16,) joint index per model slot."""
    if num_joints == 29 and n_windows in (None, 4):
        idx = reference_29()
    else:
        if n_windows is None:
            n_windows = {27: 2, 67: 5, 133: 7}.get(num_joints)
            if n_windows is None:
                raise ValueError(f"no default window count for {num_joints} joints")
        idx = synthetic(num_joints, n_windows)
    return torch.tensor(idx, dtype=torch.int32)
