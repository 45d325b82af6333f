"""Hyper-parameters + part-graph adjacency for the MI355X HWGAT backend.

Drop-in for `HWGATEParams` of the reference (hwgat/models/model_params.py:
243-403): same attribute names, same defaults, same positional tuple from
`get_model_params()`, so `configs.py:80-82` / `utils.py:55-59` work unchanged.
"""
import numpy as np
import torch
from torch import nn

# undirected edges of one 16-slot part window [3 head | 3 arm | 10 hand]
# (reference model_params.py:261-287; its four per-window lists are identical)
_PART_EDGES = [(0, 1), (0, 2), (0, 3), (3, 4), (4, 5), (5, 6), (6, 7), (6, 8), (8, 9), (8, 10),
               (6, 10), (10, 11), (10, 12), (6, 12), (12, 13), (12, 14), (14, 15), (6, 14),
               (7, 9), (9, 11), (11, 13), (13, 15), (7, 15), (7, 11), (7, 13)]


class HWGATEParams:
    def __init__(self, dataset_params, input_dim, device=None, num_kps=64, embed_dim=128):
        self.kp_dim = input_dim
        self.num_kps = num_kps                      # nW * 16; reference default 64
        self.temporal_dim = dataset_params['src_len']
        self.num_classes = dataset_params['num_class']
        self.embed_dim = embed_dim
        self.temporal_patch_size = 2
        self.pe = True
        self.depths = [2, 2, 4]
        self.num_heads = [2, 4, 8]
        self.window_size = 16
        self.drop_rate = 0.1
        self.attn_drop_rate = 0.0
        self.ff_ratio = 2.
        self.norm_layer = nn.LayerNorm
        self.device = device
        self.edges = [[list(e) for e in _PART_EDGES] for _ in range(self.num_kps // self.window_size)]
        self.adj_mat = torch.tensor(self.get_adj_mat(), dtype=torch.float32)

    def get_adj(self, index):
        """I + symmetric part graph of window `index` (W x W)."""
        a = np.eye(self.window_size)
        e = np.asarray(self.edges[index])
        a[e[:, 0], e[:, 1]] = 1
        a[e[:, 1], e[:, 0]] = 1
        return a

    def get_adj_mat(self):
        """(nW, TP*W, TP*W): same frame -> part graph, neighbouring frame ->
        same joint only, further frames -> nothing."""
        TP, W = self.temporal_patch_size, self.window_size
        frame_gap = np.abs(np.arange(TP)[:, None] - np.arange(TP)[None, :])
        out = []
        for w in range(self.num_kps // W):
            blocks = np.where(frame_gap[:, :, None, None] == 0, self.get_adj(w),
                              np.where(frame_gap[:, :, None, None] == 1, np.eye(W), 0.0))
            out.append(blocks.transpose(0, 2, 1, 3).reshape(TP * W, TP * W))
        return np.array(out)

    def get_model_params(self):
        return (self.kp_dim, self.num_kps, self.temporal_dim, self.num_classes, self.embed_dim,
                self.temporal_patch_size, self.pe, self.depths, self.num_heads, self.window_size,
                self.adj_mat, self.drop_rate, self.attn_drop_rate, self.ff_ratio, self.norm_layer,
                self.device)


# 29-joint skeleton of the reference's HGATEParams (model_params.py:424-457): 9 upper-body joints and
# two 10-joint hands with the same internal graph, rooted at joints 9 / 19 and hung from wrists 7 / 8.
_BODY_EDGES = [(2, 0), (1, 0), (0, 3), (0, 4), (3, 5), (4, 6), (5, 7), (6, 8), (7, 9), (8, 19)]
_HAND_EDGES = [(0, 1), (0, 2), (2, 3), (2, 4), (4, 5), (0, 4), (4, 6), (0, 6), (6, 7), (6, 8), (0, 8), (8, 9)]


class HGATEParams:
    """Drop-in for `HGATEParams` of the reference (model_params.py:405-486): same attributes, defaults and
    positional tuple (no window_size: an attention block is all joints of 2 frames)."""

    def __init__(self, dataset_params, input_dim, device=None, embed_dim=128):
        self.kp_dim = input_dim
        self.num_kps = 29
        self.temporal_dim = dataset_params['src_len']
        self.num_classes = dataset_params['num_class']
        self.embed_dim = embed_dim
        self.temporal_patch_size = 2
        self.pe = True
        self.depths = [2, 2, 4]
        self.num_heads = [2, 4, 8]
        self.drop_rate = 0.1
        self.attn_drop_rate = 0.0
        self.ff_ratio = 2.
        self.norm_layer = nn.LayerNorm
        self.device = device
        self.edges = [[list(e) for e in _BODY_EDGES]
                      + [[r + a, r + b] for r in (9, 19) for a, b in _HAND_EDGES]]
        self.adj_mat = torch.tensor(self.get_adj_mat(), dtype=torch.float32)

    def get_adj(self):
        """I + symmetric joint graph (K x K)."""
        a = np.eye(self.num_kps)
        e = np.asarray(self.edges[0])
        a[e[:, 0], e[:, 1]] = 1
        a[e[:, 1], e[:, 0]] = 1
        return a

    def get_adj_mat(self):
        """(TP*K, TP*K): same frame -> joint graph, neighbouring frame -> same joint only."""
        TP, K = self.temporal_patch_size, self.num_kps
        gap = np.abs(np.arange(TP)[:, None] - np.arange(TP)[None, :])
        blocks = np.where(gap[:, :, None, None] == 0, self.get_adj(),
                          np.where(gap[:, :, None, None] == 1, np.eye(K), 0.0))
        return blocks.transpose(0, 2, 1, 3).reshape(TP * K, TP * K)

    def get_model_params(self):
        return (self.kp_dim, self.num_kps, self.temporal_dim, self.num_classes, self.embed_dim,
                self.temporal_patch_size, self.pe, self.depths, self.num_heads, self.adj_mat,
                self.drop_rate, self.attn_drop_rate, self.ff_ratio, self.norm_layer, self.device)


class WGATEParams:
    """Drop-in for `WGATEParams` of the reference (model_params.py:80-241): same attributes, defaults and
    positional tuple.  The adjacency is the dense (nW, T*16, T*16) block-tridiagonal matrix the
    reference builds (same frame -> part graph, neighbouring frame -> same joint)."""

    def __init__(self, dataset_params, input_dim, device=None, num_kps=64, embed_dim=128):
        self.kp_dim = input_dim
        self.num_kps = num_kps
        self.temporal_dim = dataset_params['src_len']
        self.num_classes = dataset_params['num_class']
        self.embed_dim = embed_dim
        self.pe = True
        self.depths = 8
        self.num_heads = 8
        self.window_size = 16
        self.drop_rate = 0.1
        self.attn_drop_rate = 0.0
        self.ff_ratio = 2.
        self.norm_layer = nn.LayerNorm
        self.kp_norm = True
        self.device = device
        self.edges = [[list(e) for e in _PART_EDGES] for _ in range(self.num_kps // self.window_size)]
        self.adj_mat = torch.tensor(self.get_adj_mat(), dtype=torch.float32)

    def get_adj(self, index):
        a = np.eye(self.window_size)
        e = np.asarray(self.edges[index])
        a[e[:, 0], e[:, 1]] = 1
        a[e[:, 1], e[:, 0]] = 1
        return a

    def get_adj_mat(self):
        F, W, K = self.temporal_dim, self.window_size, self.num_kps
        gap = np.abs(np.arange(F)[:, None] - np.arange(F)[None, :])
        out = []
        for w in range(K // W):
            blocks = np.where(gap[:, :, None, None] == 0, self.get_adj(w),
                              np.where(gap[:, :, None, None] == 1, np.eye(W), 0.0))
            out.append(blocks.transpose(0, 2, 1, 3).reshape(F * W, F * W))
        return np.array(out)

    def get_model_params(self):
        return (self.kp_dim, self.num_kps, self.temporal_dim, self.num_classes, self.embed_dim, self.pe,
                self.depths, self.num_heads, self.window_size, self.ff_ratio, self.adj_mat, self.drop_rate,
                self.attn_drop_rate, self.norm_layer, self.device)
