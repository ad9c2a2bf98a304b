"""Helpers shared by the CPU (oracle) and GPU (HIP) golden tests."""
import ast
import os
import types
from collections import OrderedDict

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RATIOS_Q2 = np.array([0.9479, 0.0521], dtype=np.float64)

CASE_NAMES = ["c1_gauss", "c1_cat_w1", "c1_cat_wr", "gauss_mmd10", "gauss_norsamp", "gauss_z128",
              "gauss_z512", "gauss_s32", "gauss_s28", "cat_s56", "gauss_rgb"]


def case_inputs(O, cfg, seed):
    """labels (N * in_ch frames of i.i.d. Bernoulli pixels), the normalised image (N, in_ch, S, S) and the loss target of a golden case
    (oracle/make_golden.py run_case)."""
    in_ch = cfg.get("in_ch", 1)
    labels = O.synthetic_labels(cfg["N"] * in_ch, cfg["S"], seed=seed)
    image = O.normalise(labels, cfg["S"]).view(cfg["N"], in_ch, cfg["S"], cfg["S"])
    categorical = cfg["out_ch"] > in_ch
    return labels, image, categorical, (labels if categorical else image)
TRAJ_NAMES = ["traj_gauss", "traj_cat"]


def load(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    cfg = ast.literal_eval(str(g["cfg"]))
    return g, cfg


def class_weight(cfg, device="cpu"):
    if cfg["weight"] is None:
        return None
    if cfg["weight"] == "ones":
        return torch.FloatTensor([1] * cfg["out_ch"]).to(device)
    return torch.FloatTensor(1 - RATIOS_Q2).to(device)


def make_args(cfg, device="cpu"):
    return types.SimpleNamespace(data_ratio_of_labels=class_weight(cfg, device), dataset="MovingMNIST", quiet=True)


class LabelLoader:
    def __init__(self, O, n, size, steps, seed0):
        self.b = [O.synthetic_labels(n, size, seed=seed0 + i).view(n, size * size) for i in range(steps)]

    def __iter__(self):
        return iter(self.b)
