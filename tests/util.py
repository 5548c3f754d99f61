import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def linf_peak(a, b):
    """Per batch item: max|a-b| / max|b| (the 'L-inf / peak' metric of SURVEY.md / BASELINE.md)."""
    a = np.asarray(a, np.float64).reshape(a.shape[0], -1)
    b = np.asarray(b, np.float64).reshape(b.shape[0], -1)
    return np.abs(a - b).max(1) / np.maximum(np.abs(b).max(1), 1e-300)


def rel_l2(a, b):
    a = np.asarray(a, np.float64).reshape(a.shape[0], -1)
    b = np.asarray(b, np.float64).reshape(b.shape[0], -1)
    return np.sqrt(((a - b) ** 2).sum(1) / np.maximum((b ** 2).sum(1), 1e-300))
