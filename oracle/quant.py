"""Test infrastructure: an independent restatement of per-out-channel e4m3 weight quantisation (BASELINE configs[4]), used to
cross-check canonswap_amd.pack.quantize_conv_weights_e4m3 and to run the fp32 oracle on the quantised network."""
from __future__ import annotations

import numpy as np

# every non-negative OCP e4m3fn value: subnormals k * 2^-9 (k = 0..7), normals (1 + m/8) * 2^(e-7) for e = 1..15, without the NaN code
_GRID = np.array(sorted({k * 2.0 ** -9 for k in range(8)} |
                        {(1 + m / 8) * 2.0 ** (e - 7) for e in range(1, 16) for m in range(8) if not (e == 15 and m == 7)}))


def nearest_e4m3(x):
    """Nearest grid value by search (ties to the even code), saturating at 448."""
    x = np.asarray(x, np.float64)
    ax = np.minimum(np.abs(x), 448.0)
    hi = np.clip(np.searchsorted(_GRID, ax), 1, len(_GRID) - 1)
    lo = hi - 1
    dl, dh = ax - _GRID[lo], _GRID[hi] - ax
    pick_hi = (dh < dl) | ((dh == dl) & (hi % 2 == 0))         # codes are ordered like the grid: even index = even mantissa
    return np.sign(x) * np.where(pick_hi, _GRID[hi], _GRID[lo])


def quantize_rows(w):
    rows = np.asarray(w, np.float64).reshape(w.shape[0], -1)
    scale = np.abs(rows).max(axis=1) / 448.0
    scale[scale == 0] = 1.0
    return (nearest_e4m3(rows / scale[:, None]) * scale[:, None]).reshape(w.shape).astype(np.float32)
