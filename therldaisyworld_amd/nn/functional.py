"""``daisy.nn.functional`` of the reference on the MI355X path: the toroidal 3x3 convolution and the
observation-neighbourhood masks (ref: daisy/nn/functional.py).

``ft_convolve`` (ref :12-49) is the reference's FFT convolution; it is *exactly* the toroidal true convolution
``out[i, j] = sum_ab k[a, b] * x[i - (a - 1), j - (b - 1)]`` (orientation pinned by fixture G6 with
an asymmetric kernel).  Here it is one float64 device call (``dw_conv3x3_f64``: nine taps per cell, no FFT).  The step
kernels never call it: inside ``forward()`` the five convolutions of the reference are fused into the wave-strip kernels'
register window (neighbours by DPP, csrc/dw_step_stream.hpp / dw_step_fused.hpp) or the LDS halo tile of the narrow-grid
kernel (csrc/dw_step_tiled.hpp).  ``glorot`` belongs to the MLP policy (agents/mlp.py).
"""
from collections import OrderedDict

import numpy as np


_engines = OrderedDict()                      # (planes, H, W) -> Engine; a few shapes stay resident
_MAX_ENGINES = 4


def _engine_for(shape):
    from ..engine import Engine, default_params
    eng = _engines.pop(shape, None)
    if eng is None:
        while len(_engines) >= _MAX_ENGINES:
            _engines.popitem(last=False)[1].close()
        eng = Engine(default_params(shape[0], shape[1], shape[2], 0))      # raises without a gfx950 device: no CPU path
    _engines[shape] = eng
    return eng


def ft_convolve(grid, kernel):
    """Toroidal convolution of ``grid`` (B, C, H, W) with a 3x3 ``kernel`` (1, 1, 3, 3) -> (B, C, H, W) float64
    (ref daisy/nn/functional.py:12-49; callers: calculate_albedo :377-394, calculate_daisy_density :423-432).

    Same signature and result as the reference (within a few 1e-16 relative: the reference's FFT round-off), evaluated on
    the device.  Differences, all towards accepting more: any H, W >= 3 works (the reference divides by zero for 3 and 4
    and fails on a shape error for 6); a kernel of the grid's own shape (the reference's un-padded branch, which none of
    its callers takes) is refused.
    """
    x = np.asarray(grid, dtype=np.float64)
    k = np.asarray(kernel, dtype=np.float64)
    if x.ndim != 4:
        raise ValueError(f"grid must have shape (B, C, H, W), got {x.shape}")
    if k.shape[-2:] != (3, 3) or k.size != 9:
        raise ValueError(f"only 3x3 kernels of shape (1, 1, 3, 3) are supported, got {k.shape}")
    B, Cn, H, W = x.shape
    if H < 3 or W < 3:
        raise ValueError("grid must be at least 3x3")
    planes = np.ascontiguousarray(x.reshape(B * Cn, H, W))
    out = _engine_for((B * Cn, H, W)).conv3x3(planes, k.reshape(3, 3))
    return out.reshape(B, Cn, H, W)


def _ball(radius, metric):
    ax = np.arange(-radius, radius + 1)
    cc, rr = np.meshgrid(ax, ax)
    return (metric(np.abs(cc), np.abs(rr)) <= radius).astype(np.float64)


def make_von_neumann(radius=1):
    """L1 ball (ref :51-63)."""
    return _ball(radius, lambda a, b: a + b)


def make_moore(radius=1):
    """L-infinity ball (ref :65-77)."""
    return _ball(radius, np.maximum)


def make_circular(radius=1):
    """L2 ball (ref :79-90)."""
    return _ball(radius, lambda a, b: np.sqrt(a ** 2 + b ** 2))


def make_neighborhood(radius=1, mode="moore"):
    """ref :93-103 — unknown modes fall back to von Neumann with the same warning."""
    if mode == "moore":
        return make_moore(radius)
    if mode == "von_neumann":
        return make_von_neumann(radius)
    if mode == "circular":
        return make_circular(radius)
    print(f"neighborhood mode {mode} not recognized, using von Neumann default")
    return make_von_neumann(radius)
