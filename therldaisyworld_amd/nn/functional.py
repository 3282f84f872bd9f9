"""Observation-neighbourhood masks (ref: daisy/nn/functional.py:51-103).

Only the masks live here: the reference's FFT convolution (``ft_convolve``, :12-49) is replaced by
the LDS-tiled 9-tap toroidal stencil inside the HIP step kernel (csrc/dw_step_*.hpp), and
``glorot`` belongs to the out-of-scope MLP policy.
"""
import numpy as np


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
