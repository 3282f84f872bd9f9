"""therldaisyworld_amd — MI355X-native (gfx950) implementation of the RLDaisyWorld grid update.

``RLDaisyWorld`` is a drop-in for ``daisy.daisy_world_rl.RLDaisyWorld`` of
riveSunder/therldaisyworld; the per-cell / per-agent work runs in hand-written HIP kernels behind the
C ABI declared in ``include/daisyworld_hip.h``.  Importing this package does not need a GPU; creating
an environment does (there is no CPU fallback).
"""
from .daisy_world_rl import RLDaisyWorld  # noqa: F401
from ._ffi import DaisyHipError  # noqa: F401
from .engine import Engine, default_params  # noqa: F401
from .agents.greedy import Greedy  # noqa: F401
from .agents.mlp import MLP  # noqa: F401

__all__ = ["RLDaisyWorld", "Engine", "default_params", "Greedy", "MLP", "DaisyHipError"]
