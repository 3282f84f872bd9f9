"""Build libdaisyworld_hip.so (the gfx950 kernels + C ABI) in-tree with hipcc.

hipcc cross-compiles for gfx950 without a GPU, so this runs in the build container; the resulting
``.so`` is git-ignored but travels with the repo snapshot to the GPU box.
"""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdaisyworld_hip.so")
SOURCES = ["dw_api.hip"]
DEPS = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp"))) + \
       [os.path.join("..", "..", "include", "daisyworld_hip.h")]
# -fno-slp-vectorize: the float32 map is written in packed form by hand (dw_physics.hpp); what the SLP
# vectoriser adds on top are packed adds whose operand pairs have to be assembled with v_mov first (the pair
# sums of a row) - without it the DPP neighbour moves fold into v_add_f32_dpp, the fused kernels lose 5 % of
# their instructions and the exact ones 35 VGPRs.
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra",
         "-Wno-unused-parameter", "-fno-slp-vectorize"]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


TUNING_LIB = os.path.join(HERE, "libdaisyworld_hip_tuning.so")


def build_tuning_library(verbose: bool = False) -> str:
    """-DDW_TUNING build with ablation hooks (tools/kbench.py only; select it with DW_LIB=...)."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, *FLAGS, "-DDW_TUNING", "-o", TUNING_LIB, *[os.path.join(CSRC, s) for s in SOURCES]]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return TUNING_LIB


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP extension if it is missing or older than its sources.  Returns its path."""
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libdaisyworld_hip.so")
    cmd = [hipcc, *FLAGS, "-o", LIB, *[os.path.join(CSRC, s) for s in SOURCES]]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
