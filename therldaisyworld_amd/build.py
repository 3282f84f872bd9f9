"""Build libdaisyworld_hip.so (the gfx950 kernels + C ABI) in-tree with hipcc.

hipcc cross-compiles for gfx950 without a GPU, so this runs in the build container; the resulting
``.so`` is git-ignored but travels with the repo snapshot to the GPU box.

Staleness is decided by CONTENT, not by mtime (a checkout or an rsync does not preserve mtimes): the
library carries a hash of every source it was built from plus the compiler flags (``dw_build_id()``,
also findable in the file as the bytes ``DW_BUILD_ID=<hex>``), and it is rebuilt whenever that hash
differs from the hash of the sources next to it.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
HEADER = os.path.join(HERE, "..", "include", "daisyworld_hip.h")
LIB = os.path.join(HERE, "libdaisyworld_hip.so")
VARIANTS = os.path.join(HERE, "variants")                     # tuning builds (tools/kbench.py, DW_LIB=...)
SOURCES = ["dw_api.hip"]
# -fno-slp-vectorize: the float32 map is written in packed form by hand (dw_physics.hpp); what the SLP
# vectoriser adds on top are packed adds whose operand pairs have to be assembled with v_mov first (the pair
# sums of a row) - without it the DPP neighbour moves fold into v_add_f32_dpp, the fused kernels lose 5 % of
# their instructions and the exact ones 35 VGPRs.
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra",
         "-Wno-unused-parameter", "-fno-slp-vectorize"]
_MARK = b"DW_BUILD_ID="


def _deps():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".hpp"))] + [HEADER]


def source_id(defines=()) -> str:
    """Hash of the sources, the public header, the flags and the variant's -D defines (16 hex digits)."""
    h = hashlib.sha256()
    for path in _deps():
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
    h.update(" ".join([*FLAGS, *sorted(defines)]).encode())
    return h.hexdigest()[:16]


def library_id(path: str) -> str | None:
    """The build id embedded in a built library (None if the file is missing or carries none)."""
    try:
        with open(path, "rb") as f:
            blob = f.read()
    except OSError:
        return None
    i = blob.find(_MARK)
    return None if i < 0 else blob[i + len(_MARK): i + len(_MARK) + 16].decode("ascii", "replace")


def _compile(out: str, defines=(), verbose: bool = False) -> str:
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build " + os.path.basename(out))
    cmd = [hipcc, *FLAGS, *[f"-D{d}" for d in defines], f'-DDW_BUILD_ID="{source_id(defines)}"', "-o", out,
           *[os.path.join(CSRC, s) for s in SOURCES]]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return out


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP extension unless the one in the tree was built from exactly these sources."""
    if force or library_id(LIB) != source_id():
        _compile(LIB, (), verbose)
    return LIB


def build_variant(name: str, defines, force: bool = False, verbose: bool = False) -> str:
    """A tuning build with extra -D defines, e.g. build_variant("split", ["DW_FAST_SPLIT=1"]) ->
    variants/libdaisyworld_hip_split.so (same ABI; select it with Engine(lib_path=...) or DW_LIB)."""
    os.makedirs(VARIANTS, exist_ok=True)
    out = os.path.join(VARIANTS, f"libdaisyworld_hip_{name}.so")
    if force or library_id(out) != source_id(defines):
        _compile(out, tuple(defines), verbose)
    return out


# ---- the optional host helper (gcc; NumPy's legacy random stream in bulk, include/daisyworld_host.h) ---------------
HOST_LIB = os.path.join(HERE, "libdaisyworld_host.so")
HOST_SOURCE = os.path.join(CSRC, "dw_hostrng.c")
HOST_HEADER = os.path.join(HERE, "..", "include", "daisyworld_host.h")
HOST_FLAGS = ["-O3", "-fPIC", "-shared", "-Wall", "-Wextra"]
_HOST_MARK = b"DW_HOST_BUILD_ID="


def host_source_id() -> str:
    h = hashlib.sha256()
    for path in (HOST_SOURCE, HOST_HEADER):
        with open(path, "rb") as f:
            h.update(f.read())
    h.update(" ".join(HOST_FLAGS).encode())
    return h.hexdigest()[:16]


def host_library_id(path: str = HOST_LIB) -> str | None:
    try:
        with open(path, "rb") as f:
            blob = f.read()
    except OSError:
        return None
    i = blob.find(_HOST_MARK)
    return None if i < 0 else blob[i + len(_HOST_MARK): i + len(_HOST_MARK) + 16].decode("ascii", "replace")


def build_host_library(force: bool = False, verbose: bool = False) -> str:
    """Compile libdaisyworld_host.so with gcc unless the one in the tree was built from exactly these sources."""
    if force or host_library_id() != host_source_id():
        gcc = shutil.which("gcc") or shutil.which("cc")
        if not gcc:
            raise RuntimeError("gcc not found: cannot build " + os.path.basename(HOST_LIB))
        cmd = [gcc, *HOST_FLAGS, f'-DDW_HOST_BUILD_ID="{host_source_id()}"', "-o", HOST_LIB, HOST_SOURCE]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=CSRC)
    return HOST_LIB


def build_tuning_library(verbose: bool = False) -> str:
    """-DDW_TUNING build with ablation hooks (tools/kbench.py only)."""
    return build_variant("tuning", ["DW_TUNING"], verbose=verbose)


if __name__ == "__main__":
    import sys
    if len(sys.argv) > 2:                                   # python -m therldaisyworld_amd.build NAME DEFINE...
        print(build_variant(sys.argv[1], sys.argv[2:], force=True, verbose=True))
    else:
        print(build_library(force=True, verbose=True))
        print(build_host_library(force=True, verbose=True))
