"""Board telemetry beside a running workload: power and engine clock from rocm-smi (bench.py's `"power"` object)."""
from __future__ import annotations

import re
import shutil
import subprocess
import threading
import time


def board_power_while(work, settle_s=0.45, samples=3, device=0):
    """Board power and engine clock WHILE `work()` (a blocking library call: ctypes releases the GIL) runs on a helper
    thread - rocm-smi read beside it, after `settle_s` (the power manager needs a few hundred ms to reach its steady
    state).  bench.py calls it with an UNTIMED extra pass after the timed region.  Failures (no rocm-smi, unexpected output, the
    work over before the first sample) come back as {"board_w": None, "error": ...}, never as an exception."""
    smi = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    err = []

    def guarded():
        try:
            work()
        except Exception as e:                      # (reported in the object instead of a traceback on a helper thread)
            err.append(f"work: {type(e).__name__}: {e}")

    th = threading.Thread(target=guarded)
    got = []
    t_start = time.perf_counter()
    th.start()
    try:
        time.sleep(settle_s)
        while th.is_alive() and len(got) < samples:
            txt = subprocess.run([smi, "-d", str(device), "--showpower", "--showclocks", "--showmaxpower"],
                                 capture_output=True, text=True, timeout=20).stdout
            w = re.search(r"Current Socket Graphics Package Power \(W\): *([0-9.]+)", txt) or \
                re.search(r"Average Graphics Package Power \(W\): *([0-9.]+)", txt)
            c = re.search(r"sclk clock level: *\S+ *\((\d+)Mhz\)", txt)
            mx = re.search(r"Max Graphics Package Power \(W\): *([0-9.]+)", txt)
            if w and c and th.is_alive():           # (a sample that ended after the work did is not one of it)
                got.append((float(w.group(1)), int(c.group(1)), float(mx.group(1)) if mx else None))
    except Exception as e:
        err.append(f"rocm-smi: {type(e).__name__}: {e}")
    th.join()
    if not got:
        return {"board_w": None, "error": "; ".join(err) or f"no sample while the work ran ({time.perf_counter() - t_start:.2f} s)"}
    return {"board_w": max(g[0] for g in got), "limit_w": got[0][2], "sclk_mhz": sorted(g[1] for g in got)[len(got) // 2],
            "samples": [[g[0], g[1]] for g in got], "busy_s": round(time.perf_counter() - t_start, 2),
            "source": "rocm-smi --showpower --showclocks beside an untimed extra pass of the same launches"}

