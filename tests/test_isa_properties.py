"""Properties of the gfx950 code the performance claims rest on, checked on the assembly hipcc emits (no GPU needed;
one extra compilation of csrc/dw_api.hip with --save-temps, ~25 s):

  * no MFMA anywhere (north star: this path is a stencil, not a dense contraction);
  * the step kernels' hot loops hold no scratch (spill) traffic;
  * register budgets: every float32 step kernel fits 128 VGPRs (4 waves/SIMD), every exact one 168 (3 waves/SIMD) -
    except the packed / STATS exact fused variants, planned for 2;
  * the hot loops are made of packed float32 arithmetic (v_pk_*), 6 transcendentals per cell-evaluation.
"""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def asm():
    import shutil
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available")
    import isa_report
    return open(isa_report.build([])).read()


def _kernels(text):
    """name -> (info dict, body text) for every kernel of the module."""
    out = {}
    for name in re.findall(r"\.amdhsa_kernel (\S+)\n", text):
        m = re.search(r"\n" + re.escape(name) + r":[^\n]*\n(.*?)\n\.Lfunc_end", text, re.S)
        info = re.search(re.escape(name) + r":.*?; Kernel info:(.*?)(?=\n\t\.(?:text|section)|\Z)", text, re.S)
        if not (m and info):
            continue
        vals = {k: int(v) for k, v in re.findall(r"; (\w+)\s*[:=] (\d+)", info.group(1))}
        out[name] = (vals, m.group(1))
    return out


def _hot_loop(body):
    lines = body.split("\n")
    labels = {m.group(1): i for i, ln in enumerate(lines) for m in [re.match(r"(\.LBB\S+):", ln)] if m}
    best, best_pk = None, -1
    for i, ln in enumerate(lines):
        m = re.match(r"\ts_c?branch\S* (\.LBB\S+)", ln)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            npk = sum(1 for x in lines[labels[m.group(1)]:i + 1] if x.startswith("\tv_pk_"))
            if npk > best_pk:
                best, best_pk = (labels[m.group(1)], i), npk
    return lines[best[0]:best[1] + 1] if best else []


def _main_path(loop):
    """The loop without its queue-push segments: split at labels and branches, drop segments with v_mbcnt / ds_write_b128."""
    segs, cur = [], []
    for ln in loop:
        if re.match(r"(\.LBB\S+):", ln):
            segs.append(cur)
            cur = []
        cur.append(ln)
        if re.match(r"\ts_c?branch", ln):
            segs.append(cur)
            cur = []
    segs.append(cur)
    return [ln for sg in segs if not any("v_mbcnt" in x or "ds_write_b128" in x for x in sg) for ln in sg]


def test_no_mfma_and_no_cuda_shims(asm):
    assert not re.search(r"\tv_mfma", asm)
    assert not re.search(r"\tv_smfma|\tv_wmma", asm)


def test_step_kernels_register_budgets_and_clean_hot_loops(asm):
    ks = _kernels(asm)
    step = {n: v for n, v in ks.items() if "step_stream" in n}
    assert len(step) >= 20
    for name, (info, body) in step.items():
        exact = "exact" in name
        fused = "fused2" in name
        loop = _hot_loop(body)
        assert loop, name
        # no scratch traffic on the row loop's main path (segments that push a near-tie cell into the LDS queue -
        # they hold v_mbcnt / ds_write_b128 and run for well under 1 % of the row maps - may reload a spilled value)
        assert not any(ln.startswith("\tscratch_") for ln in _main_path(loop)), f"{name}: scratch traffic inside the row loop"
        npk = sum(1 for ln in loop if ln.startswith("\tv_pk_"))
        ntr = sum(1 for ln in loop if re.match(r"\tv_(sqrt|rcp)_f32", ln))
        assert npk >= 150 and ntr >= 48, (name, npk, ntr)     # the map is packed float32 + 6 transcendentals per cell
        if not exact:
            # 3 waves/SIMD by design: the ring variant with per-step world flags and the packed fused variants (no
            # measurable difference to 4 on 65536 x 16^2 ... 2048 x 128^2, and no scratch at all)
            three = "fused2ILi2ELb0ELb1E" in name or "fused2ILi1ELb1E" in name
            assert info["Occupancy"] >= (3 if three else 4), (name, info["NumVgprs"])
        else:
            plain_fused = any(f"fused2_exactILi{m}ELb0ELb0E" in name for m in (0, 1, 2))   # not packed, no STATS
            stats_fused = any(f"fused2_exactILi{m}ELb0ELb1E" in name for m in (0, 1))      # STATS, not the ring
            if "step_stream_exactILi" in name or plain_fused or stats_fused:
                assert info["NumVgprs"] <= 168 and info["Occupancy"] >= 3, (name, info["NumVgprs"])
            else:
                assert info["NumVgprs"] <= 256 and info["Occupancy"] >= 2, (name, info["NumVgprs"])


def test_fused_float32_kernels_instruction_budget(asm):
    """The float32 step-pair kernels are VALU-issue-bound: their speed IS their instruction count.
    One iteration of the hot loop = 3 rows x 4 columns x 2 steps = 24 cell-evaluations per lane; budget: 33 VALU
    instructions per cell-evaluation for the plain variants (round 2 ends at 32.4-33.0: 18 packed + 6 transcendental
    + 4 clamp/round + 1.5 conversions + neighbour sums), with every edge-column sum folded into v_add_f32_dpp where
    the neighbour needs no `old` value."""
    ks = _kernels(asm)
    for mode in (0, 1):                                    # overlapped strips, in-wave rotation (W = 256)
        name = next(n for n in ks if f"step_stream_fused2ILi{mode}ELb0ELb0E" in n)
        loop = _hot_loop(ks[name][1])
        valu = [ln for ln in loop if ln.startswith("\tv_")]
        trans = [ln for ln in valu if re.match(r"\tv_(sqrt|rcp)_f32", ln)]
        assert len(trans) == 144, (name, len(trans))        # 24 cell-evaluations x 6
        assert len(valu) <= 33 * 24, (name, len(valu))
        assert not any(ln.startswith("\tv_mov_b32_dpp") for ln in loop), name


def test_first_step_stream_kernels_are_packed_and_spill_free(asm):
    """step_first_stream (the first step of an episode on every shape the wave-strip kernels take): packed float32 in its
    row loop, no scratch anywhere in the kernel, float32-only variants at >= 4 waves/SIMD, the bounded exact ones at >= 3."""
    ks = {n: v for n, v in _kernels(asm).items() if "step_first_stream" in n}
    assert len(ks) == 16                                      # float / double input x float32-only / bounded x HALO 0 .. 3
    for name, (info, body) in ks.items():
        assert not re.search(r"\tscratch_", body), name
        prec, halo = (int(x) for x in re.search(r"step_first_streamI[fd]Li(\d)ELi(\d)E", name).groups())
        bounded = prec == 3
        assert info["Occupancy"] >= (3 if bounded else 4), (name, info["NumVgprs"])
        loop = _hot_loop(body)
        npk = sum(1 for ln in loop if ln.startswith("\tv_pk_"))
        ntr = sum(1 for ln in loop if re.match(r"\tv_(sqrt|rcp)_f32", ln))
        assert npk >= 40 and ntr == (32 if bounded else 24), (name, npk, ntr)   # 4 cells x (6 + the bound's two roots)


def test_episode_wave_step_loop_has_no_scratch_and_few_branches(asm):
    """episode_wave (one wave per world: every instruction of the step costs its lone wave ~5-7 cycles): no scratch access in
    either instantiation (a by-reference struct of the reachable covers once lived there: 56 bytes stored and a dependent
    load per step), the first-to-graze loop is the mask form (one v_readlane + one v_cmp per agent: at most 12 instructions
    per iteration), and the float32 kernel keeps 4 waves/SIMD."""
    ks = {n: v for n, v in _kernels(asm).items() if re.search(r"episode_waveILb[01]EE", n)}
    assert len(ks) == 2
    for name, (info, body) in ks.items():
        assert not re.search(r"\tscratch_", body), name
        lines = body.split("\n")
        labels = {m.group(1): i for i, ln in enumerate(lines) for m in [re.match(r"(\.LBB\S+):", ln)] if m}
        # innermost loops that read a lane: the conflict loop
        loops = []
        for i, ln in enumerate(lines):
            m = re.match(r"\ts_cbranch\S* (\.LBB\S+)", ln)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                seg = [x for x in lines[labels[m.group(1)]:i + 1] if x.startswith("\t") and not x.startswith("\t;")]
                if any(x.startswith("\tv_readlane_b32") for x in seg) and len(seg) < 40:
                    loops.append(seg)
        assert loops, name
        assert min(len(seg) for seg in loops) <= 12, (name, [len(s) for s in loops])
        if "ILb0EE" in name:
            assert info["Occupancy"] >= 4, (name, info["NumVgprs"])
