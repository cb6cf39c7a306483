"""Guard of k_sweep's register allocation (compile only: hipcc cross-compiles gfx950 without a GPU).

The persistent sweep kernel sits exactly at its 256-VGPR budget (two waves per SIMD: eight role waves per workgroup, one workgroup per CU).
DESIGN.md 8 (round 4) measured that an unrelated source change moved its spill count from 21-24 to 13 VGPRs and the benchmark by 6 %: what
the phases of a step cost is as much a question of what the allocator keeps in registers across the role loops as of the algorithm.  This
test makes such a change visible in `pytest -m "not gpu"` instead of on the next benchmark: it compiles the kernel's translation unit for
the device only, with the product's flags (the Makefile's own SWEEPFLAGS), and reads `-Rpass-analysis=kernel-resource-usage`."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "stan4bart_amd", "csrc")
MAX_SPILLED_VGPRS = 2        # product build since round 5: 0 (round 4: 14, worth 3 - 6 % of the sweep; a second inlined copy of the drawing code: 20)
MAX_SCRATCH_BYTES = 1880     # product build since round 5: 1848 bytes per lane (the hand-over tail's global-memory control step, not the step loops)


def _usage(extra=()):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        pytest.skip("hipcc not found")
    mk = open(os.path.join(CSRC, "Makefile")).read()
    cxx = re.search(r"^CXXFLAGS \?= (.*)$", mk, re.M).group(1).split()
    swp = re.search(r"^SWEEPFLAGS \?= (.*)$", mk, re.M).group(1).split()
    cmd = [hipcc, "--offload-arch=gfx950", *cxx, *swp, *extra, "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-c", "-o", os.devnull, "dev_sweep.hip"]
    out = subprocess.run(cmd, cwd=CSRC, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:]
    blocks = re.split(r"remark: Function Name: ", out.stdout)[1:]
    out = {}
    for kernel, key in (("7k_sweepE", "k_sweep"), ("14k_sweep_streamE", "k_sweep_stream"), ("11k_sweep_fewE", "k_sweep_few"),
                        ("10k_sweep_spE", "k_sweep_sp"), ("14k_sweep_few_spE", "k_sweep_few_sp"), ("9k_sweep_wE", "k_sweep_w")):      # (mangled names: s4b::k_sweep, ...)
        hit = [b for b in blocks if kernel in b.split()[0]]
        assert len(hit) == 1, [b.split()[0] for b in blocks]

        def field(name, text=hit[0]):
            return int(re.search(name + r": (\d+)", text).group(1))
        out[key] = dict(vgprs=field("VGPRs"), spill=field("VGPRs Spill"), scratch=field(r"ScratchSize \[bytes/lane\]"), occupancy=field(r"Occupancy \[waves/SIMD\]"),
                        lds=field(r"LDS Size \[bytes/block\]"))
    return out


def test_k_sweep_register_allocation_is_the_one_that_was_measured():
    both = _usage()
    u = both["k_sweep"]
    assert u["vgprs"] <= 256 and u["occupancy"] >= 2, u          # eight waves of one workgroup must fit a CU
    assert u["spill"] <= MAX_SPILLED_VGPRS, f"k_sweep now spills {u['spill']} VGPRs (measured build: 0 since round 5, 14 in round 4; 21-24 cost 6 % of the benchmark): {u}"
    assert u["scratch"] <= MAX_SCRATCH_BYTES, f"k_sweep's scratch grew to {u['scratch']} bytes per lane (measured build: 1848): {u}"
    # the streaming variant (n > 1.04e6): its pass keeps two batches of observations in flight per thread — no spill inside that loop
    v = both["k_sweep_stream"]
    assert v["vgprs"] <= 256 and v["occupancy"] >= 2 and v["spill"] <= MAX_SPILLED_VGPRS and v["scratch"] <= MAX_SCRATCH_BYTES, v      # (round 5: 0)
    # the launch for few observations per thread (same body, statistics with the missing quads left out)
    w = both["k_sweep_few"]
    assert w["vgprs"] <= 256 and w["occupancy"] >= 2 and w["spill"] <= MAX_SPILLED_VGPRS and w["scratch"] <= MAX_SCRATCH_BYTES, w
    # cgm(split.probs): the same two launches with the weighted predictor choice in the wave-register control code (+ 2 KiB of LDS: the table of the weights)
    for key in ("k_sweep_sp", "k_sweep_few_sp"):
        z = both[key]
        assert z["vgprs"] <= 256 and z["occupancy"] >= 2 and z["spill"] <= MAX_SPILLED_VGPRS and z["scratch"] <= MAX_SCRATCH_BYTES, (key, z)
        assert z["lds"] <= u["lds"] + 2072, (key, z, u)
    # observation weights: the instantiation with the skippable quads (without them: 878 spilled registers), + 32 KiB of LDS for the workgroup's weights
    y = both["k_sweep_w"]
    assert y["vgprs"] <= 256 and y["occupancy"] >= 2 and y["spill"] <= MAX_SPILLED_VGPRS and y["scratch"] <= MAX_SCRATCH_BYTES, y
    assert y["lds"] <= u["lds"] + 32768 + 64, (y, u)
