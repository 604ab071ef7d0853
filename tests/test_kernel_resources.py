"""The register-resident design of the chain kernels only holds while nothing spills: compile the kernel files with
-Rpass-analysis=kernel-resource-usage (hipcc cross-compiles without a GPU) and assert that no chain / weight-gradient kernel of the
fp32 hot path uses scratch memory.  (tools/kernel_resources.sh prints the same table; profiles/r03_kernel_resources.txt is its output.)"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "tiny-nerf-pytorch_amd", "csrc")
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-result".split()


def resources(src):
    r = subprocess.run(["hipcc", *FLAGS, "-c", src, "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"], cwd=CSRC,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out, cur = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
        for key, pat in (("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("vgpr_spill", r"VGPRs Spill: (\d+)"), ("vgprs", r" VGPRs: (\d+)"),
                         ("agprs", r"AGPRs: (\d+)"), ("occupancy", r"Occupancy \[waves/SIMD\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    return out


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
@pytest.mark.parametrize("src,prefixes", [("mlpx3.hip", ("k_renderx3", "k_dgradx3", "k_mlpx3_fwd", "k_mlpx3_bwd", "k_tilex3_fwd", "k_tilex3_bwd", "k_compx3")),
                                          ("wgrad.hip", ("k_wgrad", "k_finish")),
                                          ("mlp_fwd.hip", ("k_render_fused", "k_mlp_fwd")), ("mlp_bwd.hip", ("k_train_bwd", "k_mlp_bwd"))])
def test_hot_path_kernels_use_no_scratch(src, prefixes):
    res = resources(src)
    hit = {n: r for n, r in res.items() if any(p in n for p in prefixes)}
    assert len(hit) >= len(prefixes), sorted(res)
    for name, r in hit.items():
        assert r["scratch"] == 0, (name, r)                    # nothing lives in memory that was meant to live in registers
        # "VGPRs Spill" also counts values the allocator parks in a FREE accumulator register (v_accvgpr_write / read, no memory):
        # the 256-wide dgrad kernels keep one loop-invariant address there (the fp32-MFMA ones two).  Anything beyond that would be real pressure.
        assert r["vgpr_spill"] <= 2, (name, r)


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
@pytest.mark.parametrize("src", ["mlp16_fwd.hip", "mlp16_bwd.hip"])
def test_bf16_chain_kernels_spill_budget(src):
    """bf16 mode (two waves per SIMD, 256 registers): inference and dgrad kernels use no scratch; the 256-wide TRAINING forward still
    parks 3 kernel-lifetime values in memory (stored once in the prologue, read back three times; round 3: 12) — pinned here so that
    it cannot grow unnoticed."""
    res = resources(src)
    hit = {n: r for n, r in res.items() if "k_render16" in n or "k_dgrad16" in n or "k_wgrad16" in n}
    assert hit, sorted(res)
    for name, r in hit.items():
        train_fwd_256 = "k_render16ILi256ELb1" in name
        assert r["vgpr_spill"] <= (3 if train_fwd_256 else 0), (name, r)
        assert r["scratch"] <= (16 if train_fwd_256 else 0), (name, r)


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
@pytest.mark.parametrize("src", ["mlpx3.hip"])
def test_compiler_emits_no_m0_use_of_its_own(src):
    """The x3 chain kernels write M0 once per weight-stream stage and issue the stage's LDS-DMA pieces without touching it again
    (mlpx3_core.hpp tx_m0_set / tx_issue_piece_m0; every other DMA saves and restores M0): valid only while hipcc itself never reads or
    writes M0 in these kernels.  Every M0 reference in the ISA must sit inside an inline-asm block."""
    r = subprocess.run(["hipcc", *FLAGS, "-S", "--cuda-device-only", src, "-o", "-"], cwd=CSRC, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    in_asm, n_asm, bad = False, 0, []
    for line in r.stdout.splitlines():
        s = line.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True
        elif s.startswith(";;#ASMEND"):
            in_asm = False
        elif s and not s.startswith((";", ".", "#")) and re.search(r"\bm0\b", s):
            if in_asm:
                n_asm += 1
            else:
                bad.append(s)
    assert n_asm > 0, "no LDS-DMA found: the check looks at the wrong thing"
    assert not bad, bad[:5]
