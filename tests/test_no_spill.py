"""Static check of the gfx950 code of the convolution kernels: a register spill inside their loops turns every scratch reload into a
`s_waitcnt vmcnt(0)` that drains the prefetch pipeline, and the allocator has proved fragile — so the
measured default kernels must compile spill-free, and the opt-in ones within the bounds written down in DESIGN.md."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "audio-diffusion_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _usage(src):
    # (the flags of csrc/build.sh: k_conv_wino*.hip are built without SLP vectorisation)
    extra = ["-fno-slp-vectorize"] if src.startswith("k_conv_wino") else []
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + extra + ["-c", src, "-o", os.devnull,
                        "-Rpass-analysis=kernel-resource-usage"], cwd=CSRC, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out, name = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            out[name] = {}
        m = re.search(r"(ScratchSize \[bytes/lane\]|VGPRs Spill|VGPRs|AGPRs): (\d+)", line)
        if m and name:
            out[name][m.group(1).split(" ")[0]] = int(m.group(2))
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.parametrize("src,max_scratch", [
    ("k_conv_wino_f4.hip", 0),           # the headline kernel: conv_wino6_kernel, Winograd F(4x4,3x3)
    ("k_conv_wino_f2.hip", 0),           # F(2x2,3x3): conv_wino5_kernel, conv_wino4_kernel
    ("k_conv_wino.hip", 0),              # the filter packers
    ("k_conv_wgrad.hip:sp8|sp_kernel|pf_kernelILi3ELb1|pf_kernelILi1ELb1", 0),   # fp32 weight gradients on the fast paths (the
                                         # generic 1x1 fallback `pf_kernel<1, false>` is known to spill; it serves odd chunkings only)
    ("k_conv_bf16.hip", 0),              # measured defaults: forward / data gradient, weight gradient, packing
    ("k_conv_bf16b.hip", 0),             # round 4: blocked-image forward / data gradient (2 workgroups per CU) and weight gradient
    ("k_conv1x1_bf16.hip", 0),
])
def test_conv_kernels_compile_without_spills(src, max_scratch):
    src, _, only = src.partition(":")
    usage = _usage(src)
    kernels = {k: v for k, v in usage.items() if "kernel" in k and (not only or re.search(only, k))}
    assert kernels, usage
    worst = max(v.get("ScratchSize", 0) for v in kernels.values())
    assert worst <= max_scratch, {k: v for k, v in kernels.items() if v.get("ScratchSize", 0) > max_scratch}
