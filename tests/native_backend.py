"""Test helper: selects which build of the kernel sources a test drives.

"emu" -> tests/emu/libadm_emu.so : the SAME kernel sources compiled by g++ against the fiber emulator
          (tests/emu/hip_emu.h); runs in the GPU-less container; tensors live on the CPU.
"hip" -> audio-diffusion_amd/audiodiffusion/libadm_hip.so : the product library on a real MI355X (-m gpu).
"""
import os
import subprocess

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_LIB = os.path.join(ROOT, "tests", "emu", "libadm_emu.so")
BUILD = os.path.join(ROOT, "audio-diffusion_amd", "csrc", "build.sh")

_built = False


def ensure_emu_built():
    global _built
    if not _built and os.environ.get("ADM_EMU_NOBUILD") != "1":    # spawned rank workers: the parent has built it already
        subprocess.run(["bash", BUILD, "emu"], check=True, capture_output=True)
        os.environ["ADM_EMU_NOBUILD"] = "1"                         # inherited by multiprocessing children
    _built = True
    return EMU_LIB


def select(backend):
    """Loads the library for `backend` and returns the torch device tensors must live on."""
    from audiodiffusion import _native
    if backend == "emu":
        _native.load(ensure_emu_built())
        return torch.device("cpu")
    assert torch.cuda.is_available(), "gpu tests need a MI355X"
    _native.load()  # product library; raises loudly if missing
    assert _native.is_device_build()
    return torch.device("cuda:0")


BACKENDS = [pytest.param("emu", id="emu"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]
