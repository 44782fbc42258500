"""The C-ABI shared library loads without a GPU and exports every entry point include/adm.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "adm.h")
LIBS = {
    "hip": os.path.join(ROOT, "audio-diffusion_amd", "audiodiffusion", "libadm_hip.so"),
    "emu": os.path.join(ROOT, "tests", "emu", "libadm_emu.so"),
}


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)            # comments mention function names too
    names = re.findall(r"\b(adm_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_header_declares_the_hot_path_entry_points():
    names = declared_symbols()
    for must in ("adm_sample_loop", "adm_unet_forward", "adm_sched_step", "adm_mel_forward", "adm_mel_inverse",
                 "adm_vae_decode", "adm_unet_forward_backward", "adm_adamw_ema_step", "adm_conv2d", "adm_set_option"):
        assert must in names
    assert len(names) >= 40


@pytest.mark.parametrize("which", ["hip", "emu"])
def test_library_exports_every_declared_symbol(which):
    path = LIBS[which]
    assert os.path.exists(path), f"{path} missing: run `python -c 'import __graft_entry__ as g; g.build()'`"
    lib = ctypes.CDLL(path)
    missing = [n for n in declared_symbols() if not hasattr(lib, n)]
    assert not missing, f"{which} library lacks {missing}"
    lib.adm_version.restype = ctypes.c_int
    assert lib.adm_version() >= 100
    lib.adm_is_device_build.restype = ctypes.c_int
    assert lib.adm_is_device_build() == (1 if which == "hip" else 0)
