"""A/B of two builds of the native library on the SAME box (box-to-box clock spread is +-2 %, larger than most kernel changes):
    python tools/ab_lib_probe.py <path to libadm_hip.so>     # runs tools/gpu_probe.py trainstep (PROBE_B / PROBE_MP) on that build
Build the other side from a git worktree (bash audio-diffusion_amd/csrc/build.sh) and copy its .so next to this script."""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-diffusion_amd"))
from audiodiffusion import _native
_native.DEFAULT_LIB = os.path.abspath(sys.argv[1])
sys.argv = ["gpu_probe.py", "trainstep"]
runpy.run_path(os.path.join(ROOT, "tools", "gpu_probe.py"), run_name="__main__")
