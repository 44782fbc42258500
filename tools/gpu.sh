#!/usr/bin/env bash
# Local wrapper around gpurun: rebuild BOTH libraries and check the exported symbols first (a stale libadm_hip.so travels to the
# GPU box as it is), then run the given command there.   tools/gpu.sh <timeout_s> '<command>' [logfile]
set -e
cd "$(dirname "$0")/.."
bash audio-diffusion_amd/csrc/build.sh | tail -1
bash audio-diffusion_amd/csrc/build.sh emu | tail -1
python -m pytest tests/test_abi.py -q -x 2>&1 | tail -1
gpurun --timeout "$1" -- "$2" > "${3:-/tmp/gpu.log}" 2>&1
