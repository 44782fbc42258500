"""bench.py ITSELF under `--gpus 2` semantics: the driver's launch line (torch.distributed.run, one rank per device), with
the ADM_BENCH_EMU=1 test hook swapping the MI355X for the CPU-emulation build of the kernels and RCCL for gloo. Everything
else — rank / world plumbing, row-sharded noise, per-step all_gather, barrier-bracketed timing, MAX over ranks, the
training leg's bucketed all-reduce issued from inside the reverse pass, the one JSON line on rank 0 — is the code the
8-GPU scaling run executes."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(n, extra=(), bare=False):
    from native_backend import ensure_emu_built
    ensure_emu_built()
    env = dict(os.environ, ADM_BENCH_EMU="1", ADM_EMU_THREADS="2", OMP_NUM_THREADS="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    port = 29800 + os.getpid() % 1000
    if n == 1 or bare:         # bare: no launcher — bench.py starts its own ranks
        cmd = [sys.executable, "bench.py"]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), "bench.py"]
    cmd += ["--gpus", str(n), "--steps", "2", "--warmup", "1", "--batch-per-gpu", "2", "--ddim-steps", "3", *extra]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                       # exactly one JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_py_two_ranks_prints_one_contract_line():
    d = _run(2)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 4 and d["value"] > 0
    assert abs(d["value"] - 2 * 2 * 2 / (d["ms_per_step"] * 2 / 1e3)) <= 0.02 * d["value"]     # whole-job aggregate
    tr = d["train"]
    assert "error" not in tr, tr
    assert tr["global_batch"] == 4 and tr["value"] > 0
    assert tr["allreduce_buckets_overlapped"] == tr["allreduce_buckets"] >= 1      # every bucket fired inside the reverse pass
    assert "error" not in d["mel"], d["mel"]


def test_bench_py_result_is_independent_of_the_rank_count_and_a_bare_gpus_2_launches_its_own_ranks():
    """Same global batch on 1 rank and on 2: the gathered uint8 images (checksum) are identical — rows never interact."""
    one = _run(1, ["--batch-per-gpu", "4", "--no-train-leg", "--no-mel-leg"])
    os.environ["ADM_BENCH_FORCE_PG"] = "1"      # 1 rank but with the process group, so `gathered` exists
    try:
        one_pg = _run(1, ["--batch-per-gpu", "4", "--no-train-leg", "--no-mel-leg"])
    finally:
        del os.environ["ADM_BENCH_FORCE_PG"]
    # the 2-rank run is BARE — `python bench.py --gpus 2` with no launcher around it, as the driver writes its N = 1 line: bench.py
    # re-executes itself under torch.distributed.run on 127.0.0.1 and rank 0 prints the one contract line — and in `--scaling
    # strong` mode: one global batch of 4 (config 3 as written) split by rows instead of 2 per rank; the same 4 rows either way
    two = _run(2, ["--no-train-leg", "--no-mel-leg", "--scaling", "strong", "--global-batch", "4"], bare=True)
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and two["config"]["global_batch"] == 4 and two["value"] > 0
    assert one["gathered_checksum"] is None
    assert one_pg["gathered_checksum"] == two["gathered_checksum"] and two["gathered_checksum"] > 0
    # an UNEVEN global batch: 3 rows on 2 ranks = shards of 2 and 1, the short one padded for the gather; the first
    # three rows of the same seed on one rank give the same images (rows never interact, and the noise of row r does not depend on the batch)
    os.environ["ADM_BENCH_FORCE_PG"] = "1"
    try:
        three = _run(1, ["--batch-per-gpu", "3", "--no-train-leg", "--no-mel-leg"])
    finally:
        del os.environ["ADM_BENCH_FORCE_PG"]
    uneven = _run(2, ["--no-train-leg", "--no-mel-leg", "--scaling", "strong", "--global-batch", "3"])
    assert uneven["config"]["global_batch"] == 3 and uneven["value"] > 0
    assert uneven["gathered_checksum"] == three["gathered_checksum"] > 0


def test_bench_py_reports_a_hung_training_leg_beside_the_measured_headline():
    """The side legs run after the headline is measured; a training leg that does not come back (a collective some rank never
    reaches) must cost its own record only: every rank's watchdog fires, rank 0 prints the line with the failure in place."""
    d = _run(2, ["--no-mel-leg", "--train-leg-timeout", "0.05"])
    assert d["value"] > 0 and d["n_gpus"] == 2
    assert "timeout" in d["train"]["error"]



def test_bench_py_eight_ranks_weak_with_the_training_leg_and_strong_against_one_rank():
    """The world size BASELINE configs 3 and 5 are written for: 8 ranks (gloo, toy network on the emulator). Weak mode with the training
    leg — every gradient bucket all-reduced from inside the reverse pass on EVERY rank (MIN over the job) — and config 3 as written
    (`--scaling strong`: one global batch of 8 split by rows) against the same 8 rows on one rank: identical gathered images."""
    d = _run(8, ["--batch-per-gpu", "1", "--no-mel-leg"])
    assert d["n_gpus"] == 8 and d["config"]["global_batch"] == 8 and d["scaling"] == "weak" and d["value"] > 0
    tr = d["train"]
    assert "error" not in tr, tr
    assert tr["global_batch"] == 8 * tr["batch_per_gpu"] and tr["allreduce_buckets_overlapped"] == tr["allreduce_buckets"] >= 1
    assert tr["allreduce_overlapped_on_every_rank"] is True
    eight = _run(8, ["--no-train-leg", "--no-mel-leg", "--scaling", "strong", "--global-batch", "8"])
    os.environ["ADM_BENCH_FORCE_PG"] = "1"
    try:
        one = _run(1, ["--batch-per-gpu", "8", "--no-train-leg", "--no-mel-leg"])
    finally:
        del os.environ["ADM_BENCH_FORCE_PG"]
    assert eight["scaling"] == "strong" and eight["config"]["global_batch"] == 8
    assert eight["gathered_checksum"] == one["gathered_checksum"] > 0


def test_bench_py_one_rank_training_leg_runs_the_overlapped_all_reduce_in_its_own_group():
    """The driver's N = 1 line: no launcher, no process group — the training leg builds a one-rank group of its own so that the
    bucket-hook -> asynchronous all-reduce path executes under a real backward pass (every bucket, overlapped)."""
    d = _run(1, ["--no-mel-leg"])
    tr = d["train"]
    assert "error" not in tr, tr
    own = tr["one_rank_group"]          # the headline of the leg is measured WITHOUT the group, the one-rank run is reported beside it
    assert "error" not in own and own["allreduce_buckets_overlapped"] == own["allreduce_buckets"] >= 1 and own["ms_per_step"] > 0
    assert tr["allreduce_overlapped_on_every_rank"] is None and tr["ms_per_step"] > 0
