"""bench.py's multi-rank branch executed on the ONE GPU a box of this pool has: `--gpus 2
--one-gpu-loopback` spawns two ranks under torch.distributed.run, both on device 0, RCCL's socket transport over `lo`.  What runs is
what an 8-GPU driver run runs -- self-spawn, NCCL-backend process group, broadcast, side-stream all-reduces per bucket, finish(),
barrier + max-over-ranks timing, the overlap record, ONE rank-0 JSON line -- at OF-tiny size."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(args, timeout=900):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    # stdout carries ONE line, the JSON record (RCCL's version banner and every warning go to stderr)
    assert r.returncode == 0 and len(lines) == 1 and lines[0].startswith("{"), (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    return json.loads(lines[0])


@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
def test_bench_two_ranks_on_one_gpu_over_loopback():
    out = _run(["--gpus", "2", "--one-gpu-loopback", "--family", "OF-tiny", "--batch", "2", "--T", "2", "--L", "24", "--steps", "2",
                "--warmup", "1", "--no-cpu-baseline", "--no-reference-eager", "--sweep-comm", "--sweep-steps", "2"])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 4 and out["config"]["parallelism"] == "dp2" and "one_gpu_loopback" in out["config"]
    ov = out["overlap"]
    assert ov["rccl_ranks"] == 2 and ov["backend"] == "nccl" and ov["collectives_per_step"] >= 3
    assert ov["exposed_wait_ms_per_step"] is not None and ov["allreduce_bytes_per_step_per_gpu"] > 0
    assert out["value"] > 0 and abs(out["value"] - out["config"]["images_per_step"] / out["ms_per_step"] * 1e3) <= 0.02 * out["value"]
    assert out["loss_last_step"] == out["loss_last_step"]            # not NaN
    # --sweep-comm: the tuning table of the first real-node run, in the same invocation
    sw = out["comm_sweep"]["settings"]
    assert [(s["reserve_cus"], s["wire_dtype"]) for s in sw] == [(r, w) for w in ("float32", "bfloat16") for r in (0, 8, 16, 32)]
    for s in sw:
        assert s["ms_per_step"] > 0 and s["exposed_wait_ms_per_step"] is not None
        assert s["allreduce_ms"]["n"] >= 2 * 3 and s["allreduce_ms"]["min"] > 0 and len(s["last_step_buckets"]) >= 3
        assert all(b["ms"] > 0 and b["bytes"] > 0 for b in s["last_step_buckets"])
    fp32 = sum(b["bytes"] for b in sw[0]["last_step_buckets"])
    bf16 = sum(b["bytes"] for b in sw[4]["last_step_buckets"])
    assert bf16 * 2 == fp32                                          # half the bytes on the wire


@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
def test_bench_single_rank_line_at_tiny_size():
    out = _run(["--family", "OF-tiny", "--batch", "2", "--T", "2", "--L", "24", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                "--no-reference-eager"])
    assert out["n_gpus"] == 1 and out["overlap"]["rccl_ranks"] == 1 and "one_gpu_loopback" not in out["config"] and "comm_sweep" not in out
    assert out["roofline"]["all_gemm_frac"] > 0
