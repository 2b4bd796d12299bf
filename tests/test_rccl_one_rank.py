"""RCCL on the one GPU a builder's box has: a group of ONE rank over backend `nccl` in a fresh child process.
Nothing here measures scaling (there is nobody to exchange with); it proves that the library loads beside
libsvt_hip.so, that the collective code path of sparsearray_amd/parallel.py and of bench.py works against the real
backend and gives, bit for bit, the results it gives without a group.  The reference has no multi-device path
(R/thread-control.R:87-92: one process, one OpenMP team)."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _env(port):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update({"HSA_ENABLE_IPC_MODE_LEGACY": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    return env


def test_parallel_py_collectives_over_rccl_one_rank(hip, tmp_path):
    torch.cuda.empty_cache()
    out = tmp_path / "verdict.json"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "workers", "rccl_one_rank_worker.py"), str(out)],
                       env=_env(30500 + os.getpid() % 400), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    v = json.loads(out.read_text())
    assert v["backend"] == "nccl" and v["world_size"] == 1
    assert v["crossprod_identical"] and v["colsums_identical"] and v["colvars_identical"] and v["rowsum_identical"], v
    assert v["allreduce_of_ones"] == [1.0] * 4
    assert v["all_reduces_in_flight_after_5_steps"] >= 1       # the asynchronous form really was used


def test_bench_collective_path_over_rccl_one_rank(hip):
    """bench.py's N > 1 code path (process group, barriers, MAX over ranks, `multi_gpu` diagnostics with the
    all-reduce alone / product alone / spare-CU variant) with backend nccl and one rank; same product as the plain
    one-GPU line."""
    torch.cuda.empty_cache()
    common = ["--nrow", "262144", "--ncol", "4000", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-extras"]

    def line(extra, port):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common + extra, env=_env(port),
                           capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
        return json.loads(p.stdout.strip().splitlines()[-1])
    plain = line([], 30950 + os.getpid() % 200)
    forced = line(["--force-collectives", "--backend", "nccl"], 31200 + os.getpid() % 200)
    m = forced["multi_gpu"]
    assert m["backend"] == "nccl" and m["world_size"] == 1
    assert m["allreduce_alone_ms"] > 0 and m["product_alone_ms"] > 0 and m["allreduce_bytes"] == 4000 * 128 * 8
    assert "multi_gpu" not in plain
    a, b = plain["config"]["result_checksum"], forced["config"]["result_checksum"]
    assert a == b
    assert forced["n_gpus"] == 1 and forced["value"] > 0
