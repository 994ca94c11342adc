"""bench.py's own launcher: `python bench.py --gpus 2` (no torchrun) must spawn its ranks, run the sharded
workloads through kjarni_amd.distributed and print exactly one JSON line.  Exercised on the CPU with --dry-run-cpu
(gloo + a host stub in place of the HIP encoder); the GPU path differs only in the encoder object and the backend."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, launcher=()):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, *launcher, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                       timeout=300, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("workload,extra", [("rerank", ["--pairs", "257"]), ("embed", ["--sentences", "65"])])
def test_plain_invocation_spawns_its_ranks(workload, extra):
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", workload, "--dry-run-cpu", *extra])
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["warmup"] == 1 and r["data"] == "dry-run"
    if workload == "rerank":
        assert r["scaling"] == "strong" and r["config"]["rows_per_step"] == 257 and r["config"]["rows_per_gpu"] == 129
        assert r["unit"] == "pairs/s"
        assert "rerank" not in r
    else:
        assert r["scaling"] == "weak" and r["config"]["rows_per_step"] == 130 and r["unit"] == "sentences/s"
        # the default workload also carries the north star's strong-scaling rerank leg, on the same N
        leg = r["rerank"]
        assert leg["n_gpus"] == 2 and leg["scaling"] == "strong" and leg["pairs_per_step"] == 100000
        assert leg["pairs_per_gpu"] == 50000 and leg["pairs_per_s"] > 0 and leg["ms_per_step"] > 0 and leg["steps"] == 2
        # ... and the third sharded path of SURVEY.md section 8(e): the cosine scan over a corpus sharded by rows (local search,
        # ONE all-gather of the candidate lists, merge by (score, global index)); the leg asserts its own result
        sh = r["scan_sharded"]
        assert sh["n_gpus"] == 2 and sh["scaling"] == "weak" and sh["queries_1"]["ms_per_search"] > 0
        assert sh["queries_64"]["doc_queries_per_s"] > 0 and sh["queries_64"]["steps"] == 2
    # the communicator the ranks built spans exactly --gpus ranks (gloo here, RCCL on the GPU path)
    assert r["collective"]["ranks"] == 2 and r["collective"]["allreduce_of_ones"] == 2
    assert r["value"] > 0 and r["higher_is_better"] is True and r["vs_baseline"] is None


def test_under_torchrun():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--workload", "rerank", "--pairs", "100", "--dry-run-cpu"],
             launcher=("-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                       "--master-port", "29533"))
    assert r["n_gpus"] == 2 and r["config"]["rows_per_gpu"] == 50


def test_world_size_mismatch_is_an_error():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run-cpu"], capture_output=True,
                       text=True, timeout=120, env=env, cwd=ROOT)
    assert p.returncode == 2 and "WORLD_SIZE" in p.stderr


def test_single_rank_has_no_collective_and_can_skip_the_rerank_leg():
    r = _run(["--gpus", "1", "--steps", "1", "--warmup", "0", "--dry-run-cpu", "--sentences", "10", "--no-rerank-leg"])
    assert r["n_gpus"] == 1 and "collective" not in r and "rerank" not in r and "scan_sharded" not in r
    r = _run(["--gpus", "1", "--steps", "1", "--warmup", "0", "--dry-run-cpu", "--sentences", "10", "--pairs", "77"])
    assert r["rerank"]["pairs_per_step"] == 77 and r["rerank"]["n_gpus"] == 1


def test_eight_ranks_as_the_driver_launches_them():
    """The driver's 8-GPU form (--gpus 8, one rank per GPU) on the CPU stub: the communicator spans 8 ranks, the embed leg is
    weak (rows per step = 8 x sentences), the rerank leg strong over 100 003 pairs (uneven blocks: 3 ranks hold one pair more),
    every rank builds only its own rows, and the host-thread pool of a rank is its share of the granted CPUs."""
    r = _run(["--gpus", "8", "--steps", "1", "--warmup", "1", "--dry-run-cpu", "--sentences", "33", "--pairs", "100003"])
    assert r["n_gpus"] == 8 and r["config"]["rows_per_step"] == 8 * 33 and r["scaling"] == "weak"
    assert r["collective"]["ranks"] == 8 and r["collective"]["allreduce_of_ones"] == 8
    leg = r["rerank"]
    assert leg["n_gpus"] == 8 and leg["scaling"] == "strong" and leg["pairs_per_step"] == 100003 and leg["pairs_per_gpu"] == 12501
    assert 1 <= r["host_threads_per_rank"] <= max(1, (os.cpu_count() or 8) // 8)
    assert r["scan_sharded"]["n_gpus"] == 8 and r["scan_sharded"]["corpus"].startswith("[8 x ")


def test_in_process_arrangement_drives_every_device_from_one_process():
    """--in-process: ONE process, the library's EncoderGroup arrangement (kjarni_hip_group_*_allgather; what FFI callers get).
    On the CPU stub: weak embed rows (N x sentences), the strong rerank leg over uneven blocks, every device's buffer holding
    every row, and the JSON naming the arrangement that produced `value`."""
    r = _run(["--gpus", "3", "--in-process", "--steps", "2", "--warmup", "1", "--dry-run-cpu", "--sentences", "21", "--pairs", "100"])
    assert r["n_gpus"] == 3 and r["scaling"] == "weak" and r["config"]["rows_per_step"] == 63 and r["config"]["rows_per_gpu"] == 21
    assert r["config"]["arrangement"].startswith("in-process group") and len(r["steps_ms"]) == 2
    assert "collective" not in r   # no torch.distributed communicator in this arrangement
    leg = r["rerank"]
    assert leg["scaling"] == "strong" and leg["pairs_per_step"] == 100 and leg["pairs_per_gpu"] == 34 and leg["n_gpus"] == 3
    r = _run(["--gpus", "2", "--in-process", "--steps", "1", "--warmup", "0", "--dry-run-cpu", "--workload", "rerank", "--pairs", "9"])
    assert r["unit"] == "pairs/s" and r["config"]["rows_per_gpu"] == 5 and "rerank" not in r


def test_in_process_refuses_a_multi_process_launch():
    env = dict(os.environ, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--in-process", "--dry-run-cpu"],
                       capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    assert p.returncode == 2 and "ONE process" in p.stderr


def test_the_line_names_the_arrangement_and_the_io():
    r = _run(["--gpus", "1", "--steps", "1", "--warmup", "0", "--dry-run-cpu", "--sentences", "10", "--no-rerank-leg"])
    assert r["config"]["arrangement"].startswith("one process per GPU") and "resident in HBM" in r["config"]["io"]
