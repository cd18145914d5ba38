"""bench.py --gpus N outside torch.distributed.run must launch its own ranks from a process that has not touched
torch / HIP, relay their output and return their exit code (the driver may call it either way)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def test_self_launch_relays_rank_failures_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-tier check of the launcher's failure path")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--settle", "0"],
                       env=_env(), cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "bench.py needs a GPU" in p.stderr                  # the ranks were started and said why they stopped


def test_parent_does_not_import_torch_before_launching():
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[: src.index("def main()")]
    assert "import torch" not in head.replace("import torch\n            g =", "")     # only inside cpu_share(), called by ranks
    body = src[src.index("def main()"):]
    assert body.index("self_launch(args)") < body.index("import torch")


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_rehearsal(gpu):
    """N = 2 ranks sharing the one GPU of this box (gloo for the count reduce): rc 0, one JSON line, whole-job
    value = 2 shards, and the reduced counts are the sum of two different shards (mapped fraction as at N = 1)."""
    args = ["--config", "2", "--reads", "3000000", "--steps", "3", "--warmup", "1", "--settle", "0", "--cpu-sample", "0"]
    outs = {}
    for n in (1, 2):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)] + args,
                           env=_env(SCG_BENCH_SHARE_GPU="1"), cwd=ROOT, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-2000:]
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1
        outs[n] = json.loads(lines[0])
    assert outs[2]["n_gpus"] == 2 and outs[2]["scaling"] == "weak"
    assert outs[2]["config"]["reads_per_gpu"] == 3000000
    assert abs(outs[2]["mapped_fraction"] - outs[1]["mapped_fraction"]) < 0.002
    assert outs[1]["e2e"] and "pinned_batches" in outs[1]["e2e"] and "fastq_file" in outs[1]["e2e"]


@pytest.mark.gpu
def test_rccl_path_at_world_size_one(gpu):
    """SCG_BENCH_FORCE_DIST=1: init_process_group("nccl", device_id=...) -- RCCL on ROCm -- the all_reduce of the bound
    count vector, the barriers and the MAX reduction of the step time all execute at world size 1 on this box's one GPU
    (the multi-GPU form of the bench differs only in the number of ranks).  Same counts as without the process group."""
    args = ["--config", "2", "--reads", "2000000", "--steps", "3", "--warmup", "1", "--settle", "0", "--cpu-sample", "0",
            "--e2e-sample", "0", "--e2e-file-sample", "0"]
    outs = {}
    for forced in ("0", "1"):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + args,
                           env=_env(SCG_BENCH_FORCE_DIST=forced, HSA_ENABLE_IPC_MODE_LEGACY="0"), cwd=ROOT, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-3000:]
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1
        outs[forced] = json.loads(lines[0])
    assert "RCCL all-reduce" in outs["1"]["config"]["parallelism"] and "RCCL" not in outs["0"]["config"]["parallelism"]
    assert outs["1"]["mapped_fraction"] == outs["0"]["mapped_fraction"] and outs["1"]["n_gpus"] == 1
