"""`-m gpu`: the N > 1 path on real hardware.  Two ranks run the REAL sharded sampling flow - afm.dist.sharded_sample
over the native CMDM and CDM loops, Philox noise keyed by the global sample index, one all_gather at the end - and the gathered result
must be bit-identical to the single-process run of the same job.  On a 1-GPU box both ranks share cuda:0 (gloo; the device shards are
gathered through host memory); with >= 2 GPUs the same test also runs one rank per GPU over RCCL ("nccl"), so the first multi-GPU box
exercises RCCL in the test suite and not first in the bench.  `bench.py --gpus 2` is run end to end both ways: under torch.distributed.run and as the plain command (which launches its own ranks)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "helpers", "dist_gpu_worker.py")


def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _launch(nproc, script_args, env_extra, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + script_args
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-6000:]
    return r


def _single_process(total):
    sys.path.insert(0, os.path.join(ROOT, "tests", "helpers"))
    import dist_gpu_worker as w
    return w.job(torch.device("cuda:0"), total, 0, 1)


@pytest.mark.parametrize("total", [5])          # uneven shards: 3 + 2 samples
def test_two_ranks_sharing_the_gpu_equal_one_process(tmp_path, total):
    out = str(tmp_path / "two_ranks.pt")
    _launch(2, [WORKER, out, str(total)], dict(AFM_TEST_SHARE_GPU="1", AFM_DIST_BACKEND="gloo"))
    got = torch.load(out)
    assert got["world"] == 2 and got["backend"] == "gloo"
    motion, contact = _single_process(total)
    assert got["motion"].shape == (total, 24, 263) and got["contact"].shape == (total, 1024, 6)
    assert torch.equal(got["motion"], motion.cpu()), (got["motion"] - motion.cpu()).abs().max()
    assert torch.equal(got["contact"], contact.cpu()), (got["contact"] - contact.cpu()).abs().max()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="one rank per GPU over RCCL needs >= 2 GPUs (1-GPU box: the gloo variant above)")
def test_two_ranks_over_rccl_equal_one_process(tmp_path):
    out = str(tmp_path / "two_ranks_rccl.pt")
    _launch(2, [WORKER, out, "6"], {})
    got = torch.load(out)
    assert got["world"] == 2 and got["backend"] == "nccl"
    motion, contact = _single_process(6)
    assert torch.equal(got["motion"], motion.cpu()) and torch.equal(got["contact"], contact.cpu())


def test_bench_two_ranks_end_to_end():
    """`bench.py --gpus 2` exactly as the driver launches it (torch.distributed.run, one JSON line from rank 0): strong scaling is the
    headline for N > 1, the weak-scaling rate rides along, both values are positive and consistent with their ms_per_step."""
    share = torch.cuda.device_count() < 2
    env = dict(AFM_BENCH_SHARE_GPU="1", AFM_DIST_BACKEND="gloo") if share else {}
    r = _launch(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2"], env)
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 5 and line["warmup"] == 2
    assert line["scaling"] == "strong" and line["config"]["batch_per_gpu"] == 16 and line["config"]["job_samples"] == 32
    assert line["value"] > 0 and abs(line["value"] * line["ms_per_step"] / 1e3 - 1.0) < 1e-2            # strong: value = K / t
    other = line["other_scaling_mode"]
    assert other["scaling"] == "weak" and other["batch_per_gpu"] == 32 and other["job_samples"] == 64 and other["value"] > 0
    assert line["roofline"] and line["roofline"]["frac"] < 1 and line["secondary"] is None and line["cpu_baseline"] is None
    # who was in the job, and what the final all_gather costs on its own
    seen = line["ranks_seen"]
    assert [e["rank"] for e in seen] == [0, 1] and all(e["device_name"] for e in seen) and line["batch_per_gpu_by_rank"] == [16, 16]
    assert line["shared_gpu_test_mode"] == share and line["distinct_devices"] == (1 if share else 2)
    assert all(e["backend"] == ("gloo" if share else "nccl") for e in seen)
    g = line["final_all_gather"]
    assert g["bytes_per_rank"] == 16 * 196 * 263 * 4 and g["max_over_ranks_median_us"] > 0 and g["backend"] == seen[0]["backend"]
    assert set(line["scaling_modes"]) == {"weak", "strong"}


def test_plain_bench_command_with_two_gpus_launches_its_own_ranks():
    """the PLAIN command `python bench.py --gpus 2 --steps 5 --warmup 2` - no torch.distributed.run in front, no RANK /
    WORLD_SIZE in the environment - returns rc 0 with one JSON line, n_gpus 2 and two ranks seen (bench.py re-executes itself under
    torch.distributed.run; on a 1-GPU box the two ranks share cuda:0 over gloo, with >= 2 GPUs they run one per GPU over RCCL)."""
    share = torch.cuda.device_count() < 2
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "AFM_SELF_LAUNCHED")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    if share:
        env.update(AFM_BENCH_SHARE_GPU="1", AFM_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2"], capture_output=True, text=True,
                       env=env, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-6000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-3000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 5 and line["warmup"] == 2 and line["value"] > 0
    assert [e["rank"] for e in line["ranks_seen"]] == [0, 1] and len(line["ranks_seen"]) == 2
    assert line["distinct_devices"] == (1 if share else 2) and line["batch_per_gpu_by_rank"] == [16, 16]


def test_bench_two_ranks_over_rccl_one_rank_per_gpu():
    """The first multi-GPU lease, made boring: `bench.py --gpus 2` with NO backend override and NO device sharing, i.e.
    exactly the driver's command - the line must prove that RCCL saw two ranks on two DISTINCT devices (`ranks_seen` comes from an
    all_gather over the job's own backend).  A 1-GPU box cannot run it and says why."""
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip(f"needs >= 2 GPUs for one rank per GPU over RCCL; this box has {n} (the shared-GPU gloo variant above covers the control flow)")
    env = {k: "" for k in ("AFM_BENCH_SHARE_GPU", "AFM_DIST_BACKEND")}
    r = _launch(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2"], env)
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    seen = line["ranks_seen"]
    assert line["n_gpus"] == 2 and line["distinct_devices"] == 2 and not line["shared_gpu_test_mode"]
    assert [e["rank"] for e in seen] == [0, 1] and all(e["backend"] == "nccl" for e in seen)
    assert len({(e["host"], e["pci_bus_id"]) for e in seen}) == 2 and line["batch_per_gpu_by_rank"] == [16, 16]
    assert line["final_all_gather"]["backend"] == "nccl" and line["final_all_gather"]["max_over_ranks_median_us"] > 0
