"""World-size-2 `gloo` tests (CPU) of the N>1 path: contiguous sharding of the (batch x k_sample)
dimension, sample-index keyed work, one all_gather at the end, result independent of the rank count."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from afm import dist as adist


def test_shard_range_covers_everything():
    for total in (1, 7, 32, 33):
        for world in (1, 2, 3, 8):
            spans = [adist.shard_range(total, r, world) for r in range(world)]
            assert sum(c for _, c in spans) == total
            assert [s for s, _ in spans] == [sum(c for _, c in spans[:r]) for r in range(world)]
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def test_job_shard_weak_and_strong():
    """bench.py's two node modes: weak = 32 samples on every rank, strong = one 32-sample job split over the ranks."""
    for world in (1, 2, 4, 8):
        weak = [adist.job_shard("weak", 32, r, world) for r in range(world)]
        assert [w[0] for w in weak] == [32 * r for r in range(world)] and all(w[1] == 32 and w[2] == 32 * world for w in weak)
        strong = [adist.job_shard("strong", 32, r, world) for r in range(world)]
        assert sum(c for _, c, _ in strong) == 32 and all(t == 32 for _, _, t in strong)
        assert [s0 for s0, _, _ in strong] == [sum(c for _, c, _ in strong[:r]) for r in range(world)]
        assert all(c == 32 // world for _, c, _ in strong)                   # B = 4 per GPU at 8 GPUs
    with pytest.raises(ValueError):
        adist.job_shard("both", 32, 0, 1)


def _strong_worker(rank, world, port, total, q):
    """One `total`-sample job sharded over the ranks exactly as `bench.py --scaling strong` does it: job_shard -> local run keyed by
    the GLOBAL sample index -> one all_gather."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    adist.init_process_group("gloo")
    i0, cnt, job = adist.job_shard("strong", total, rank, world)
    cond = (torch.arange(job, dtype=torch.float32) * 0.5)[i0:i0 + cnt]
    local = _fake_sampler(dict(cond=cond), cnt, i0)
    out = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(out, local)
    q.put((rank, torch.cat(out, 0)))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    """A port nobody listens on right now (a fixed number collides with a socket of the previous run still in TIME_WAIT, or with another job)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def test_strong_scaling_two_ranks_equal_one_process():
    total = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_strong_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = _fake_sampler(dict(cond=torch.arange(total, dtype=torch.float32) * 0.5), total, 0)
    assert torch.equal(got[0], want) and torch.equal(got[1], want)


def _seen_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    adist.init_process_group("gloo")
    seen = adist.ranks_seen(torch.device("cpu"), world)
    g = adist.time_all_gather(torch.full((3, 4), float(rank)), world, reps=3)
    q.put((rank, seen, g))
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_seen_and_gather_timing_bookkeeping_two_ranks():
    """What bench.py puts into the N > 1 line (`ranks_seen`, `distinct_devices`, `final_all_gather`): every rank sees every rank, in rank
    order, with the backend that carried the gather; CPU ranks count as distinct; the gather timing reports the shard size."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_seen_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {r: (seen, g) for r, seen, g in (q.get(timeout=120) for _ in range(2))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in (0, 1):
        seen, g = got[r]
        assert [e["rank"] for e in seen] == [0, 1] and [e["local_rank"] for e in seen] == [0, 1]
        assert all(e["backend"] == "gloo" and e["device"] is None for e in seen) and seen[0]["pid"] != seen[1]["pid"]
        assert adist.distinct_devices(seen) == 2
        assert g["bytes_per_rank"] == 48 and g["reps"] == 3 and g["backend"] == "gloo" and 0 < g["min_us"] <= g["median_us"] <= g["max_us"]
    # two ranks on ONE device are seen as one device (what bench.py refuses outside the shared-GPU test mode)
    shared = [dict(rank=0, device=0, visible_devices=None), dict(rank=1, device=0, visible_devices=None)]
    assert adist.distinct_devices(shared) == 1
    assert adist.distinct_devices([dict(rank=0, device=0, visible_devices="0"), dict(rank=1, device=0, visible_devices="1")]) == 2
    # the physical identity wins over the visibility mask: overlapping masks that land on the same bus id are ONE device,
    # the same mask and index on two hosts are two
    same_gpu = [dict(rank=0, device=0, visible_devices="0,1", pci_bus_id="0000:05:00", host="a"), dict(rank=1, device=0, visible_devices="0", pci_bus_id="0000:05:00", host="a")]
    assert adist.distinct_devices(same_gpu) == 1
    two_hosts = [dict(rank=0, device=0, visible_devices="0", pci_bus_id="0000:05:00", host="a"), dict(rank=1, device=0, visible_devices="0", pci_bus_id="0000:05:00", host="b")]
    assert adist.distinct_devices(two_hosts) == 2
    # single process: no process group needed
    assert adist.ranks_seen(torch.device("cpu"), 1)[0]["rank"] == 0


def test_plain_bench_command_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus N` without a launcher environment: the decision and the command line are host logic -
    N > 1 and no RANK / WORLD_SIZE -> re-execute under torch.distributed.run with a loopback rendezvous on a free port, arguments passed
    through; under a launcher (or N = 1) nothing is launched.  The launch itself is run end to end with a stand-in script whose ranks
    rendezvous over gloo and print one line from rank 0."""
    import subprocess
    import sys
    assert adist.needs_self_launch(2, env={}) and adist.needs_self_launch(8, env={"MASTER_ADDR": "127.0.0.1"})
    assert not adist.needs_self_launch(1, env={}) and not adist.needs_self_launch(2, env={"WORLD_SIZE": "2", "RANK": "0"})
    cmd = adist.self_launch_command("/x/bench.py", ["--gpus", "2", "--steps", "5"], 2, port=12345, python="py")
    assert cmd == ["py", "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "12345",
                   "/x/bench.py", "--gpus", "2", "--steps", "5"]
    assert adist.self_launch_command("/x/bench.py", [], 4)[7] != "0"          # a free port was picked
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "standin.py"
    script.write_text(
        "import json, os, sys\n"
        f"sys.path.insert(0, {os.path.join(root, 'afford-motion_amd')!r})\n"
        "import torch, torch.distributed as dist\n"
        "from afm import dist as adist\n"
        "n = int(sys.argv[sys.argv.index('--gpus') + 1])\n"
        "if adist.needs_self_launch(n) and not os.environ.get('AFM_SELF_LAUNCHED'):\n"
        "    sys.exit(adist.self_launch(os.path.abspath(__file__), sys.argv[1:], n))\n"
        "rank, world, local = adist.init_process_group('gloo')\n"
        "seen = adist.ranks_seen(None, world)\n"
        "t = torch.tensor([float(rank + 1)]); dist.all_reduce(t)\n"
        "if rank == 0: print(json.dumps({'n_gpus': world, 'ranks': [e['rank'] for e in seen], 'sum': t.item(), 'argv': sys.argv[1:]}))\n"
        "dist.barrier(); dist.destroy_process_group()\n")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "AFM_SELF_LAUNCHED")}
    r = subprocess.run([sys.executable, str(script), "--gpus", "2", "--tag", "x"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    import json
    line = json.loads(lines[0])
    assert line == {"n_gpus": 2, "ranks": [0, 1], "sum": 3.0, "argv": ["--gpus", "2", "--tag", "x"]}


def test_shard_kwargs_slices_only_per_sample_entries():
    kw = dict(x_mask=torch.zeros(6, 4), c_text=["a"] * 6, sigma=0.8, table=torch.zeros(3, 2))
    out = adist.shard_kwargs(kw, 2, 3, 6)
    assert out["x_mask"].shape == (3, 4) and len(out["c_text"]) == 3 and out["sigma"] == 0.8 and out["table"].shape == (3, 2)


def _fake_sampler(kw, count, index0):
    """Deterministic per GLOBAL sample index (what Philox keying gives the real sampler)."""
    idx = torch.arange(index0, index0 + count, dtype=torch.float32)
    return idx[:, None, None] * 10 + kw["cond"][:, None, None] + torch.arange(6, dtype=torch.float32).view(1, 2, 3)


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    adist.init_process_group("gloo")
    kw = dict(cond=torch.arange(total, dtype=torch.float32) * 0.5, flag=True)
    out = adist.sharded_sample(_fake_sampler, total, kw, rank, world)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [8, 5])
def test_two_rank_gather_matches_single_process(total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = adist.sharded_sample(_fake_sampler, total, dict(cond=torch.arange(total, dtype=torch.float32) * 0.5, flag=True), 0, 1)
    assert torch.equal(got[0], want) and torch.equal(got[1], want)


def test_adm_to_amdm_glue_oracle_matches_reference_golden():
    """The CPU oracle's glue (oracle/glue_ref.py) against the reference-generated golden; the product's glue is a HIP kernel
    (afm_contact_glue), checked against the same golden in tests/test_gpu_cdm.py, and refuses CPU tensors."""
    from conftest import golden
    from oracle import glue_ref
    g = golden("adm_to_amdm_glue")
    cond = glue_ref.adm_to_amdm_condition(g["sample"], sigma=float(g["sigma"]), mean=float(g["mean"]), std=float(g["std"]))
    assert torch.allclose(cond, g["cond"].float(), atol=1e-6)
    import pytest
    from afm import ffi
    with pytest.raises(ffi.AfmError):
        adist.adm_to_amdm_condition(g["sample"])
