"""Multi-GPU sampling: shard the flattened (batch x k_sample) dimension over the ranks of one node,
replicate the (<= 50 MB) weights, run each rank's loop independently and gather once at the end.

The reference samples its k repeats sequentially on one GPU (test.py:88-101) and has no
inference-time parallelism; samples are independent for the whole loop in eval mode (SURVEY.md
section 8e), so the only collective is one all_gather of [B/G, L, D] float32 (RCCL over xGMI on the
GPU box - torch's "nccl" backend; gloo in the CPU tests).  Noise is keyed by the GLOBAL sample index,
so the gathered result does not depend on the number of ranks.
"""
from __future__ import annotations

import os
from typing import Any, Callable, Dict, Tuple

import torch
import torch.distributed as dist


def env_rank_world() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def init_process_group(backend: str = None) -> Tuple[int, int, int]:
    """One process per GPU (launched by torch.distributed.run); no-op for a single process."""
    rank, world, local = env_rank_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # AFM_DIST_BACKEND=gloo: functional testing of the multi-rank flow on a box with fewer GPUs than ranks
            backend = os.environ.get("AFM_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def needs_self_launch(n_gpus: int, env=None) -> bool:
    """True when a script asked for `n_gpus` > 1 ranks but was started as a plain process (no torch.distributed.run environment):
    `python bench.py --gpus N` then launches its own ranks (self_launch) instead of failing on WORLD_SIZE."""
    env = os.environ if env is None else env
    return n_gpus > 1 and not (env.get("WORLD_SIZE") and env.get("RANK") is not None)


def self_launch_command(script: str, argv: list, n_gpus: int, port: int = None, python: str = None) -> list:
    """The command line a plain `python <script> --gpus N ...` re-executes itself as: the reference's own launch convention
    (scripts/t2m_contact_motion/train_ddp.sh:9: one process per GPU of ONE node under torch.distributed.run, rendezvous on the loopback
    address with a free port), with the script's arguments passed through unchanged."""
    import socket
    import sys
    if port is None:
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n_gpus)}", "--master-addr", "127.0.0.1",
            "--master-port", str(int(port)), script] + list(argv)


def self_launch(script: str, argv: list, n_gpus: int) -> int:
    """Run `script` as `n_gpus` ranks (see self_launch_command) and return the launcher's exit code; the ranks inherit stdout / stderr, so
    rank 0's single JSON line is this process's output.  AFM_SELF_LAUNCHED marks the children (a child never launches again)."""
    import subprocess
    env = dict(os.environ, AFM_SELF_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")             # torch.distributed.run would set 1 (and warn): the CPU side of a rank is enqueue only
    return subprocess.call(self_launch_command(script, argv, n_gpus), env=env)


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [start, start+count) of `total` samples; the first total % world ranks get one extra."""
    base, extra = divmod(total, world)
    count = base + (1 if rank < extra else 0)
    start = rank * base + min(rank, extra)
    return start, count


def job_shard(scaling: str, batch: int, rank: int, world: int) -> Tuple[int, int, int]:
    """(first global sample, samples on this rank, samples in the whole job) for the two ways a node is used:
    "weak"   every rank owns `batch` samples (the job grows with the node: 32 per GPU, the metric's configuration);
    "strong" ONE job of `batch` samples (k_sample = 32 of test.py:88-101) is sharded contiguously over the ranks."""
    if scaling == "weak":
        return rank * batch, batch, batch * world
    if scaling != "strong":
        raise ValueError(f"scaling must be 'weak' or 'strong', got {scaling!r}")
    start, count = shard_range(batch, rank, world)
    return start, count, batch


def shard_kwargs(kwargs: Dict[str, Any], start: int, count: int, total: int) -> Dict[str, Any]:
    """Slice every per-sample entry (tensor or list whose leading size is `total`) of a batch dict."""
    out = {}
    for k, v in kwargs.items():
        if isinstance(v, torch.Tensor) and v.dim() > 0 and v.shape[0] == total:
            out[k] = v[start:start + count]
        elif isinstance(v, (list, tuple)) and len(v) == total:
            out[k] = v[start:start + count]
        else:
            out[k] = v
    return out


def all_gather_rows(local: torch.Tensor, world: int = None) -> list:
    """all_gather of equally-shaped shards -> list in rank order.  RCCL ("nccl") gathers device tensors in place over xGMI; the gloo
    backend (CPU tests, and the shared-GPU functional tests of the N > 1 control flow) has no device all_gather, so device shards take
    one round trip through host memory there - never on the measured path."""
    world = dist.get_world_size() if world is None else world
    if dist.get_backend() == "gloo" and local.is_cuda:
        host = local.detach().cpu().contiguous()
        bufs = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(bufs, host)
        return [b.to(local.device) for b in bufs]
    bufs = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(bufs, local.contiguous())
    return bufs


def ranks_seen(device: torch.device = None, world: int = None) -> list:
    """Who is in the job: every rank's (rank, local_rank, backend, device index, device name, PCI bus id) gathered on all ranks, in rank
    order.  On an N-GPU box under "nccl" this is the proof that RCCL saw N ranks on N DISTINCT devices (bench.py puts it into the N > 1
    line and refuses to report a line whose ranks share a device unless the shared-GPU test switch says so); under gloo (CPU tests,
    shared-GPU functional tests) it exercises the same bookkeeping.  One all_gather_object of a small dict per rank."""
    rank, env_world, local = env_rank_world()
    world = (dist.get_world_size() if dist.is_initialized() else env_world) if world is None else world
    import socket
    me = {"rank": rank, "local_rank": local, "backend": dist.get_backend() if dist.is_initialized() else None, "device": None,
          "device_name": None, "pci_bus_id": None, "pid": os.getpid(), "host": socket.gethostname()}
    if device is not None and device.type == "cuda":
        idx = device.index if device.index is not None else torch.cuda.current_device()
        props = torch.cuda.get_device_properties(idx)
        bus = getattr(props, "pci_bus_id", None)
        if bus is not None:                           # domain:bus:device - the bus number alone is not unique across PCI domains
            bus = f"{getattr(props, 'pci_domain_id', 0):04x}:{bus:02x}:{getattr(props, 'pci_device_id', 0):02x}"
        me.update(device=idx, device_name=torch.cuda.get_device_name(idx), pci_bus_id=bus)
        me["visible_devices"] = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES") or os.environ.get("CUDA_VISIBLE_DEVICES")
    if world == 1 or not dist.is_initialized():
        return [me]
    out = [None] * world
    dist.all_gather_object(out, me)
    return sorted(out, key=lambda r: r["rank"])


def device_identity(r: dict):
    """Physical identity of a rank's device: (host, PCI bus id) when the bus id is known - two ranks with different but overlapping
    visibility masks that land on the same GPU collapse to one, ranks on different hosts with equal masks stay apart -, else the
    (host, visible-device mask, device index) triple; CPU ranks (device None) count once each."""
    if r.get("device") is None:
        return ("cpu", r.get("host"), r["rank"])
    if r.get("pci_bus_id") is not None:
        return ("pci", r.get("host"), r["pci_bus_id"])
    return ("idx", r.get("host"), r.get("visible_devices"), r["device"])


def distinct_devices(seen: list) -> int:
    """Number of physically distinct devices among the ranks of `ranks_seen` (see device_identity)."""
    return len({device_identity(r) for r in seen})


def time_all_gather(local: torch.Tensor, world: int = None, reps: int = 20) -> dict:
    """Wall time of the path's only collective, measured on its own: `reps` all_gathers of this rank's shard, each between a barrier and a
    device synchronise; the median in microseconds (max over ranks is taken by the caller when it wants one number)."""
    import statistics
    import time
    world = dist.get_world_size() if world is None else world
    ts = []
    for _ in range(reps + 2):
        if local.is_cuda:
            torch.cuda.synchronize(local.device)
        dist.barrier()
        t0 = time.perf_counter()
        all_gather_rows(local, world)
        if local.is_cuda:
            torch.cuda.synchronize(local.device)
        ts.append(1e6 * (time.perf_counter() - t0))
    ts = ts[2:]                                         # the first calls set the communicator up
    return {"median_us": round(statistics.median(ts), 1), "min_us": round(min(ts), 1), "max_us": round(max(ts), 1), "reps": reps,
            "bytes_per_rank": local.numel() * local.element_size(), "backend": dist.get_backend()}


def sharded_sample(sample_fn: Callable[[Dict[str, Any], int, int], torch.Tensor], total: int, model_kwargs: Dict[str, Any],
                   rank: int = None, world: int = None, gather: bool = True) -> torch.Tensor:
    """Run ``sample_fn(shard_kwargs, count, sample_index0) -> [count, ...]`` on this rank's shard and
    all_gather the shards in rank order -> [total, ...] on every rank (one collective, at the end)."""
    if rank is None or world is None:
        rank, world, _ = env_rank_world()
    start, count = shard_range(total, rank, world)
    local = sample_fn(shard_kwargs(model_kwargs, start, count, total), count, start)
    if world == 1 or not gather:
        return local
    counts = [shard_range(total, r, world)[1] for r in range(world)]
    mx = max(counts)
    pad = local if count == mx else torch.cat([local, local.new_zeros((mx - count,) + tuple(local.shape[1:]))], 0)
    bufs = all_gather_rows(pad.contiguous(), world)
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], 0)


def adm_to_amdm_condition(sample: torch.Tensor, sigma: float = 0.8, mean: float = 0.0, std: float = 1.0) -> torch.Tensor:
    """In-process ADM -> AMDM hand-off (the reference goes through .npy files): denormalise + clip to
    [1e-20, 1] (datasets/humanml3d.py:494-511), dist = sqrt(-2 ln(c) sigma^2) (utils/evaluate.py:56-66),
    consumer c = exp(-dist^2 / (2 sigma^2)) (datasets/humanml3d.py:773-774) - one HIP kernel (afm_contact_glue), the sample never leaves HBM."""
    from . import ffi
    ffi.require_gpu(sample)
    x = ffi.f32c(sample)
    out = torch.empty_like(x)
    ffi.check(ffi.load().afm_contact_glue(x.data_ptr(), out.data_ptr(), x.numel(), float(sigma) ** 2, float(mean), float(std), ffi.stream_of(x)),
              "afm_contact_glue")
    return out
