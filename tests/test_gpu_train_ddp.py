"""`-m gpu`: the reference's DDP recipe (train_ddp.py:63-65: SyncBatchNorm.convert_sync_batchnorm + DistributedDataParallel) over
our modules.  Two ranks share cuda:0 (the GPU box has one GPU; RCCL refuses two ranks on one device, so the process group is
gloo, which moves the CUDA tensors through the host) - what is tested is the math: synchronised batch statistics and
averaged gradients must equal one process running the concatenated batch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make(dev, blocks):
    from afm import scene as S
    from gpu_util import load_named_weights
    enc = S.SceneMapEncoder(point_feat_dim=6, planes=[32, 64, 128, 256], blocks=list(blocks), num_points=1024)
    load_named_weights(enc)
    return enc.to(dev).train()


def _data():
    from afm import synth
    xyz, con = synth.scene_cloud(4, 1024, seed=21), synth.contact_map(4, 1024, seed=21)
    dy = synth.gaussian("ddp_dy", (4, 16, 256))
    return xyz, con, dy


def _rank(rank, world, port, out_path, blocks):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    enc = torch.nn.SyncBatchNorm.convert_sync_batchnorm(_make(dev, blocks))
    ddp = torch.nn.parallel.DistributedDataParallel(enc, find_unused_parameters=True, broadcast_buffers=False)
    xyz, con, dy = _data()
    sl = slice(rank * 2, rank * 2 + 2)
    out = ddp(xyz[sl].to(dev), con[sl].to(dev))
    ((out * dy[sl].to(dev)).sum() / 2).backward()                # per-rank mean over its 2 samples; DDP averages the ranks
    if rank == 0:
        torch.save({"out": out.detach().cpu(), "grads": {n: p.grad.cpu() for n, p in enc.named_parameters()},
                    "rm": enc.enc1[0].bn.running_mean.cpu(), "rv": enc.enc1[0].bn.running_var.cpu()}, out_path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("blocks,tight_tol,loose_tol", [((2, 2, 2, 1), 2e-5, 1e-4), ((2, 2, 2, 2), 5e-3, 2e-2)])
def test_syncbn_ddp_two_ranks_equal_one_process(tmp_path, blocks, tight_tol, loose_tol):
    from test_gpu_train import _zero_grad_name
    path = str(tmp_path / "rank0.pt")
    mp.spawn(_rank, args=(2, _free_port(), path, blocks), nprocs=2, join=True)
    got = torch.load(path)
    dev = torch.device("cuda:0")
    enc = _make(dev, blocks)
    xyz, con, dy = _data()
    out = enc(xyz.to(dev), con.to(dev))
    ((out * dy.to(dev)).sum() / 4).backward()
    err = (out[:2].detach().cpu() - got["out"]).abs().max().item()
    print(f"[parity] SyncBatchNorm forward, 2 ranks vs 1 process: max|diff|={err:.3e}")
    assert err <= 1e-4
    # Criterion: relative L2 error per parameter tensor.  Two evaluations of the same network that differ only in summation
    # order (here: per-rank partial statistics) can flip a ReLU whose pre-activation is within rounding of 0.  With the
    # reference's block layout (2,2,2,2) and this input exactly one element of the last level (64 rows) flips: the few
    # gradients it feeds move by ~1e-2 and everything upstream by ~1e-3, so that case gets a loose bound; the (2,2,2,1) layout
    # has no flip: measured in round 5 median 6.0e-6, worst tensor 2.0e-5 - the bound is back at 1e-4 for the worst tensor (round 3 had loosened it to
    # 5e-4) and 2e-5 for the median; the synchronised statistics and gradient averaging themselves are exact.  The (2,2,2,2)
    # bounds stay where they were and are now justified by the measurement below: the SAME batch in reverse sample order on ONE process moves the
    # same tensors by 3.5e-3 (worst) - the flip belongs to the network and this input, not to the two ranks (2 ranks vs 1 process: 1.4e-3 worst).
    errs = []
    for n, p in enc.named_parameters():
        if _zero_grad_name(n):
            continue
        ref = p.grad.cpu().double()
        errs.append(((got["grads"][n].double() - ref).norm().item() / max(ref.norm().item(), 1e-9), n))
    # What "the same mathematics in another summation order" costs ON ONE PROCESS (justify the bound per tensor): the same
    # batch with its samples in reverse order - BatchNorm statistics, weight gradients and every reduction over the concatenated rows then add
    # the same numbers in another order, exactly what two ranks' partial sums do.  A tensor may differ between 2 ranks and 1 process by a small
    # multiple of what it differs between these two single-process runs (or 1e-4, whichever is larger).
    enc2 = _make(dev, blocks)
    out2 = enc2(xyz.flip(0).contiguous().to(dev), con.flip(0).contiguous().to(dev))
    ((out2 * dy.flip(0).contiguous().to(dev)).sum() / 4).backward()
    enc3 = _make(dev, blocks)
    out3 = enc3(xyz.to(dev), con.to(dev))
    ((out3 * dy.to(dev)).sum() / 4).backward()
    noise = {}
    for (n, p2), (_, p3) in zip(enc2.named_parameters(), enc3.named_parameters()):
        ref = p3.grad.cpu().double()
        noise[n] = (p2.grad.cpu().double() - ref).norm().item() / max(ref.norm().item(), 1e-9)
    errs.sort(reverse=True)
    median = errs[len(errs) // 2][0]
    worst_noise = sorted(((v, n) for n, v in noise.items() if not _zero_grad_name(n)), reverse=True)[:2]
    print(f"[parity] blocks={blocks}: summation-order noise of ONE process (samples reversed): largest {[(round(v, 6), n) for v, n in worst_noise]}")
    if tight_tol <= 1e-4:            # the layout without a ReLU flip: every tensor within 1e-4 or 4 x its own single-process order noise
        over = [(e, noise[n], n) for e, n in errs if e > max(1e-4, 4.0 * noise[n])]
        assert not over, over[:5]
    print(f"[parity] blocks={blocks}: DDP-averaged gradients vs single-process full batch over {len(errs)} tensors: rel-L2 median {median:.2e}, "
          f"largest {[(round(e, 6), n) for e, n in errs[:2]]}")
    # the median bound is not one box's number either - at least `tight_tol`, or 4 x the median single-process order noise of the same
    # tensors (gloo vs RCCL, one GPU vs two change the per-rank summation order, not its size)
    noise_median = sorted(v for n, v in noise.items() if not _zero_grad_name(n))[len(errs) // 2]
    assert median <= max(tight_tol, 4.0 * noise_median) and errs[0][0] <= loose_tol, (median, noise_median, errs[:5])
    assert (enc.enc1[0].bn.running_mean.cpu() - got["rm"]).abs().max().item() <= 1e-5
    assert (enc.enc1[0].bn.running_var.cpu() - got["rv"]).abs().max().item() <= 1e-4
