"""Differentiable point-cloud operators of the training path (SceneMapEncoder under model.train(),
reference models/scene_models/pointtransformer.py:26-69,102-123): BatchNorm on batch statistics (optionally synchronised
across ranks = nn.SyncBatchNorm of train_ddp.py), neighbour gather / grouping, group max-pool and the vector-attention
glue, each a torch.autograd Function whose forward and backward are HIP kernels (csrc/pointnet_train.hip).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch
import torch.distributed as dist

from . import ffi
from .autograd import _c, _st


def _ws(nbytes: int, dev) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=dev)


_ONES = {}


def _ones(n: int, dev) -> torch.Tensor:
    """Cached [n] tensor of ones per device (a fill, once): the gamma that makes afm_bn_fold return rstd."""
    key = (n, str(dev))
    if key not in _ONES:
        _ONES[key] = torch.ones(n, device=dev, dtype=torch.float32)
    return _ONES[key]


def _sync_group():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def _host_sync_needed(t: torch.Tensor) -> None:
    """RCCL collectives are stream-ordered with our kernels (torch makes its communication stream wait on the current
    stream).  The gloo backend copies CUDA tensors through the host instead; fence the device around it."""
    import os
    if dist.get_backend() != "nccl" or os.environ.get("AFM_DEBUG_SYNC"):
        torch.cuda.current_stream(t.device).synchronize()


# ------------------------------------------------------------------------------------------------ BatchNorm
class _BatchNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, residual, bn, relu, sync):
        lib = ffi.load()
        xc, g, b = _c(x), _c(gamma), _c(beta)
        res = None if residual is None else _c(residual)
        Cn = xc.shape[-1]
        rows = xc.numel() // Cn
        dev = xc.device
        mean, rstd, scale, shift = (torch.empty(Cn, device=dev, dtype=torch.float32) for _ in range(4))
        count = rows
        batch_stats = bn.training or bn.running_mean is None
        if batch_stats:
            stats = torch.empty(3 * Cn, device=dev, dtype=torch.float32)
            ws = _ws(lib.afm_colstats_workspace_bytes(rows, Cn), dev)
            ffi.check(lib.afm_colstats(xc.data_ptr(), rows, Cn, stats.data_ptr(), ws.data_ptr(), ws.numel(), _st(xc)), "afm_colstats")
            world = 1
            if sync and _sync_group():
                world = dist.get_world_size()
                gathered = torch.empty(world, 3 * Cn, device=dev, dtype=torch.float32)
                _host_sync_needed(stats)
                dist.all_gather_into_tensor(gathered, stats) if dist.get_backend() == "nccl" else \
                    dist.all_gather(list(gathered.unbind(0)), stats)     # RCCL: 3*C floats per rank per BatchNorm
                stats = gathered
                count = rows * world
            track = bn.training and bn.track_running_stats and bn.running_mean is not None
            mom = 0.1 if bn.momentum is None else float(bn.momentum)
            ffi.check(lib.afm_bn_finalize(stats.data_ptr(), world, rows, g.data_ptr(), b.data_ptr(), float(bn.eps), mom,
                                          bn.running_mean.data_ptr() if track else None, bn.running_var.data_ptr() if track else None,
                                          mean.data_ptr(), rstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), Cn, _st(xc)), "afm_bn_finalize")
            if track and bn.num_batches_tracked is not None:
                bn.num_batches_tracked += 1
        else:                                                            # frozen BatchNorm (eval): running statistics
            # afm_bn_fold (HIP): scale = gamma rstd, shift = beta - mean scale; with gamma = 1 the same kernel yields rstd itself -
            # no eager ATen arithmetic on this path either
            mean = ffi.f32c(bn.running_mean.detach())
            var = ffi.f32c(bn.running_var.detach())
            ones = _ones(Cn, dev)
            st = _st(xc)
            ffi.check(lib.afm_bn_fold(g.data_ptr(), b.data_ptr(), mean.data_ptr(), var.data_ptr(), float(bn.eps), None, scale.data_ptr(),
                                      shift.data_ptr(), Cn, st), "afm_bn_fold")
            unused_shift = torch.empty_like(rstd)
            ffi.check(lib.afm_bn_fold(ones.data_ptr(), ones.data_ptr(), mean.data_ptr(), var.data_ptr(), float(bn.eps), None, rstd.data_ptr(),
                                      unused_shift.data_ptr(), Cn, st), "afm_bn_fold")
        y = torch.empty_like(xc)
        ffi.check(lib.afm_colaffine(xc.data_ptr(), scale.data_ptr(), shift.data_ptr(), ffi.ptr(res), 1 if relu else 0, y.data_ptr(), rows, Cn,
                                    _st(xc)), "afm_colaffine")
        ctx.save_for_backward(xc, y if relu else None, mean, rstd, g)
        ctx.cfg = (rows, Cn, count, batch_stats, sync, residual is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = ffi.load()
        xc, y, mean, rstd, g = ctx.saved_tensors
        rows, Cn, count, batch_stats, sync, has_res = ctx.cfg
        dy = _c(dy)
        dev = xc.device
        stats = torch.empty(2 * Cn, device=dev, dtype=torch.float32)
        ws = _ws(lib.afm_colstats_workspace_bytes(rows, Cn), dev)
        ffi.check(lib.afm_bn_bwd_stats(dy.data_ptr(), xc.data_ptr(), ffi.ptr(y), mean.data_ptr(), rstd.data_ptr(), rows, Cn, stats.data_ptr(),
                                       ws.data_ptr(), ws.numel(), _st(xc)), "afm_bn_bwd_stats")
        dbeta, dgamma = stats[:Cn].clone(), stats[Cn:].clone()           # local sums (DDP averages parameter grads itself)
        if not batch_stats:
            stats = torch.zeros_like(stats)                              # statistics are constants: dx = g * gamma * rstd
        elif sync and _sync_group():
            _host_sync_needed(stats)
            dist.all_reduce(stats)
        dx = torch.empty_like(xc)
        dres = torch.empty_like(xc) if has_res else None
        ffi.check(lib.afm_bn_bwd_apply(dy.data_ptr(), xc.data_ptr(), ffi.ptr(y), mean.data_ptr(), rstd.data_ptr(), g.data_ptr(), stats.data_ptr(),
                                       count, dx.data_ptr(), ffi.ptr(dres), rows, Cn, _st(xc)), "afm_bn_bwd_apply")
        return dx, dgamma, dbeta, dres, None, None, None


def batch_norm(x, bn: torch.nn.BatchNorm1d, *, relu: bool = False, residual: Optional[torch.Tensor] = None, sync: Optional[bool] = None):
    """relu?(BatchNorm1d(x) + residual) over a row-major [rows, C] matrix; batch statistics when ``bn.training``, running
    statistics otherwise.  ``sync`` (default: the module is an nn.SyncBatchNorm, i.e. the model went through
    `SyncBatchNorm.convert_sync_batchnorm` as in train_ddp.py:63) all-reduces the statistics over the process group."""
    if sync is None:
        sync = isinstance(bn, torch.nn.SyncBatchNorm)
    return _BatchNormFn.apply(x, bn.weight, bn.bias, residual, bn, relu, sync)


# ------------------------------------------------------------------------------------------------ gather / group
# Deterministic scatter-adds (round 6): the backward of every row gather is a segmented sum over the INVERSE of its index list
# (afm_scatter_plan: for every destination row its entries, ascending), built once per index tensor and shared by every operator that
# scatters through it (the k / v / position gathers and the grouping of a layer use ONE kNN list) - no f32 atomics, so two training steps on
# the same data produce the same bits.  The cache is keyed by (address, size, destinations) and HOLDS the index tensor (autograd hands the
# backward a new Python object for the same storage, so object identity would miss every time; while the entry lives the address cannot be
# recycled for other contents), checks the tensor's version counter, and keeps the most recent 48 lists.
_PLANS: dict = {}


def scatter_plan(idx: torch.Tensor, n_dst: int) -> torch.Tensor:
    """int32 plan of afm_scatter_plan for `idx` (any shape, flattened) over `n_dst` destination rows."""
    key = (idx.data_ptr(), idx.numel(), int(n_dst), str(idx.device))
    hit = _PLANS.get(key)
    if hit is not None and hit[1] == idx._version:
        return hit[2]
    lib = ffi.load()
    words = lib.afm_scatter_plan_words(idx.numel(), int(n_dst))
    if words < 0:
        ffi.check(int(words), "afm_scatter_plan_words")
    plan = torch.empty(int(words), dtype=torch.int32, device=idx.device)
    ffi.check(lib.afm_scatter_plan(idx.data_ptr(), idx.numel(), int(n_dst), plan.data_ptr(), _st(idx)), "afm_scatter_plan")
    _PLANS.pop(key, None)
    while len(_PLANS) >= 48:                           # (dicts keep insertion order: the oldest list goes first)
        _PLANS.pop(next(iter(_PLANS)))
    _PLANS[key] = (idx, idx._version, plan)
    return plan


def _segment_sum(src: torch.Tensor, col_offset: int, idx: torch.Tensor, n_dst: int, C_: int, row_div: int = 1, d2: Optional[torch.Tensor] = None) -> torch.Tensor:
    plan = scatter_plan(idx, n_dst)
    dst = torch.empty(n_dst, C_, device=src.device, dtype=torch.float32)
    ffi.check(ffi.load().afm_segment_sum_rows(src.data_ptr(), src.shape[1], col_offset, row_div, ffi.ptr(d2), plan.data_ptr(), dst.data_ptr(), n_dst, C_,
                                              _st(src)), "afm_segment_sum_rows")
    return dst


class _GatherFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, idx):
        xc = _c(x)
        out = torch.empty(idx.numel(), xc.shape[1], device=xc.device, dtype=torch.float32)
        ffi.check(ffi.load().afm_gather_rows(xc.data_ptr(), idx.data_ptr(), out.data_ptr(), idx.numel(), xc.shape[1], _st(xc)), "afm_gather_rows")
        ctx.save_for_backward(idx)
        ctx.shape = tuple(xc.shape)
        return out

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        dy = _c(dy)
        return _segment_sum(dy, 0, idx, ctx.shape[0], dy.shape[1]), None


def gather(x, idx):
    """x[idx] for int32 row indices (any shape, flattened) -> [idx.numel(), C]."""
    return _GatherFn.apply(x, idx)


def broadcast_rows(ctx_rows: torch.Tensor, n: int) -> torch.Tensor:
    """[B, C] -> [B * n, C]: row b repeated for the n points of sample b (the `repeat` of a per-sample context onto its points,
    cdm.py:236-243, pointtransformer.py:90-92); backward = the scatter-add of the row gather."""
    return _BroadcastRowsFn.apply(ctx_rows, n)


class _BroadcastRowsFn(torch.autograd.Function):
    """Row b of [B, C] repeated n times; backward = the sum of every sample's n consecutive rows (afm_group_sum: one fixed-order reduction per
    sample and channel - a segment of n rows is not a job for the inverse-index plan of the general gather)."""

    @staticmethod
    def forward(ctx, rows, n):
        rc = _c(rows)
        B = rc.shape[0]
        idx = torch.arange(B, device=rc.device, dtype=torch.int32).repeat_interleave(n)
        out = torch.empty(B * n, rc.shape[1], device=rc.device, dtype=torch.float32)
        ffi.check(ffi.load().afm_gather_rows(rc.data_ptr(), idx.data_ptr(), out.data_ptr(), idx.numel(), rc.shape[1], _st(rc)), "afm_gather_rows")
        ctx.dims = (B, n, rc.shape[1])
        return out

    @staticmethod
    def backward(ctx, dy):
        B, n, Cn = ctx.dims
        dy = _c(dy)
        d = torch.empty(B, Cn, device=dy.device, dtype=torch.float32)
        ffi.check(ffi.load().afm_group_sum(dy.data_ptr(), d.data_ptr(), B, n, Cn, 1.0, _st(dy)), "afm_group_sum")
        return d, None


class _InterpolateFn(torch.autograd.Function):
    """pointops.interpolation (pointops.py:164-178) + the fused `base +` of TransitionUp (pointtransformer.py:98); idx / d2 [n, k] from afm_knn."""

    @staticmethod
    def forward(ctx, feat, base, idx, d2):
        fc = _c(feat)
        n, k = idx.shape
        out = torch.empty(n, fc.shape[1], device=fc.device, dtype=torch.float32)
        b = None if base is None else _c(base)
        ffi.check(ffi.load().afm_interpolate(fc.data_ptr(), idx.data_ptr(), d2.data_ptr(), ffi.ptr(b), out.data_ptr(), n, fc.shape[1], k, _st(fc)),
                  "afm_interpolate")
        ctx.save_for_backward(idx, d2)
        ctx.dims = (fc.shape[0], fc.shape[1], base is not None)
        return out

    @staticmethod
    def backward(ctx, dout):
        idx, d2 = ctx.saved_tensors
        m, c, has_base = ctx.dims
        dout = _c(dout)
        n, k = idx.shape
        dfeat = _segment_sum(dout, 0, idx, m, c, row_div=k, d2=d2)      # deterministic form of afm_interpolate_bwd (same weights, fixed order)
        return dfeat, (dout if has_base else None), None, None


def interpolate(xyz_src, xyz_dst, feat, batch: int, m: int, n: int, base=None, k: int = 3):
    """Differentiable form of afm.pointops.interpolate (the neighbour search itself carries no gradient)."""
    from . import pointops
    with torch.no_grad():
        idx, d2 = pointops.knn(k, xyz_src, xyz_dst, batch, m, n)
    return _InterpolateFn.apply(feat, base, idx, d2)


class _GroupPointsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, new_xyz, feat, idx, k):
        xyz, new_xyz = _c(xyz), _c(new_xyz)
        f = None if feat is None else _c(feat)
        Cn = 0 if f is None else f.shape[1]
        rows = idx.numel()
        out = torch.empty(rows, 3 + Cn, device=xyz.device, dtype=torch.float32)
        ffi.check(ffi.load().afm_group_points(xyz.data_ptr(), new_xyz.data_ptr(), ffi.ptr(f), idx.data_ptr(), out.data_ptr(), rows, k, Cn,
                                              _st(xyz)), "afm_group_points")
        ctx.save_for_backward(idx)
        ctx.fshape = None if f is None else tuple(f.shape)
        return out

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        dfeat = None
        if ctx.fshape is not None and ctx.needs_input_grad[2]:
            dy = _c(dy)
            dfeat = _segment_sum(dy, 3, idx, ctx.fshape[0], ctx.fshape[1])
        return None, None, dfeat, None, None


def group_points(xyz, new_xyz, feat, idx, k: int):
    """[xyz[idx] - new_xyz | feat[idx]] -> [m*k, 3 + C]  (pointops.queryandgroup with use_xyz=True; feat may be None)."""
    return _GroupPointsFn.apply(xyz, new_xyz, feat, idx, k)


class _GroupMaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k):
        xc = _c(x)
        Cn = xc.shape[-1]
        m = xc.numel() // (k * Cn)
        y = torch.empty(m, Cn, device=xc.device, dtype=torch.float32)
        arg = torch.empty(m, Cn, device=xc.device, dtype=torch.int32)
        ffi.check(ffi.load().afm_group_max(xc.data_ptr(), y.data_ptr(), arg.data_ptr(), m, k, Cn, _st(xc)), "afm_group_max")
        ctx.save_for_backward(arg)
        ctx.dims = (m, k, Cn)
        return y

    @staticmethod
    def backward(ctx, dy):
        (arg,) = ctx.saved_tensors
        m, k, Cn = ctx.dims
        dy = _c(dy)
        dx = torch.empty(m * k, Cn, device=dy.device, dtype=torch.float32)
        ffi.check(ffi.load().afm_group_max_bwd(dy.data_ptr(), arg.data_ptr(), dx.data_ptr(), m, k, Cn, _st(dy)), "afm_group_max_bwd")
        return dx, None


def group_max(x, k: int):
    """nn.MaxPool1d(k) over the neighbour axis: [m*k, C] -> [m, C]."""
    return _GroupMaxFn.apply(x, k)


# ------------------------------------------------------------------------------------------------ vector attention glue
class _PtW0Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, kg, q, pr, k):
        kg, q, pr = _c(kg), _c(q), _c(pr)
        m, Cn = q.shape
        out = torch.empty_like(kg)
        ffi.check(ffi.load().afm_pt_w0(kg.data_ptr(), q.data_ptr(), pr.data_ptr(), out.data_ptr(), m, k, Cn, _st(kg)), "afm_pt_w0")
        ctx.dims = (m, k, Cn)
        return out

    @staticmethod
    def backward(ctx, d):
        m, k, Cn = ctx.dims
        d = _c(d)
        dq = torch.empty(m, Cn, device=d.device, dtype=torch.float32)
        ffi.check(ffi.load().afm_group_sum(d.data_ptr(), dq.data_ptr(), m, k, Cn, -1.0, _st(d)), "afm_group_sum")
        return d, dq, d, None


def pt_w0(kg, q, pr, k: int):
    """k_g - q[:, None] + p_r over [m*k, C] (pointtransformer.py:34)."""
    return _PtW0Fn.apply(kg, q, pr, k)


class _PtAggregateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vg, pr, w2, k, share):
        vg, pr, w2 = _c(vg), _c(pr), _c(w2)
        Cn = vg.shape[1]
        m = vg.shape[0] // k
        out = torch.empty(m, Cn, device=vg.device, dtype=torch.float32)
        sw = torch.empty_like(w2)
        ffi.check(ffi.load().afm_pt_aggregate(vg.data_ptr(), pr.data_ptr(), w2.data_ptr(), out.data_ptr(), sw.data_ptr(), m, k, Cn, share, _st(vg)),
                  "afm_pt_aggregate")
        ctx.save_for_backward(vg, pr, sw)
        ctx.dims = (m, k, Cn, share)
        return out

    @staticmethod
    def backward(ctx, dout):
        vg, pr, sw = ctx.saved_tensors
        m, k, Cn, share = ctx.dims
        dout = _c(dout)
        da = torch.empty_like(vg)
        dw2 = torch.empty_like(sw)
        ffi.check(ffi.load().afm_pt_aggregate_bwd(vg.data_ptr(), pr.data_ptr(), sw.data_ptr(), dout.data_ptr(), da.data_ptr(), dw2.data_ptr(), m, k,
                                                  Cn, share, _st(vg)), "afm_pt_aggregate_bwd")
        return da, da, dw2, None, None


def pt_aggregate(vg, pr, w2, k: int, share: int):
    """sum_k (v_g + p_r) * softmax_k(w2) with the weights shared by `share` channel groups (pointtransformer.py:35-37)."""
    return _PtAggregateFn.apply(vg, pr, w2, k, share)
