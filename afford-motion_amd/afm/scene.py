"""Point-cloud branch of the hot path: TransitionDown ("set abstraction"), PointTransformerLayer /
Block and SceneMapEncoder (reference models/scene_models/pointtransformer.py:9-123,
models/modules.py:124-167).  Inference (`run`, no autograd) uses the fused kernels below; with autograd enabled and
trainable parameters the `run_train` methods compose the same functions from differentiable HIP passes
(afm/autograd_points.py), BatchNorm on batch statistics when the module is in train mode.

The nn.Modules below are parameter containers with the reference's state-dict keys; the math runs
in the HIP point kernels (afm/pointops.py -> csrc/pointops.hip, csrc/pointnet.hip):
  * furthest point sampling and brute-force kNN replace the external `pointops_cuda`;
  * the gather is never materialised: gather + Linear + BN + ReLU + max-pool is one kernel,
    and the whole vector-attention layer (gather, position MLP, weight MLP, softmax over the k
    neighbours, weighted sum) is one kernel;
  * kNN is computed once per resolution level and shared by every layer of that level (the
    reference recomputes the identical query twice per layer, pointtransformer.py:29-30);
  * every sample has the same number of points, so batch offsets are implicit.
"""
from __future__ import annotations

from typing import List

import torch
import torch.nn as nn

from . import autograd as AG
from . import autograd_points as AP
from . import ffi, ops, pointops


def _cached(module: nn.Module, slot: str, tensors, build):
    """Per-module cache of a derived tensor pack, rebuilt when one of `tensors` was written in place, replaced or moved."""
    key = tuple((id(t), t._version, t.data_ptr()) for t in tensors)
    hit = module.__dict__.get(slot)
    if hit is None or hit[0] != key:
        hit = (key, build(), tuple(tensors))            # the tensors are held: a freed one's id / address cannot come back and match
        module.__dict__[slot] = hit
    return hit[1]


def _bn_fold(bn: nn.BatchNorm1d, lin_bias: torch.Tensor = None):
    """eval-mode BatchNorm as y = x * scale + shift (optionally with a preceding nn.Linear's bias folded in: BN(Wx + lb) = scale Wx +
    (lb scale + shift)).  Computed by afm_bn_fold (HIP) once per weight version and cached on the module."""
    srcs = [bn.weight, bn.bias, bn.running_mean, bn.running_var] + ([lin_bias] if lin_bias is not None else [])

    def build():
        ffi.require_gpu(bn.weight)
        w, b, m, v = (ffi.f32c(t.detach()) for t in srcs[:4])
        lb = None if lin_bias is None else ffi.f32c(lin_bias.detach())
        scale, shift = torch.empty_like(w), torch.empty_like(w)
        ffi.check(ffi.load().afm_bn_fold(w.data_ptr(), b.data_ptr(), m.data_ptr(), v.data_ptr(), float(bn.eps), ffi.ptr(lb), scale.data_ptr(),
                                         shift.data_ptr(), w.numel(), ffi.stream_of(w)), "afm_bn_fold")
        return scale, shift
    return _cached(bn, "_afm_fold" + ("_lb" if lin_bias is not None else ""), srcs, build)


class PointTransformerLayer(nn.Module):
    def __init__(self, in_planes, out_planes, share_planes=8, nsample=16):
        super().__init__()
        self.mid_planes = mid_planes = out_planes // 1
        self.out_planes, self.share_planes, self.nsample = out_planes, share_planes, nsample
        self.linear_q = nn.Linear(in_planes, mid_planes)
        self.linear_k = nn.Linear(in_planes, mid_planes)
        self.linear_v = nn.Linear(in_planes, out_planes)
        self.linear_p = nn.Sequential(nn.Linear(3, 3), nn.BatchNorm1d(3), nn.ReLU(inplace=True), nn.Linear(3, out_planes))
        self.linear_w = nn.Sequential(nn.BatchNorm1d(mid_planes), nn.ReLU(inplace=True),
                                      nn.Linear(mid_planes, mid_planes // share_planes),
                                      nn.BatchNorm1d(mid_planes // share_planes), nn.ReLU(inplace=True),
                                      nn.Linear(out_planes // share_planes, out_planes // share_planes))

    def run(self, p: torch.Tensor, x: torch.Tensor, knn_idx: torch.Tensor, out_scale=None, out_shift=None, relu=False):
        """p [n,3], x [n,c], knn_idx [n,k] (global rows) -> [n,c]; optional fused y*scale+shift (+ReLU)."""
        c = self.out_planes
        srcs = [self.linear_q.weight, self.linear_k.weight, self.linear_v.weight, self.linear_q.bias, self.linear_k.bias, self.linear_v.bias]
        wqkv, bqkv = _cached(self, "_afm_qkv", srcs, lambda: (torch.cat([t.detach() for t in srcs[:3]], 0), torch.cat([t.detach() for t in srcs[3:]], 0)))
        qkv = ops.linear(x, wqkv, bqkv)                                            # [n, 3c]: one GEMM on the packed q | k | v weights (packed once per weight version)
        ps, pb = _bn_fold(self.linear_p[1])
        w0s, w0b = _bn_fold(self.linear_w[0])
        w3s, w3b = _bn_fold(self.linear_w[3])
        return pointops.pt_attention(
            p, qkv, knn_idx, c, self.share_planes,
            self.linear_p[0].weight, self.linear_p[0].bias, ps, pb, self.linear_p[3].weight, self.linear_p[3].bias,
            w0s, w0b, self.linear_w[2].weight, self.linear_w[2].bias, w3s, w3b,
            self.linear_w[5].weight, self.linear_w[5].bias, out_scale, out_shift, relu)


    def run_train(self, p, x, knn_idx):
        """Differentiable form of `run` (pointtransformer.py:26-38): every BatchNorm is its own pass."""
        k, c, n = self.nsample, self.out_planes, x.shape[0]
        idx = knn_idx.reshape(-1)
        xq = AG.linear(x, self.linear_q.weight, self.linear_q.bias)
        kg = AP.gather(AG.linear(x, self.linear_k.weight, self.linear_k.bias), idx)        # [n*k, c]
        vg = AP.gather(AG.linear(x, self.linear_v.weight, self.linear_v.bias), idx)
        pr = AP.group_points(p, p, None, idx, k)                                           # [n*k, 3] relative xyz
        pr = AG.linear(pr, self.linear_p[0].weight, self.linear_p[0].bias)
        pr = AP.batch_norm(pr, self.linear_p[1], relu=True)
        pr = AG.linear(pr, self.linear_p[3].weight, self.linear_p[3].bias)                 # [n*k, c]
        w = AP.pt_w0(kg, xq, pr, k)
        w = AP.batch_norm(w, self.linear_w[0], relu=True)
        w = AG.linear(w, self.linear_w[2].weight, self.linear_w[2].bias)
        w = AP.batch_norm(w, self.linear_w[3], relu=True)
        w = AG.linear(w, self.linear_w[5].weight, self.linear_w[5].bias)                   # [n*k, c/s]
        return AP.pt_aggregate(vg, pr, w, k, self.share_planes)


class TransitionDown(nn.Module):
    def __init__(self, in_planes, out_planes, stride=1, nsample=16):
        super().__init__()
        self.stride, self.nsample = stride, nsample
        if stride != 1:
            self.linear = nn.Linear(3 + in_planes, out_planes, bias=False)
            self.pool = nn.MaxPool1d(nsample)
        else:
            self.linear = nn.Linear(in_planes, out_planes, bias=False)
        self.bn = nn.BatchNorm1d(out_planes)
        self.relu = nn.ReLU(inplace=True)

    def run(self, p: torch.Tensor, x: torch.Tensor, batch: int, geo=None):
        """p [B*n,3], x [B*n,c] -> (p', x') with n' = n // stride points per sample; `geo` = a precomputed `geometry(p, batch)`."""
        scale, shift = _bn_fold(self.bn)
        if self.stride == 1:
            return p, ops.linear(x, self.linear.weight, shift, scale=scale, act=ffi.ACT_RELU)
        if geo is not None:
            n_p, knn_idx = geo
        else:
            n = p.shape[0] // batch
            m = n // self.stride
            idx = pointops.furthest_point_sampling(p, batch, n, m)                      # [B*m] global rows
            n_p = pointops.gather_rows(p, idx)                                         # [B*m, 3]
            knn_idx, _ = pointops.knn(self.nsample, p, n_p, batch, n, m)                # [B*m, k]
        y = pointops.transition_down(p, x, n_p, knn_idx, self.linear.weight, scale, shift)
        return n_p, y


    def geometry(self, p, batch: int):
        """The part of the layer that depends on the coordinates only: (sampled points, their neighbours among the input points)."""
        n = p.shape[0] // batch
        m = n // self.stride
        with torch.no_grad():
            idx = pointops.furthest_point_sampling(p, batch, n, m)
            n_p = pointops.gather_rows(p, idx)
            knn_idx, _ = pointops.knn(self.nsample, p, n_p, batch, n, m)
        return n_p, knn_idx

    def run_train(self, p, x, batch: int, geo=None):
        """Differentiable form of `run` (pointtransformer.py:53-69); `geo` = a precomputed `geometry(p, batch)`."""
        if self.stride == 1:
            return p, AP.batch_norm(AG.linear(x, self.linear.weight), self.bn, relu=True)
        n_p, knn_idx = geo if geo is not None else self.geometry(p, batch)
        g = AP.group_points(p, n_p, x, knn_idx.reshape(-1), self.nsample)                  # [m*k, 3+c]
        y = AP.batch_norm(AG.linear(g, self.linear.weight), self.bn, relu=True)            # BN over all m*k rows
        return n_p, AP.group_max(y, self.nsample)


class PointTransformerBlock(nn.Module):
    expansion = 1

    def __init__(self, in_planes, planes, share_planes=8, nsample=16):
        super().__init__()
        self.linear1 = nn.Linear(in_planes, planes, bias=False)
        self.bn1 = nn.BatchNorm1d(planes)
        self.transformer2 = PointTransformerLayer(planes, planes, share_planes, nsample)
        self.bn2 = nn.BatchNorm1d(planes)
        self.linear3 = nn.Linear(planes, planes * self.expansion, bias=False)
        self.bn3 = nn.BatchNorm1d(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)

    def run(self, p, x, knn_idx):
        s1, b1 = _bn_fold(self.bn1)
        s2, b2 = _bn_fold(self.bn2)
        s3, b3 = _bn_fold(self.bn3)
        y = ops.linear(x, self.linear1.weight, b1, scale=s1, act=ffi.ACT_RELU)
        y = self.transformer2.run(p, y, knn_idx, out_scale=s2, out_shift=b2, relu=True)
        return ops.linear(y, self.linear3.weight, b3, scale=s3, residual=x, act_post=ffi.ACT_RELU)   # relu(bn3(linear3) + identity)


    def run_train(self, p, x, knn_idx):
        """Differentiable form of `run` (pointtransformer.py:115-123)."""
        y = AP.batch_norm(AG.linear(x, self.linear1.weight), self.bn1, relu=True)
        y = AP.batch_norm(self.transformer2.run_train(p, y, knn_idx), self.bn2, relu=True)
        return AP.batch_norm(AG.linear(y, self.linear3.weight), self.bn3, relu=True, residual=x)


class SceneMapEncoder(nn.Module):
    """[xyz | per-point feature] -> N/64 group tokens of width planes[-1] (modules.py:124-167)."""

    def __init__(self, point_feat_dim: int, planes: List, blocks: List, num_points: int = 8192) -> None:
        super().__init__()
        self.num_points = num_points
        self.c = point_feat_dim + 3
        self.in_planes = self.c
        share_planes = 8
        self.strides, self.nsamples = [1, 4, 4, 4], [8, 16, 16, 16]
        # inference: FPS / kNN of all levels on a side stream under the feature passes (the training path's geometry pyramid; bit-identical).  Measured in
        # round 6 and NOT faster - 6.18-6.23 ms against 6.17-6.31 ms for 32 scenes of 8192 points: in inference the geometry chain (FPS 1.9 + kNN 2.1 ms) is
        # longer than all feature kernels together (2.2 ms), and an FPS workgroup that shares its CU with attention waves loses the issue slots it is bound by
        self.overlap_geometry = False
        for i in range(4):
            setattr(self, f"enc{i + 1}", self._make_enc(planes[i], blocks[i], share_planes, self.strides[i], self.nsamples[i]))

    @property
    def num_groups(self):
        return self.num_points // 64

    def _make_enc(self, planes, blocks, share_planes, stride, nsample):
        layers = [TransitionDown(self.in_planes, planes, stride, nsample)]
        self.in_planes = planes
        for _ in range(1, blocks):
            layers.append(PointTransformerBlock(planes, planes, share_planes, nsample=nsample))
        return nn.Sequential(*layers)

    def _geometry_pyramid(self, p0: torch.Tensor, B: int):
        """Farthest-point sampling and the neighbour lists of all four levels depend on the coordinates only (pointtransformer.py:53-69 computes
        them inside each layer), so a training step computes them up front on a side stream: the 2048-step FPS chain of level 2 occupies 32
        workgroups for 1.5 ms and now runs UNDER level 1's feature passes instead of in front of level 2's.  One event per level; the main
        stream waits for a level's event right before that level's first consumer.  -> [(sampled points, down-sampling kNN, level kNN, event)]."""
        main = torch.cuda.current_stream(p0.device)
        side = getattr(self, "_geo_stream", None)
        if side is None or side.device != p0.device:
            side = self._geo_stream = ffi.stream_pool(p0.device, 1)[0]        # process-wide pool (hardware queues are few: ffi.stream_pool)
        side.wait_stream(main)
        out = []
        with torch.cuda.stream(side), torch.no_grad():
            p = p0
            for lvl in range(4):
                enc = getattr(self, f"enc{lvl + 1}")
                n_p, knn_down = (p, None) if enc[0].stride == 1 else enc[0].geometry(p, B)
                n = n_p.shape[0] // B
                knn_self = pointops.knn(self.nsamples[lvl], n_p, n_p, B, n, n)[0] if len(enc) > 1 else None
                ev = torch.cuda.Event()
                ev.record(side)
                for t in (n_p, knn_down, knn_self):
                    if t is not None:
                        t.record_stream(main)
                out.append((n_p, knn_down, knn_self, ev))
                p = n_p
        return out

    def forward_train(self, p: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
        """Same function with the autograd tape attached (BatchNorm per `self.training`)."""
        ffi.require_gpu(p, x)
        B, N = p.shape[0], p.shape[1]
        p0 = ffi.f32c(p).reshape(B * N, 3)
        x0 = p0 if self.c == 3 else torch.cat((p0, ffi.f32c(x).reshape(B * N, -1)), 1)
        main = torch.cuda.current_stream(p0.device)
        geo = self._geometry_pyramid(p0, B)
        for lvl in range(4):
            enc = getattr(self, f"enc{lvl + 1}")
            n_p, knn_down, knn_self, ev = geo[lvl]
            if knn_down is None:                          # stride 1: a per-point linear, no geometry - run it before waiting
                p0, x0 = enc[0].run_train(p0, x0, B)
                main.wait_event(ev)
            else:
                main.wait_event(ev)
                p0, x0 = enc[0].run_train(p0, x0, B, geo=(n_p, knn_down))
            for blk in list(enc)[1:]:
                x0 = blk.run_train(p0, x0, knn_self)
        return x0.view(B, -1, x0.shape[-1])

    def forward(self, p: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
        ffi.require_gpu(p, x)
        if torch.is_grad_enabled() and any(q.requires_grad for q in self.parameters()):
            return self.forward_train(p, x)
        with torch.no_grad():
            B, N = p.shape[0], p.shape[1]
            p0 = ffi.f32c(p).reshape(B * N, 3)
            x0 = p0 if self.c == 3 else torch.cat((p0, ffi.f32c(x).reshape(B * N, -1)), 1)
            if not self.overlap_geometry:
                for lvl in range(4):
                    enc = getattr(self, f"enc{lvl + 1}")
                    p0, x0 = enc[0].run(p0, x0, B)
                    n = p0.shape[0] // B
                    if len(enc) > 1:
                        knn_idx, _ = pointops.knn(self.nsamples[lvl], p0, p0, B, n, n)  # shared by every block of the level
                        for blk in list(enc)[1:]:
                            x0 = blk.run(p0, x0, knn_idx)
                return x0.view(B, -1, x0.shape[-1])
            # Round 6: the inference path takes the geometry pyramid of the training path - FPS and the neighbour lists of all four
            # levels depend on the coordinates only and run on a side stream (the 2047-round FPS chain of level 2 occupies 32 of the 256 CUs for
            # 1.5 ms), under level 1's per-point layers and vector attention instead of in front of level 2.  Same kernels on the same inputs: the
            # output is bit-identical to the sequential form (`overlap_geometry = False`).
            main = torch.cuda.current_stream(p0.device)
            geo = self._geometry_pyramid(p0, B)
            for lvl in range(4):
                enc = getattr(self, f"enc{lvl + 1}")
                n_p, knn_down, knn_self, ev = geo[lvl]
                if knn_down is None:                          # stride 1: a per-point linear, no geometry - run it before waiting
                    p0, x0 = enc[0].run(p0, x0, B)
                    main.wait_event(ev)
                else:
                    main.wait_event(ev)
                    p0, x0 = enc[0].run(p0, x0, B, geo=(n_p, knn_down))
                for blk in list(enc)[1:]:
                    x0 = blk.run(p0, x0, knn_self)
            return x0.view(B, -1, x0.shape[-1])


class TransitionUp(nn.Module):
    """Decoder stage (reference pointtransformer.py:72-99).  Head mode (out_planes None): every point is concatenated with
    ReLU(linear2(mean of its sample)), then Linear+BN+ReLU.  Fusion mode: Linear+BN+ReLU of the fine level plus the 3-NN
    inverse-distance interpolation of Linear+BN+ReLU of the coarse level."""

    def __init__(self, in_planes, out_planes=None):
        super().__init__()
        if out_planes is None:
            self.linear1 = nn.Sequential(nn.Linear(2 * in_planes, in_planes), nn.BatchNorm1d(in_planes), nn.ReLU(inplace=True))
            self.linear2 = nn.Sequential(nn.Linear(in_planes, in_planes), nn.ReLU(inplace=True))
        else:
            self.linear1 = nn.Sequential(nn.Linear(out_planes, out_planes), nn.BatchNorm1d(out_planes), nn.ReLU(inplace=True))
            self.linear2 = nn.Sequential(nn.Linear(in_planes, out_planes), nn.BatchNorm1d(out_planes), nn.ReLU(inplace=True))

    @staticmethod
    def _lin_bn_relu(x, lin: nn.Linear, bn: nn.BatchNorm1d):
        s, b = _bn_fold(bn, lin.bias)                                                       # BN(Wx + bias) = s*Wx + (s*bias + shift)
        return ops.linear(x, lin.weight, b, scale=s, act=ffi.ACT_RELU)

    def run_head(self, x: torch.Tensor, batch: int):
        n = x.shape[0] // batch
        g = ops.linear(pointops.segment_mean(x, batch, n), self.linear2[0].weight, self.linear2[0].bias, act=ffi.ACT_RELU)   # [B, c]
        cat = torch.cat((x, g.repeat_interleave(n, dim=0)), 1)
        return self._lin_bn_relu(cat, self.linear1[0], self.linear1[1])

    def run_fuse(self, p1, x1, p2, x2, batch: int):
        n1, n2 = p1.shape[0] // batch, p2.shape[0] // batch
        a = self._lin_bn_relu(x1, self.linear1[0], self.linear1[1])
        b = self._lin_bn_relu(x2, self.linear2[0], self.linear2[1])
        return pointops.interpolate(p2, p1, b, batch, n2, n1, base=a)

    # ---- differentiable forms (BatchNorm per `self.training`; pointtransformer.py:84-99)
    def run_head_train(self, x, batch: int):
        n = x.shape[0] // batch
        g = AG.linear(AG.segment_mean(x.view(batch, n, -1)), self.linear2[0].weight, self.linear2[0].bias, act=ffi.ACT_RELU)      # [B, c]
        cat = torch.cat((x, AP.broadcast_rows(g, n)), 1)
        return AP.batch_norm(AG.linear(cat, self.linear1[0].weight, self.linear1[0].bias), self.linear1[1], relu=True)

    def run_fuse_train(self, p1, x1, p2, x2, batch: int):
        n1, n2 = p1.shape[0] // batch, p2.shape[0] // batch
        a = AP.batch_norm(AG.linear(x1, self.linear1[0].weight, self.linear1[0].bias), self.linear1[1], relu=True)
        b = AP.batch_norm(AG.linear(x2, self.linear2[0].weight, self.linear2[0].bias), self.linear2[1], relu=True)
        return AP.interpolate(p2, p1, b, batch, n2, n1, base=a)


class SceneMapEncoderDecoder(nn.Module):
    """Multi-scale contact encoder of the CMDM `trans_dec` variant (reference models/modules.py:55-122): the SceneMapEncoder
    levels followed by an FPN-style decoder; returns the four level features [x4, x3, x2, x1] as [B, n_l, planes_l]
    (n_l = N/64, N/16, N/4, N), which become the memories of the cross-attention layers."""

    def __init__(self, point_feat_dim: int, planes: List, blocks: List, num_points: int = 8192) -> None:
        super().__init__()
        self.num_points = num_points
        self.c = point_feat_dim + 3
        self.in_planes, share = self.c, 8
        self.strides, self.nsamples = [1, 4, 4, 4], [8, 16, 16, 16]
        for i in range(4):
            layers = [TransitionDown(self.in_planes, planes[i], self.strides[i], self.nsamples[i])]
            self.in_planes = planes[i]
            layers += [PointTransformerBlock(planes[i], planes[i], share, nsample=self.nsamples[i]) for _ in range(1, blocks[i])]
            setattr(self, f"enc{i + 1}", nn.Sequential(*layers))
        for i in (3, 2, 1, 0):
            layers = [TransitionUp(self.in_planes, None if i == 3 else planes[i])]
            self.in_planes = planes[i]
            layers.append(PointTransformerBlock(planes[i], planes[i], share, nsample=self.nsamples[i]))
            setattr(self, f"dec{i + 1}", nn.Sequential(*layers))

    @property
    def num_groups(self):
        return self.num_points // 64

    def forward(self, p: torch.Tensor, x: torch.Tensor):
        ffi.require_gpu(p, x)
        if torch.is_grad_enabled() and any(q.requires_grad for q in self.parameters()):
            return self.forward_train(p, x)
        with torch.no_grad():
            B, N = p.shape[0], p.shape[1]
            p0 = ffi.f32c(p).reshape(B * N, 3)
            x0 = p0 if self.c == 3 else torch.cat((p0, ffi.f32c(x).reshape(B * N, -1)), 1)
            ps, xs, knns = [], [], []
            for lvl in range(4):
                enc = getattr(self, f"enc{lvl + 1}")
                p0, x0 = enc[0].run(p0, x0, B)
                n = p0.shape[0] // B
                ki, _ = pointops.knn(self.nsamples[lvl], p0, p0, B, n, n)
                for blk in list(enc)[1:]:
                    x0 = blk.run(p0, x0, ki)
                ps.append(p0); xs.append(x0); knns.append(ki)
            outs = [None] * 4
            y = self.dec4[1].run(ps[3], self.dec4[0].run_head(xs[3], B), knns[3])
            outs[3] = y
            for lvl in (2, 1, 0):
                dec = getattr(self, f"dec{lvl + 1}")
                y = dec[1].run(ps[lvl], dec[0].run_fuse(ps[lvl], xs[lvl], ps[lvl + 1], y, B), knns[lvl])
                outs[lvl] = y
            return [outs[l].view(B, -1, outs[l].shape[-1]) for l in (3, 2, 1, 0)]

    def forward_train(self, p: torch.Tensor, x: torch.Tensor):
        """Same function with the autograd tape attached (BatchNorm per `self.training`): differentiable HIP operators level by level."""
        B, N = p.shape[0], p.shape[1]
        p0 = ffi.f32c(p).reshape(B * N, 3)
        x0 = p0 if self.c == 3 else torch.cat((p0, ffi.f32c(x).reshape(B * N, -1)), 1)
        ps, xs, knns = [], [], []
        for lvl in range(4):
            enc = getattr(self, f"enc{lvl + 1}")
            p0, x0 = enc[0].run_train(p0, x0, B)
            n = p0.shape[0] // B
            with torch.no_grad():
                ki, _ = pointops.knn(self.nsamples[lvl], p0, p0, B, n, n)
            for blk in list(enc)[1:]:
                x0 = blk.run_train(p0, x0, ki)
            ps.append(p0); xs.append(x0); knns.append(ki)
        outs = [None] * 4
        y = self.dec4[1].run_train(ps[3], self.dec4[0].run_head_train(xs[3], B), knns[3])
        outs[3] = y
        for lvl in (2, 1, 0):
            dec = getattr(self, f"dec{lvl + 1}")
            y = dec[1].run_train(ps[lvl], dec[0].run_fuse_train(ps[lvl], xs[lvl], ps[lvl + 1], y, B), knns[lvl])
            outs[lvl] = y
        return [outs[l].view(B, -1, outs[l].shape[-1]) for l in (3, 2, 1, 0)]


class PointTransformerSeg(nn.Module):
    """Frozen scene backbone of the HUMANISE / novel ADM (reference pointtransformer.py:126-213,
    `pointtransformer_seg_repro`: blocks [2,3,4,6,3]): (xyz [B,N,3], colour [B,N,c-3]) -> per-point features [B,N,32].
    Step-invariant in sampling, so the CDM evaluates it once per scene batch."""

    def __init__(self, blocks=(2, 3, 4, 6, 3), c: int = 6, num_points: int = 8192):
        super().__init__()
        self.num_points, self.c = num_points, c
        self.in_planes = c
        planes, share = [32, 64, 128, 256, 512], 8
        self.strides, self.nsamples = [1, 4, 4, 4, 4], [8, 16, 16, 16, 16]
        for i in range(5):
            setattr(self, f"enc{i + 1}", self._make_enc(planes[i], blocks[i], share, self.strides[i], self.nsamples[i]))
        for i in (4, 3, 2, 1, 0):
            setattr(self, f"dec{i + 1}", self._make_dec(planes[i], share, self.nsamples[i], is_head=(i == 4)))

    def _make_enc(self, planes, blocks, share, stride, nsample):
        layers = [TransitionDown(self.in_planes, planes, stride, nsample)]
        self.in_planes = planes
        layers += [PointTransformerBlock(planes, planes, share, nsample=nsample) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def _make_dec(self, planes, share, nsample, is_head=False):
        layers = [TransitionUp(self.in_planes, None if is_head else planes)]
        self.in_planes = planes
        layers.append(PointTransformerBlock(planes, planes, share, nsample=nsample))
        return nn.Sequential(*layers)

    def forward(self, pxo):
        p, x = pxo
        ffi.require_gpu(p)
        with torch.no_grad():
            B, N = p.shape[0], p.shape[1]
            p0 = ffi.f32c(p).reshape(B * N, 3)
            x0 = p0 if self.c == 3 else torch.cat((p0, ffi.f32c(x).reshape(B * N, -1)), 1)
            ps, xs, knns = [], [], []
            for lvl in range(5):
                enc = getattr(self, f"enc{lvl + 1}")
                p0, x0 = enc[0].run(p0, x0, B)
                n = p0.shape[0] // B
                ki, _ = pointops.knn(self.nsamples[lvl], p0, p0, B, n, n)
                for blk in list(enc)[1:]:
                    x0 = blk.run(p0, x0, ki)
                ps.append(p0); xs.append(x0); knns.append(ki)
            y = self.dec5[0].run_head(xs[4], B)
            y = self.dec5[1].run(ps[4], y, knns[4])
            for lvl in (3, 2, 1, 0):
                dec = getattr(self, f"dec{lvl + 1}")
                y = dec[0].run_fuse(ps[lvl], xs[lvl], ps[lvl + 1], y, B)
                y = dec[1].run(ps[lvl], y, knns[lvl])
            return y.view(B, N, -1)
