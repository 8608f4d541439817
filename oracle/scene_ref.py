"""TEST INFRASTRUCTURE ONLY - CPU restatement of the point-cloud branch of the
hot path: TransitionDown ("set abstraction"), PointTransformerLayer/Block and
SceneMapEncoder, eval mode (BatchNorm on running statistics).

Functional style over a flat state dict ``sd`` + key prefix; citations are into
/root/reference/models/.  FPS / kNN come from oracle/pointops_ref.py (parity
unpinned, see there).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

from . import pointops_ref as po

SD = Dict[str, torch.Tensor]


def _lin(sd: SD, pre: str, x):
    return F.linear(x, sd[pre + ".weight"], sd.get(pre + ".bias"))


_BATCH_STATS = False      # train-mode BatchNorm (batch statistics, biased variance) instead of the running statistics


class batch_stats:
    """``with scene_ref.batch_stats():`` evaluates every BatchNorm below on batch statistics (model.train() semantics of
    nn.BatchNorm1d; the running-statistics update is restated in `bn_running_update`)."""

    def __init__(self, on: bool = True):
        self.on = on

    def __enter__(self):
        global _BATCH_STATS
        self.prev, _BATCH_STATS = _BATCH_STATS, self.on

    def __exit__(self, *a):
        global _BATCH_STATS
        _BATCH_STATS = self.prev


def bn_running_update(running_mean, running_var, x2d, momentum: float = 0.1):
    """nn.BatchNorm1d's buffer update for one train-mode forward over rows x2d [rows, C] (unbiased variance)."""
    m, v = x2d.mean(0), x2d.var(0, unbiased=True)
    return (1 - momentum) * running_mean + momentum * m, (1 - momentum) * running_var + momentum * v


def _bn(sd: SD, pre: str, x, channel_dim: int = -1, eps: float = 1e-5):
    """BatchNorm1d over ``channel_dim``: running statistics (eval) or, inside `batch_stats()`, batch statistics."""
    if _BATCH_STATS:
        cd = channel_dim if channel_dim >= 0 else x.dim() + channel_dim
        dims = [d for d in range(x.dim()) if d != cd]
        shape = [1] * x.dim()
        shape[cd] = -1
        mean = x.mean(dims, keepdim=True)
        var = x.var(dims, unbiased=False, keepdim=True)
        return (x - mean) / torch.sqrt(var + eps) * sd[pre + ".weight"].view(shape) + sd[pre + ".bias"].view(shape)
    if channel_dim != 1 and x.dim() > 2:
        x = x.transpose(1, channel_dim if channel_dim >= 0 else x.dim() + channel_dim)
        y = F.batch_norm(x.contiguous(), sd[pre + ".running_mean"], sd[pre + ".running_var"],
                         sd[pre + ".weight"], sd[pre + ".bias"], False, 0.0, eps)
        return y.transpose(1, channel_dim if channel_dim >= 0 else y.dim() + channel_dim).contiguous()
    return F.batch_norm(x, sd[pre + ".running_mean"], sd[pre + ".running_var"],
                        sd[pre + ".weight"], sd[pre + ".bias"], False, 0.0, eps)


def query_and_group(nsample, xyz, new_xyz, feat, offset, new_offset, use_xyz=True, idx=None):
    """scene_models/pointops.py:79-100: kNN, gather, relative xyz in front."""
    if idx is None:
        idx, _ = po.knn_query(nsample, xyz, new_xyz, offset, new_offset)
    m, c = new_xyz.shape[0], feat.shape[1]
    flat = idx.view(-1).long()
    g_xyz = xyz[flat].view(m, nsample, 3) - new_xyz.unsqueeze(1)
    g_feat = feat[flat].view(m, nsample, c)
    return torch.cat((g_xyz, g_feat), -1) if use_xyz else g_feat


def transition_down(sd: SD, pre: str, p, x, o, stride: int, nsample: int):
    """scene_models/pointtransformer.py:41-69.  Returns (p', x', o', aux)."""
    aux = {}
    if stride != 1:
        counts = torch.diff(o, prepend=o.new_zeros(1))
        n_o = torch.cumsum(counts // stride, 0).to(torch.int32)
        idx = po.furthest_sampling(p, o, n_o)                                  # (m)
        n_p = p[idx.long()]
        knn_idx, _ = po.knn_query(nsample, p, n_p, o, n_o)
        g = query_and_group(nsample, p, n_p, x, o, n_o, use_xyz=True, idx=knn_idx)   # (m, k, 3+c)
        y = _lin(sd, pre + ".linear", g)                                       # (m, k, c')
        y = F.relu(_bn(sd, pre + ".bn", y.transpose(1, 2).contiguous(), channel_dim=1))  # (m, c', k)
        y = y.max(dim=2).values                                                # MaxPool1d(k)
        aux.update(fps_idx=idx, knn_idx=knn_idx)
        return n_p, y, n_o, aux
    y = F.relu(_bn(sd, pre + ".bn", _lin(sd, pre + ".linear", x)))
    return p, y, o, aux


def point_transformer_layer(sd: SD, pre: str, p, x, o, nsample: int, share_planes: int = 8, knn_idx=None):
    """scene_models/pointtransformer.py:26-38 (vector attention over the k nearest neighbours)."""
    x_q, x_k, x_v = _lin(sd, pre + ".linear_q", x), _lin(sd, pre + ".linear_k", x), _lin(sd, pre + ".linear_v", x)
    if knn_idx is None:
        knn_idx, _ = po.knn_query(nsample, p, p, o, o)
    g_k = query_and_group(nsample, p, p, x_k, o, o, use_xyz=True, idx=knn_idx)      # (n, k, 3+c)
    g_v = query_and_group(nsample, p, p, x_v, o, o, use_xyz=False, idx=knn_idx)     # (n, k, c)
    p_r, g_k = g_k[:, :, 0:3], g_k[:, :, 3:]
    # linear_p = Linear(3,3) -> BN(3) [over (n,3,k)] -> ReLU -> Linear(3,c)
    p_r = _lin(sd, pre + ".linear_p.0", p_r)
    p_r = F.relu(_bn(sd, pre + ".linear_p.1", p_r.transpose(1, 2).contiguous(), channel_dim=1).transpose(1, 2).contiguous())
    p_r = _lin(sd, pre + ".linear_p.3", p_r)                                        # (n, k, c)
    w = g_k - x_q.unsqueeze(1) + p_r                                                # out_planes // mid_planes == 1
    # linear_w = BN(c) -> ReLU -> Linear(c, c/s) -> BN(c/s) -> ReLU -> Linear(c/s, c/s)
    w = F.relu(_bn(sd, pre + ".linear_w.0", w.transpose(1, 2).contiguous(), channel_dim=1).transpose(1, 2).contiguous())
    w = _lin(sd, pre + ".linear_w.2", w)
    w = F.relu(_bn(sd, pre + ".linear_w.3", w.transpose(1, 2).contiguous(), channel_dim=1).transpose(1, 2).contiguous())
    w = _lin(sd, pre + ".linear_w.5", w)
    w = torch.softmax(w, dim=1)                                                     # over the k neighbours
    n, k, c = g_v.shape
    s = share_planes
    return ((g_v + p_r).view(n, k, s, c // s) * w.unsqueeze(2)).sum(1).view(n, c)


def point_transformer_block(sd: SD, pre: str, p, x, o, nsample: int, share_planes: int = 8, knn_idx=None):
    """scene_models/pointtransformer.py:102-123."""
    identity = x
    y = F.relu(_bn(sd, pre + ".bn1", _lin(sd, pre + ".linear1", x)))
    y = F.relu(_bn(sd, pre + ".bn2", point_transformer_layer(sd, pre + ".transformer2", p, y, o, nsample, share_planes, knn_idx)))
    y = _bn(sd, pre + ".bn3", _lin(sd, pre + ".linear3", y))
    return F.relu(y + identity)


def scene_map_encoder(sd: SD, pre: str, p: torch.Tensor, x: torch.Tensor, blocks=(2, 2, 2, 2),
                      stride=(1, 4, 4, 4), nsample=(8, 16, 16, 16), return_aux: bool = False):
    """modules.py:124-167.  p [B,N,3], x [B,N,J] -> [B, N/64, planes[-1]]."""
    B, N = p.shape[:2]
    o = torch.arange(1, B + 1, dtype=torch.int32) * N
    p0 = p.reshape(B * N, 3).contiguous()
    x0 = torch.cat((p0, x.reshape(B * N, -1)), 1)
    aux_all: List[dict] = []
    for lvl in range(4):
        e = f"{pre}.enc{lvl + 1}" if pre else f"enc{lvl + 1}"
        p0, x0, o, aux = transition_down(sd, e + ".0", p0, x0, o, stride[lvl], nsample[lvl])
        knn_idx, _ = po.knn_query(nsample[lvl], p0, p0, o, o)       # same (p, o, k) for every layer of the level
        aux["self_knn_idx"] = knn_idx
        for j in range(1, blocks[lvl]):
            x0 = point_transformer_block(sd, f"{e}.{j}", p0, x0, o, nsample[lvl], 8, knn_idx)
        aux["p"], aux["x"] = p0, x0
        aux_all.append(aux)
    out = x0.view(B, -1, x0.shape[-1])
    return (out, aux_all) if return_aux else out


def interpolation(xyz, new_xyz, feat, offset, new_offset, k: int = 3):
    """scene_models/pointops.py:164-178: k-NN inverse-distance weighted feature upsampling (dist = sqrt(d2))."""
    idx, d2 = po.knn_query(k, xyz, new_xyz, offset, new_offset)
    dist_recip = 1.0 / (torch.sqrt(d2) + 1e-8)
    weight = dist_recip / dist_recip.sum(dim=1, keepdim=True)
    out = torch.zeros(new_xyz.shape[0], feat.shape[1])
    for i in range(k):
        out += feat[idx[:, i].long(), :] * weight[:, i].unsqueeze(-1)
    return out


def transition_up(sd: SD, pre: str, p1, x1, o1, p2=None, x2=None, o2=None):
    """scene_models/pointtransformer.py:72-99.  Head mode (pxo2 is None): concat every point with the linear2-transformed
    mean of its sample, then linear1+BN+ReLU.  Fusion mode: linear1(x1) + interpolate(linear2(x2)) from the coarser level."""
    if p2 is None:
        parts, s_i = [], 0
        for e_i in [int(v) for v in o1]:
            xb = x1[s_i:e_i]
            g = F.relu(_lin(sd, pre + ".linear2.0", xb.sum(0, True) / (e_i - s_i)))
            parts.append(torch.cat((xb, g.repeat(e_i - s_i, 1)), 1))
            s_i = e_i
        x = torch.cat(parts, 0)
        return F.relu(_bn(sd, pre + ".linear1.1", _lin(sd, pre + ".linear1.0", x)))
    a = F.relu(_bn(sd, pre + ".linear1.1", _lin(sd, pre + ".linear1.0", x1)))
    b = F.relu(_bn(sd, pre + ".linear2.1", _lin(sd, pre + ".linear2.0", x2)))
    return a + interpolation(p2, p1, b, o2, o1)


def point_transformer_seg(sd: SD, pre: str, p: torch.Tensor, x: torch.Tensor, blocks=(2, 3, 4, 6, 3),
                          stride=(1, 4, 4, 4, 4), nsample=(8, 16, 16, 16, 16)):
    """scene_models/pointtransformer.py:126-213 (`pointtransformer_seg_repro`): 5-level encoder + FPN-style decoder.
    p [B,N,3], x [B,N,c-3] -> per-point features [B,N,32]."""
    B, N = p.shape[:2]
    P = lambda n: f"{pre}.{n}" if pre else n
    o = torch.arange(1, B + 1, dtype=torch.int32) * N
    p0 = p.reshape(B * N, 3).contiguous()
    x0 = torch.cat((p0, x.reshape(B * N, -1)), 1) if x is not None and x.shape[-1] > 0 else p0
    ps, xs, os_, knn = [], [], [], []
    for lvl in range(5):
        e = P(f"enc{lvl + 1}")
        p0, x0, o, _ = transition_down(sd, e + ".0", p0, x0, o, stride[lvl], nsample[lvl])
        ki, _ = po.knn_query(nsample[lvl], p0, p0, o, o)
        for j in range(1, blocks[lvl]):
            x0 = point_transformer_block(sd, f"{e}.{j}", p0, x0, o, nsample[lvl], 8, ki)
        ps.append(p0); xs.append(x0); os_.append(o); knn.append(ki)
    # decoders: every dec level = TransitionUp + (2 - 1) PointTransformerBlock
    y = transition_up(sd, P("dec5.0"), ps[4], xs[4], os_[4])
    y = point_transformer_block(sd, P("dec5.1"), ps[4], y, os_[4], nsample[4], 8, knn[4])
    for lvl in (3, 2, 1, 0):
        d = P(f"dec{lvl + 1}")
        y = transition_up(sd, d + ".0", ps[lvl], xs[lvl], os_[lvl], ps[lvl + 1], y, os_[lvl + 1])
        y = point_transformer_block(sd, d + ".1", ps[lvl], y, os_[lvl], nsample[lvl], 8, knn[lvl])
    return y.view(B, N, -1)


def scene_map_encoder_decoder(sd: SD, pre: str, p: torch.Tensor, x: torch.Tensor, blocks=(2, 2, 2, 2), stride=(1, 4, 4, 4),
                              nsample=(8, 16, 16, 16)):
    """modules.py:55-122: SceneMapEncoder levels + FPN decoder; returns [x4, x3, x2, x1] as [B, n_l, c_l]."""
    B, N = p.shape[:2]
    P = lambda n: f"{pre}.{n}" if pre else n
    o = torch.arange(1, B + 1, dtype=torch.int32) * N
    p0 = p.reshape(B * N, 3).contiguous()
    x0 = torch.cat((p0, x.reshape(B * N, -1)), 1)
    ps, xs, os_, knn = [], [], [], []
    for lvl in range(4):
        e = P(f"enc{lvl + 1}")
        p0, x0, o, _ = transition_down(sd, e + ".0", p0, x0, o, stride[lvl], nsample[lvl])
        ki, _ = po.knn_query(nsample[lvl], p0, p0, o, o)
        for j in range(1, blocks[lvl]):
            x0 = point_transformer_block(sd, f"{e}.{j}", p0, x0, o, nsample[lvl], 8, ki)
        ps.append(p0); xs.append(x0); os_.append(o); knn.append(ki)
    outs = [None] * 4
    y = transition_up(sd, P("dec4.0"), ps[3], xs[3], os_[3])
    y = point_transformer_block(sd, P("dec4.1"), ps[3], y, os_[3], nsample[3], 8, knn[3])
    outs[3] = y
    for lvl in (2, 1, 0):
        d = P(f"dec{lvl + 1}")
        y = transition_up(sd, d + ".0", ps[lvl], xs[lvl], os_[lvl], ps[lvl + 1], y, os_[lvl + 1])
        y = point_transformer_block(sd, d + ".1", ps[lvl], y, os_[lvl], nsample[lvl], 8, knn[lvl])
        outs[lvl] = y
    return [outs[l].view(B, -1, outs[l].shape[-1]) for l in (3, 2, 1, 0)]
