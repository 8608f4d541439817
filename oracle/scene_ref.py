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


def _bn(sd: SD, pre: str, x, channel_dim: int = -1, eps: float = 1e-5):
    """Eval-mode BatchNorm1d over ``channel_dim``."""
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
