"""Tensor-level wrappers of the point-cloud C-ABI entry points (FPS, kNN, fused set abstraction,
fused vector attention).  Replaces `models/scene_models/pointops.py` + `pointops_cuda` of the reference."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import ffi


def furthest_point_sampling(xyz: torch.Tensor, batch: int, n: int, m: int) -> torch.Tensor:
    """xyz [batch*n, 3] -> int32 [batch*m] global row indices (first point of each sample first)."""
    lib = ffi.load()
    ffi.require_gpu(xyz)
    xyz = ffi.f32c(xyz)
    assert xyz.shape == (batch * n, 3)
    idx = torch.empty(batch * m, dtype=torch.int32, device=xyz.device)
    ffi.check(lib.afm_fps(xyz.data_ptr(), batch, n, m, idx.data_ptr(), ffi.stream_of(xyz)), "afm_fps")
    return idx


def knn(k: int, xyz: torch.Tensor, new_xyz: torch.Tensor, batch: int, n: int, m: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """k nearest source points of every query within its own sample -> (idx int32 [batch*m, k] global rows,
    dist2 f32 [batch*m, k] squared distances), ascending."""
    lib = ffi.load()
    ffi.require_gpu(xyz, new_xyz)
    xyz, new_xyz = ffi.f32c(xyz), ffi.f32c(new_xyz)
    assert xyz.shape == (batch * n, 3) and new_xyz.shape == (batch * m, 3)
    idx = torch.empty(batch * m, k, dtype=torch.int32, device=xyz.device)
    d2 = torch.empty(batch * m, k, dtype=torch.float32, device=xyz.device)
    nbytes = lib.afm_knn_workspace_bytes(k, batch, n, m) if PRUNED_KNN else 0
    if nbytes < 0:
        ffi.check(int(nbytes), "afm_knn_workspace_bytes")
    if nbytes > 0:                                       # exact spatial pruning (round 6): Morton-sorted candidate tiles with bounding boxes; same indices, same order
        ws = _knn_workspace(int(nbytes), xyz.device)
        ffi.check(lib.afm_knn_ws(k, xyz.data_ptr(), new_xyz.data_ptr(), batch, n, m, idx.data_ptr(), d2.data_ptr(), ws.data_ptr(), ws.numel(),
                                 ffi.stream_of(xyz)), "afm_knn_ws")
        return idx, d2
    ffi.check(lib.afm_knn(k, xyz.data_ptr(), new_xyz.data_ptr(), batch, n, m, idx.data_ptr(), d2.data_ptr(),
                          ffi.stream_of(xyz)), "afm_knn")
    return idx, d2


PRUNED_KNN = True            # host-side measurement switch: False = the plain all-pairs kernel everywhere (bit-identical results)
_KNN_WS: dict = {}


def _knn_workspace(nbytes: int, device) -> torch.Tensor:
    """Scratch of afm_knn_ws (sorted copies + tile boxes), one growing buffer per device AND stream (a buffer shared by two streams would be
    overwritten by the second call while the first still reads it: the training path runs its neighbour searches on a side stream)."""
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)
    buf = _KNN_WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = _KNN_WS[key] = torch.empty(nbytes, dtype=torch.uint8, device=device)
    return buf


def gather_rows(src: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    lib = ffi.load()
    ffi.require_gpu(src, idx)
    src = ffi.f32c(src)
    idx = idx.to(torch.int32).contiguous()
    out = torch.empty(idx.numel(), src.shape[1], dtype=torch.float32, device=src.device)
    ffi.check(lib.afm_gather_rows(src.data_ptr(), idx.data_ptr(), out.data_ptr(), idx.numel(), src.shape[1],
                                  ffi.stream_of(src)), "afm_gather_rows")
    return out


def transition_down(p, x, new_p, knn_idx, weight, scale, shift) -> torch.Tensor:
    """max_j ReLU(scale * (W [p_j - p'_i ; x_j]) + shift) -> [M, cout]."""
    lib = ffi.load()
    ffi.require_gpu(p, x, new_p, knn_idx)
    p, x, new_p = ffi.f32c(p), ffi.f32c(x), ffi.f32c(new_p)
    weight, scale, shift = ffi.f32c(weight.detach()), ffi.f32c(scale.detach()), ffi.f32c(shift.detach())
    knn_idx = knn_idx.contiguous()
    M, k = knn_idx.shape
    cout, c = weight.shape[0], x.shape[1]
    assert weight.shape[1] == 3 + c
    out = torch.empty(M, cout, dtype=torch.float32, device=p.device)
    ffi.check(lib.afm_transition_down(p.data_ptr(), x.data_ptr(), c, new_p.data_ptr(), knn_idx.data_ptr(), k,
                                      weight.data_ptr(), cout, scale.data_ptr(), shift.data_ptr(), out.data_ptr(), M,
                                      ffi.stream_of(p)), "afm_transition_down")
    return out


def pt_attention(p, qkv, knn_idx, channels: int, share_planes: int, lp0_w, lp0_b, lp_s, lp_t, lp3_w, lp3_b,
                 w0_s, w0_t, w2_w, w2_b, w3_s, w3_t, w5_w, w5_b, out_scale=None, out_shift=None, relu=False) -> torch.Tensor:
    lib = ffi.load()
    ffi.require_gpu(p, qkv, knn_idx)
    a = ffi.PtAttentionArgs()
    keep = []

    def P(t: Optional[torch.Tensor]):
        if t is None:
            return None
        t = ffi.f32c(t.detach())
        keep.append(t)
        return t.data_ptr()

    p, qkv, knn_idx = ffi.f32c(p), ffi.f32c(qkv), knn_idx.contiguous()
    n = p.shape[0]
    out = torch.empty(n, channels, dtype=torch.float32, device=p.device)
    a.p, a.qkv, a.knn_idx, a.out = p.data_ptr(), qkv.data_ptr(), knn_idx.data_ptr(), out.data_ptr()
    a.n, a.channels, a.nsample, a.share_planes = n, channels, knn_idx.shape[1], share_planes
    for name, t in (("lp0_w", lp0_w), ("lp0_b", lp0_b), ("lp_bn_scale", lp_s), ("lp_bn_shift", lp_t), ("lp3_w", lp3_w),
                    ("lp3_b", lp3_b), ("w0_bn_scale", w0_s), ("w0_bn_shift", w0_t), ("w2_w", w2_w), ("w2_b", w2_b),
                    ("w3_bn_scale", w3_s), ("w3_bn_shift", w3_t), ("w5_w", w5_w), ("w5_b", w5_b),
                    ("out_scale", out_scale), ("out_shift", out_shift)):
        setattr(a, name, P(t))
    a.relu = 1 if relu else 0
    ffi.check(lib.afm_pt_attention(C.byref(a), ffi.stream_of(p)), "afm_pt_attention")
    return out


def interpolate(xyz_src, xyz_dst, feat, batch: int, m: int, n: int, base=None, k: int = 3) -> torch.Tensor:
    """k-NN inverse-distance upsampling of ``feat`` [batch*m, c] from xyz_src to the batch*n points xyz_dst
    (+ ``base`` [batch*n, c] if given) - pointops.interpolation of the reference."""
    lib = ffi.load()
    idx, d2 = knn(k, xyz_src, xyz_dst, batch, m, n)
    feat = ffi.f32c(feat)
    out = torch.empty(batch * n, feat.shape[1], dtype=torch.float32, device=feat.device)
    b = None if base is None else ffi.f32c(base)
    ffi.check(lib.afm_interpolate(feat.data_ptr(), idx.data_ptr(), d2.data_ptr(), ffi.ptr(b), out.data_ptr(), batch * n,
                                  feat.shape[1], k, ffi.stream_of(feat)), "afm_interpolate")
    return out


def segment_mean(x: torch.Tensor, batch: int, n: int) -> torch.Tensor:
    lib = ffi.load()
    ffi.require_gpu(x)
    x = ffi.f32c(x)
    out = torch.empty(batch, x.shape[1], dtype=torch.float32, device=x.device)
    ffi.check(lib.afm_segment_mean(x.data_ptr(), out.data_ptr(), batch, n, x.shape[1], ffi.stream_of(x)), "afm_segment_mean")
    return out
