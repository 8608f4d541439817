"""TEST INFRASTRUCTURE ONLY - CPU restatement of the two `pointops_cuda` kernels
the reference reaches (never imported by the product path).

PARITY UNPINNED: the CUDA source lives in the un-vendored, un-pinned dependency
`git+https://github.com/Silverster98/pointops` (reference requirements.txt:1);
it is absent from /root/reference and no reference test pins its tie-breaking.
The semantics below restate the published algorithm as seen from the reference
call sites (models/scene_models/pointops.py:10-45):

* furthest point sampling (pointops.py:10-27): per batch segment, start at the
  segment's first point, keep ``tmp[k] = min(tmp[k], |p_k - p_last|^2)`` with
  ``tmp`` initialised to 1e10 (pointops.py:22), take the arg-max each round.
* kNN query (pointops.py:30-45): brute force over the query's batch segment,
  neighbours returned in ascending distance; the wrapper returns sqrt(dist2).

Rules we add (and the HIP kernels implement bit-exactly):
  d2 = (dx*dx + dy*dy) + dz*dz evaluated in float32 WITHOUT fma contraction;
  FPS ties -> lowest index; kNN order -> lexicographic (d2, index).

Why the tie rule cannot change a result.  Upstream's FPS kernel is the usual PointNet++ one: every thread scans a
strided subset of the points and a shared-memory tree keeps `v2 > v1 ? i2 : i1`, so among EXACTLY equal running
distances it prefers the lowest THREAD (point 1024, owned by thread 0, would beat point 1, owned by thread 1, once
n > the block size) while this restatement prefers the lowest INDEX.  An exact tie of `tmp` between two points needs
them to be at identical distances from every point sampled so far; with float32 coordinates of a real (or the synthetic,
tie-free) scan that happens only for DUPLICATED points - identical xyz (prepare/generate_contact_data.py:418-423 can emit
them).  Duplicates carry identical coordinates and, being the same scene point, identical features, so whichever of
them is picked, the sampled coordinate, the gathered features and everything computed from them are identical; only the
integer index stored in `idx` may differ, and nothing downstream of TransitionDown reads it as anything but a gather
address (pointtransformer.py:61-68).  Same argument for kNN: equal-distance neighbours are duplicates of each other.
tests/test_gpu_points.py::test_fps_ties_pick_lowest_index pins OUR rule on duplicated points.
"""
from __future__ import annotations

import numpy as np
import torch


def _d2(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a [m,1,3] or [1,3]; b [1,n,3] or [n,3] -> squared distance, f32, no fma."""
    d = a - b
    d = d * d
    return (d[..., 0] + d[..., 1]) + d[..., 2]


def furthest_sampling(xyz: torch.Tensor, offset: torch.Tensor, new_offset: torch.Tensor) -> torch.Tensor:
    """xyz (n,3) f32, offset (b) int, new_offset (b) int -> idx (m) int32 of global rows."""
    xyz = xyz.detach().cpu().float().contiguous()
    off = [int(v) for v in offset]
    noff = [int(v) for v in new_offset]
    out = np.zeros(noff[-1], dtype=np.int32)
    s_n = s_m = 0
    pts = xyz.numpy()
    for e_n, e_m in zip(off, noff):
        p = pts[s_n:e_n]
        tmp = np.full(p.shape[0], 1e10, dtype=np.float32)
        cur = 0
        if e_m > s_m:
            out[s_m] = s_n
        for j in range(s_m + 1, e_m):
            d = p - p[cur]
            d = d * d
            d = (d[:, 0] + d[:, 1]) + d[:, 2]
            np.minimum(tmp, d, out=tmp)
            cur = int(np.argmax(tmp))          # first maximum == lowest index
            out[j] = s_n + cur
        s_n, s_m = e_n, e_m
    return torch.from_numpy(out)


def knn_query(nsample: int, xyz: torch.Tensor, new_xyz: torch.Tensor, offset: torch.Tensor,
              new_offset: torch.Tensor, chunk: int = 1024):
    """-> idx (m, nsample) int32 global rows, dist2 (m, nsample) f32, ascending (d2, idx)."""
    xyz = xyz.detach().cpu().float().contiguous()
    new_xyz = xyz if new_xyz is None else new_xyz.detach().cpu().float().contiguous()
    off = [int(v) for v in offset]
    noff = [int(v) for v in new_offset]
    m = new_xyz.shape[0]
    idx = torch.zeros(m, nsample, dtype=torch.int32)
    dist2 = torch.zeros(m, nsample, dtype=torch.float32)
    s_n = s_m = 0
    for e_n, e_m in zip(off, noff):
        p = xyz[s_n:e_n]
        n = p.shape[0]
        k = min(nsample, n)
        ar = torch.arange(n, dtype=torch.int64)[None, :]
        for c0 in range(s_m, e_m, chunk):
            c1 = min(c0 + chunk, e_m)
            d = _d2(new_xyz[c0:c1, None, :], p[None, :, :])            # [c, n]
            key = (d.view(torch.int32).to(torch.int64) << 32) | ar      # d >= 0 so bits are monotone
            kk = torch.topk(key, k, dim=1, largest=False, sorted=True).values
            ii = (kk & 0xFFFFFFFF).to(torch.int64)
            idx[c0:c1, :k] = (ii + s_n).to(torch.int32)
            dist2[c0:c1, :k] = torch.gather(d, 1, ii)
            if k < nsample:                                            # fewer points than k: repeat the last
                idx[c0:c1, k:] = idx[c0:c1, k - 1:k]
                dist2[c0:c1, k:] = dist2[c0:c1, k - 1:k]
        s_n, s_m = e_n, e_m
    return idx, dist2


# ---- in-place `pointops_cuda`-shaped entry points (used by oracle/stubs to run the reference)

def furthestsampling_cuda(b, n_max, xyz, offset, new_offset, tmp, idx):
    idx.copy_(furthest_sampling(xyz, offset, new_offset))


def knnquery_cuda(m, nsample, xyz, new_xyz, offset, new_offset, idx, dist2):
    i, d = knn_query(nsample, xyz, new_xyz, offset, new_offset)
    idx.copy_(i)
    dist2.copy_(d)
