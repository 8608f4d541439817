"""CPU check of the exactness argument behind `fps_pruned_kernel` (csrc/pointops.hip): an emulation of its per-round logic in numpy float32
(Morton-sorted cells, bounding-box lower bound evaluated with the same operations and association as the point distances, waves of 64
cells that skip when no cell's bound is below the cell's current maximum, ties by lowest original index) against the FPS oracle.
Asserted on every round: a skipped cell's running minima would not have changed (the bound is never violated), and the sampled
indices equal the oracle's - on random clouds, on clouds whose points all exist twice (ties every round) and for n that is not a
power of two.  The kernel itself is tested bit-exact on the GPU (tests/test_gpu_points.py); this pins the argument without one."""
import numpy as np
import pytest
import torch

from afm import synth
from oracle import pointops_ref as po


def _morton_perm(p):
    lo, hi = p.min(0), p.max(0)
    q = np.clip(((p - lo) / np.maximum(hi - lo, 1e-12) * 1023).astype(np.int64), 0, 1023)

    def part(x):
        x = x & 0x3ff; x = (x | (x << 16)) & 0x30000ff; x = (x | (x << 8)) & 0x300f00f; x = (x | (x << 4)) & 0x30c30c3; x = (x | (x << 2)) & 0x9249249
        return x
    return np.argsort(part(q[:, 0]) | (part(q[:, 1]) << 1) | (part(q[:, 2]) << 2), kind="stable")


def _fps_pruned(p, m, threads=1024):
    p = p.astype(np.float32)
    n = len(p)
    cap = 2048
    while cap < n:
        cap <<= 1
    ppt = cap // threads
    empty = 0xFFFFFFFF
    oi = np.full(cap, empty, dtype=np.int64); oi[:n] = _morton_perm(p)
    pts = np.zeros((cap, 3), np.float32); pts[:n] = p[oi[:n]]
    cells, cp = oi.reshape(threads, ppt), pts.reshape(threads, ppt, 3)
    cv = cells != empty
    tmp = np.where(cv, np.float32(1e10), np.float32(0)).astype(np.float32)
    lo = np.where(cv[..., None], cp, np.float32(np.inf)).min(1)
    hi = np.where(cv[..., None], cp, np.float32(-np.inf)).max(1)
    cell_ok = cv.any(1)

    def cell_argmax():
        tb = tmp.view(np.uint32).astype(np.int64)
        bd = tb.max(1)
        return bd, np.where((tb == bd[:, None]) & cv, cells, empty).min(1)

    bd, boi = cell_argmax()
    out, cur, scanned = [0], 0, 0
    for _ in range(1, m):
        c = p[cur]
        with np.errstate(invalid="ignore"):
            e = np.maximum(np.maximum(lo - c, c - hi), np.float32(0)).astype(np.float32)
        lb = ((e[:, 0] * e[:, 0] + e[:, 1] * e[:, 1]) + e[:, 2] * e[:, 2]).astype(np.float32)
        need = cell_ok & (lb.view(np.uint32).astype(np.int64) < bd)
        wave_scans = np.repeat(need.reshape(-1, 64).any(1), 64)
        scanned += int(wave_scans.sum()) // 64
        d = cp - c
        d2 = ((d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]).astype(np.float32)
        new = np.minimum(tmp, d2)
        assert (((new == tmp) | ~cv)[~need]).all(), "a cell whose bound said 'cannot change' would have changed"
        tmp = np.where(wave_scans[:, None] & cv, new, tmp)
        nbd, nboi = cell_argmax()
        bd, boi = np.where(wave_scans, nbd, bd), np.where(wave_scans, nboi, boi)
        key = np.where(cell_ok, bd, -1)
        cur = int(np.where(key == key.max(), boi, empty).min())
        out.append(cur)
    return np.array(out), scanned / ((m - 1) * (threads // 64))


@pytest.mark.parametrize("n,m,dup_shift", [(8192, 300, 0), (8192, 200, 70), (3000, 250, 0), (2048, 200, 33), (1536, 100, 0)])
def test_pruned_scan_is_exact(n, m, dup_shift):
    if dup_shift:
        base = synth.scene_cloud(1, n // 2, seed=27).reshape(n // 2, 3)
        p = torch.cat([base, base.roll(dup_shift, 0)], 0).contiguous()
    else:
        p = synth.scene_cloud(1, n, seed=21).reshape(n, 3)
    want = po.furthest_sampling(p, torch.tensor([n], dtype=torch.int32), torch.tensor([m], dtype=torch.int32)).numpy()
    got, frac = _fps_pruned(p.numpy(), m)
    assert np.array_equal(got, want)
    assert frac < 0.7                                  # the pruning prunes (16 % of the waves scan at n = 8192, m = 2048)
