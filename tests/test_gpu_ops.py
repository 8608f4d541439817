"""`-m gpu`: primitive HIP kernels (through the C-ABI) vs the CPU oracle / plain torch-CPU math.
Tolerances: f32 MFMA GEMM 2e-4 abs on O(1..30) outputs (K <= 1024, exact-f32 products, different
summation order); attention / LayerNorm 2e-5; DDPM update and Philox are bit-exact / statistical."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from afm import ffi, ops, synth
from conftest import golden
from gpu_util import dev, report

pytestmark = pytest.mark.gpu


def test_library_loaded_and_versioned():
    lib = ffi.load()
    assert lib.afm_version() == ffi.ABI_VERSION == 7


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (326, 512, 512), (1000, 1536, 512), (777, 263, 512),
                                   (392, 512, 263), (64, 64, 9), (10432, 512, 1024), (33, 96, 35), (5, 7, 3)])
def test_linear_shapes(M, N, K):
    x = synth.gaussian("lin_x", (M, K)); w = synth.gaussian("lin_w", (N, K)) / math.sqrt(K); b = synth.gaussian("lin_b", (N,))
    want = F.linear(x.double(), w.double(), b.double()).float()
    got = ops.linear(x.to(dev()), w.to(dev()), b.to(dev()))
    report(f"linear {M}x{N}x{K}", got, want, 2e-4)


@pytest.mark.parametrize("M,N,K", [(70001, 3, 3), (50000, 4, 4), (33333, 8, 8), (40000, 32, 3), (40001, 32, 4), (20000, 64, 8), (9999, 32, 9), (3000, 128, 3),
                                   (2000, 256, 3), (1, 3, 3), (255, 7, 5), (40003, 3, 32), (30000, 4, 32), (20001, 3, 64), (20000, 8, 64), (257, 16, 36),
                                   (100, 1, 64), (50001, 32, 16), (20000, 64, 8), (30001, 3, 16), (5000, 9, 16), (777, 1, 3)])
@pytest.mark.parametrize("act", [0, ffi.ACT_RELU])
def test_linear_thin_layers(M, N, K, act):
    """One side of the weight <= 16 wide with dense rows (the point transformer's per-neighbour linears and their input gradients):
    csrc/gemm_thin.hip streams them through LDS with an f32 FMA chain per output instead of a 64 x 64 MFMA tile."""
    x = synth.gaussian("thin_x", (M, K)); w = synth.gaussian("thin_w", (N, K)) / math.sqrt(K)
    b, sc = synth.gaussian("thin_b", (N,)), 1 + 0.1 * synth.gaussian("thin_s", (N,))
    want = F.linear(x.double(), w.double()) * sc.double() + b.double()
    want = (F.relu(want) if act else want).float()
    got = ops.linear(x.to(dev()), w.to(dev()), b.to(dev()), act=act, scale=sc.to(dev()))
    report(f"thin linear {M}x{N}x{K} act={act}", got, want, 2e-5)
    plain = ops.linear(x.to(dev()), w.to(dev()))
    report(f"thin linear {M}x{N}x{K} no epilogue", plain, F.linear(x.double(), w.double()).float(), 2e-5)
    # the arithmetic does not depend on M: a shard of the rows computes the same bits
    if M > 1000:
        lo, hi = 256 * 3 + 4, min(M, 256 * 3 + 4 + 777)
        part = ops.linear(x[lo:hi].contiguous().to(dev()), w.to(dev()), b.to(dev()), act=act, scale=sc.to(dev()))
        assert torch.equal(part, got[lo:hi])
    # ... nor on where the rows start: an OFFSET VIEW of a row pool (not 16-byte aligned when K % 4 != 0) and offset bias / scale vectors run the
    # same FMA chain through scalar accesses (the choice used to depend on pointer alignment)
    xd = x.to(dev())
    for first in (1, 2, 3):
        view = xd[first:first + min(M - 3, 300)]
        assert view.is_contiguous()
        part = ops.linear(view, w.to(dev()), b.to(dev()), act=act, scale=sc.to(dev()))
        assert torch.equal(part, got[first:first + view.shape[0]]), (first, view.data_ptr() % 16)
    pad = torch.cat([torch.zeros(1), b]).to(dev())[1:]              # bias at a 4-byte offset
    assert pad.data_ptr() % 16 != 0
    assert torch.equal(ops.linear(xd, w.to(dev()), pad, act=act, scale=sc.to(dev())), got)


@pytest.mark.parametrize("M,N,K,R", [(5000, 256, 256, 6), (40000, 256, 256, 8), (33, 128, 256, 1), (777, 320, 512, 8), (1304, 512, 512, 1), (200, 256, 96, 3)])
def test_linear_rowdot_epilogue(M, N, K, R):
    """ABI v4 row-dot epilogue (the CDM's linear2 + contact_layer collapse): rowdot_out[m, g, r] = sum over the 64-column group g of
    GELU(x W^T + b)[m, col] * w[r, col].  Checked against the stored output in float64, required to be bit-identical across tile shapes
    and arithmetics' tile choices (the summation order inside a group is fixed), and usable without storing the output at all."""
    x = synth.gaussian("rd_x", (M, K)); w = synth.gaussian("rd_w", (N, K)) / math.sqrt(K); b = synth.gaussian("rd_b", (N,))
    rw = synth.gaussian("rd_rw", (R, N))
    xd, wd, bd, rwd = x.to(dev()), w.to(dev()), b.to(dev()), rw.to(dev())
    ngrp = (N + 63) // 64
    outs = {}
    for tile in (0, 3, 5):
        prev = ops.set_gemm_tune(tile << ffi.TUNE_TILE_SHIFT)
        try:
            rd = torch.full((M, ngrp, R), float("nan"), device=dev())
            c = ops.linear(xd, wd, bd, act=ffi.ACT_GELU, rowdot_w=rwd, rowdot_out=rd)
            rd2 = torch.full((M, ngrp, R), float("nan"), device=dev())
            junk = torch.full((M, N), 7.0, device=dev())
            ops.linear(xd, wd, bd, act=ffi.ACT_GELU, rowdot_w=rwd, rowdot_out=rd2, out=junk, store=False)
        finally:
            ops.set_gemm_tune(prev)
        assert torch.equal(rd, rd2) and bool((junk == 7.0).all())          # store=False leaves the output buffer alone
        outs[tile] = (c, rd)
    c0, rd0 = outs[0]
    assert torch.equal(c0, ops.linear(xd, wd, bd, act=ffi.ACT_GELU))
    slab = K == 256 and N % 64 == 0            # tile 0 = the weight-stationary form (csrc/gemm_slab.hip): same products, its own row-dot tree
    assert torch.equal(outs[3][1], outs[5][1]), "row-dot partials depend on the tile shape"
    assert torch.equal(outs[3][0], c0) and torch.equal(outs[5][0], c0)
    if not slab:
        assert torch.equal(outs[3][1], rd0), "row-dot partials depend on the tile shape"
    cpad = torch.zeros(M, ngrp * 64, dtype=torch.float64); cpad[:, :N] = c0.double().cpu()
    wpad = torch.zeros(R, ngrp * 64, dtype=torch.float64); wpad[:, :N] = rw.double()
    want = torch.einsum("mgc,rgc->mgr", cpad.view(M, ngrp, 64), wpad.view(R, ngrp, 64))
    scale = torch.einsum("mgc,rgc->mgr", cpad.view(M, ngrp, 64).abs(), wpad.view(R, ngrp, 64).abs()) + 1e-30
    err = ((rd0.double().cpu() - want).abs() / scale).max().item()
    err3 = ((outs[3][1].double().cpu() - want).abs() / scale).max().item()
    print(f"row-dot {M}x{N}x{K} R={R}: max err / sum|c||w| = {err:.2e} (staged 64x64 tiles: {err3:.2e}; weight-stationary form: {slab})")
    assert err < 5e-7 and err3 < 5e-7
    saved = ops.get_gemm_split()                                               # native f32 MFMA kernels share the epilogue
    try:
        ops.set_gemm_split(0, 0)
        rdn = torch.empty((M, ngrp, R), device=dev())
        cn = ops.linear(xd, wd, bd, act=ffi.ACT_GELU, rowdot_w=rwd, rowdot_out=rdn)
    finally:
        ops.set_gemm_split(*saved)
    wantn = torch.einsum("mgc,rgc->mgr", torch.nn.functional.pad(cn.double().cpu(), (0, ngrp * 64 - N)).view(M, ngrp, 64), wpad.view(R, ngrp, 64))
    assert ((rdn.double().cpu() - wantn).abs() / scale).max().item() < 5e-7
    with pytest.raises(ffi.AfmError):                                          # N % 4 != 0: the row-dot form needs 16-byte rows
        ops.linear(xd, wd[:N - 1].contiguous(), None, rowdot_w=rwd[:, :N - 1].contiguous(), rowdot_out=rd)


@pytest.mark.parametrize("M,N,K", [(10432, 512, 512), (1304, 512, 1024), (777, 256, 256), (6272, 512, 512), (100, 1024, 128), (33, 64, 96)])
def test_linear_fused_layernorm_is_bit_identical_to_two_launches(M, N, K):
    """ABI v5: LayerNorm of the output rows inside the GEMM (the workgroup finishing the last column tile of a row block normalises it,
    agent-scope release / ticket / acquire).  Must equal afm_layernorm of the stored output bit for bit, for every arithmetic and tile
    shape, through a row remap, on ragged M, and repeatedly (the ticket words return to zero after every launch)."""
    x = synth.gaussian("fln_x", (M, K)); w = synth.gaussian("fln_w", (N, K)) / math.sqrt(K); b = synth.gaussian("fln_b", (N,))
    res = synth.gaussian("fln_r", (M, N)); g = synth.gaussian("fln_g", (N,)) * 0.2 + 1.0; be = synth.gaussian("fln_be", (N,)) * 0.1
    xd, wd, bd, rd, gd, bed = (t.to(dev()) for t in (x, w, b, res, g, be))
    cnt = torch.zeros((M + 31) // 32, dtype=torch.int32, device=dev())
    saved = ops.get_gemm_split()
    try:
        for products in (9, 0):
            ops.set_gemm_split(products, 0)
            for tile in ((0, 3, 5, 7) if products else (0, 1, 2, 3, 5)):
                prev = ops.set_gemm_tune(tile << ffi.TUNE_TILE_SHIFT)
                try:
                    try:
                        plain = ops.linear(xd, wd, bd, residual=rd)
                    except ffi.AfmError:
                        continue                                    # this (tile, shape) combination does not exist (split-K needs K % 256 == 0)
                    want = ops.layernorm(plain, gd, bed, 1e-5)
                    for rep in range(3):
                        ln_out = torch.full((M, N), float("nan"), device=dev())
                        got_c = ops.linear(xd, wd, bd, residual=rd, ln=(gd, bed, 1e-5), ln_out=ln_out, ln_counters=cnt)
                        assert torch.equal(got_c, plain), (products, tile, rep)
                        assert torch.equal(ln_out, want), f"arith x{products} tile {tile} rep {rep}: max diff {(ln_out - want).abs().max().item():.3e}, nan {int(torch.isnan(ln_out).sum())}"
                        assert int(cnt.abs().sum()) == 0
                finally:
                    ops.set_gemm_tune(prev)
    finally:
        ops.set_gemm_split(*saved)
    ref = F.layer_norm(F.linear(x.double(), w.double(), b.double()) + res.double(), (N,), g.double(), be.double(), 1e-5).float()
    report(f"fused linear+LN {M}x{N}x{K}", ln_out, ref, 2e-4)


@pytest.mark.parametrize("M,tile", [(10432, 0), (1304, 0), (777, 3), (5216, 5), (326, 7), (5216, 13), (3136, 13)])
def test_linear_layernorm_folded_across_launches(M, tile):
    """ABI v5 (afm_linear_args.stat_out / a_stat / res_stat): a post-LN encoder layer without LayerNorm launches.  The producer stores its
    RAW output and per-row, per-64-column (mean, M2); the next linear runs on the raw rows with gamma folded into its weight and applies
    (mean, rstd) in its epilogue; the residual add normalises the raw rows on the fly.  A re-association of LayerNorm -> Linear: against
    float64 the error stays at the level of the separate-launch path."""
    d, ff = 512, 1024
    att = synth.gaussian("lnf_att", (M, d)); xin = synth.gaussian("lnf_x", (M, d)) * 2.0 + 0.7            # a residual stream with a non-zero mean
    wo = synth.gaussian("lnf_wo", (d, d)) / math.sqrt(d); bo = synth.gaussian("lnf_bo", (d,)) * 0.1
    g1 = synth.gaussian("lnf_g1", (d,)) * 0.2 + 1.0; b1 = synth.gaussian("lnf_b1", (d,)) * 0.1
    w1 = synth.gaussian("lnf_w1", (ff, d)) / math.sqrt(d); c1 = synth.gaussian("lnf_c1", (ff,)) * 0.1
    w2 = synth.gaussian("lnf_w2", (d, ff)) / math.sqrt(ff); c2 = synth.gaussian("lnf_c2", (d,)) * 0.1
    D = lambda t: t.to(dev())
    prev = ops.set_gemm_tune(tile << ffi.TUNE_TILE_SHIFT)
    try:
        # float64 reference of out_proj + residual -> LN1 -> linear1 (GELU) -> linear2 + LN1 output as residual
        t1_ref = F.linear(att.double(), wo.double(), bo.double()) + xin.double()
        x1_ref = F.layer_norm(t1_ref, (d,), g1.double(), b1.double(), 1e-5)
        h_ref = F.gelu(F.linear(x1_ref, w1.double(), c1.double()))
        t2_ref = F.linear(h_ref, w2.double(), c2.double()) + x1_ref
        # separate-launch path
        t1 = ops.linear(D(att), D(wo), D(bo), residual=D(xin))
        x1 = ops.layernorm(t1, D(g1), D(b1), 1e-5)
        h = ops.linear(x1, D(w1), D(c1), act=ffi.ACT_GELU)
        t2 = ops.linear(h, D(w2), D(c2), residual=x1)
        # folded path
        st1 = torch.full((M, d // 64, 2), float("nan"), device=dev())
        t1f = ops.linear(D(att), D(wo), D(bo), residual=D(xin), stat_out=st1)
        assert torch.equal(t1f, t1)                                            # the raw output is the same GEMM
        grp = t1.double().cpu().view(M, d // 64, 64)
        assert (st1[..., 0].double().cpu() - grp.mean(-1)).abs().max().item() < 1e-5
        assert ((st1[..., 1].double().cpu() - ((grp - grp.mean(-1, keepdim=True)) ** 2).sum(-1)).abs() / (grp.var(-1, unbiased=False) * 64 + 1e-9)).max().item() < 1e-5
        w1g = (w1.double() * g1.double()[None, :])
        hf = ops.linear(t1f, D(w1g.float()), D((c1.double() + w1.double() @ b1.double()).float()), act=ffi.ACT_GELU, a_stat=(st1, D(w1g.sum(1).float())))
        st2 = torch.empty((M, d // 64, 2), device=dev())
        t2f = ops.linear(hf, D(w2), D(c2), residual=t1f, res_stat=(st1, D(g1), D(b1)), stat_out=st2)
    finally:
        ops.set_gemm_tune(prev)
    e_sep = (h.double().cpu() - h_ref).abs().max().item(), (t2.double().cpu() - t2_ref).abs().max().item()
    e_fold = (hf.double().cpu() - h_ref).abs().max().item(), (t2f.double().cpu() - t2_ref).abs().max().item()
    print(f"LN folded across launches M={M} tile={tile}: linear1 err {e_fold[0]:.2e} (separate {e_sep[0]:.2e}), linear2 err {e_fold[1]:.2e} (separate {e_sep[1]:.2e})")
    assert e_fold[0] < max(4 * e_sep[0], 2e-5) and e_fold[1] < max(4 * e_sep[1], 4e-5)
    with pytest.raises(ffi.AfmError):                                          # the native f32 kernels do not carry the statistics
        saved = ops.get_gemm_split()
        try:
            ops.set_gemm_split(0, 0)
            ops.linear(D(att), D(wo), D(bo), residual=D(xin), stat_out=st1)
        finally:
            ops.set_gemm_split(*saved)


def test_linear_fused_layernorm_through_a_row_map():
    """The last encoder layer runs out_proj / FFN on the motion rows only: output rows (and their LayerNorm) are scattered by the c_* remap."""
    B, L, T, N, K = 5, 50, 83, 512, 512
    x = synth.gaussian("flm_x", (B * L, K)).to(dev()); w = (synth.gaussian("flm_w", (N, K)) / math.sqrt(K)).to(dev())
    g = (synth.gaussian("flm_g", (N,)) * 0.2 + 1.0).to(dev()); be = (synth.gaussian("flm_be", (N,)) * 0.1).to(dev())
    out = torch.zeros(B * T, N, device=dev()); ln_out = torch.zeros(B * T, N, device=dev())
    ops.linear(x, w, None, out=out, c_map=(L, T, T - L), rows=B * L, ln=(g, be, 1e-5), ln_out=ln_out)
    want_c = ops.linear(x, w, None)
    want = ops.layernorm(want_c, g, be, 1e-5)
    assert torch.equal(out.view(B, T, N)[:, T - L:], want_c.view(B, L, N)) and torch.equal(ln_out.view(B, T, N)[:, T - L:], want.view(B, L, N))
    assert float(ln_out.view(B, T, N)[:, :T - L].abs().sum()) == 0.0          # rows outside the map are untouched


def test_linear_bf16_one_product_is_bf16_accurate_only():
    """AFM_ARITH_BF16X1 (informational): the leading-term product alone carries bf16's 2^-9 relative rounding per operand - three orders
    of magnitude above the f32 kernels' error, and identical across tile shapes like every other arithmetic."""
    M, N, K = 5216, 512, 512
    x = synth.gaussian("b1_x", (M, K)); w = synth.gaussian("b1_w", (N, K)) / math.sqrt(K)
    ref = F.linear(x.double(), w.double())
    scale = x.double().abs() @ w.double().abs().t() + 1e-30
    saved, prev = ops.get_gemm_split(), 0
    try:
        ops.set_gemm_split(1, 0)
        got = ops.linear(x.to(dev()), w.to(dev()), None)
        prev = ops.set_gemm_tune(7 << ffi.TUNE_TILE_SHIFT)
        got_k = ops.linear(x.to(dev()), w.to(dev()), None)
    finally:
        ops.set_gemm_tune(prev)
        ops.set_gemm_split(*saved)
    err = ((got.double().cpu() - ref).abs() / scale).max().item()
    print(f"bf16 one-product {M}x{N}x{K}: max err / sum|a||w| = {err:.2e}")
    assert 1e-5 < err < 1e-2
    assert torch.equal(got, got_k)


@pytest.mark.parametrize("products", [9, 6])
@pytest.mark.parametrize("M,N,K", [(10432, 512, 1024), (9000, 1500, 256), (1000, 1536, 512), (5216, 512, 512), (777, 263, 512), (640, 96, 144)])
def test_linear_split_bf16_mode(products, M, N, K):
    """afm_linear_args.arith = AFM_ARITH_BF16X9 / X6: f32 operands split exactly into three bf16 terms, products on the bf16 matrix pipe, f32 accumulate.
    The error against float64 must stay at the native f32 MFMA kernel's level (same products, different summation order), on
    both tile variants (128x128 / 64x64), with ragged M / N edges and every epilogue input."""
    x = synth.gaussian("sp_x", (M, K)); w = synth.gaussian("sp_w", (N, K)) / math.sqrt(K); b = synth.gaussian("sp_b", (N,))
    res = synth.gaussian("sp_r", (M, N))
    x[::7] *= 50.0                                        # mixed magnitudes: the residual terms matter
    ref = F.gelu(F.linear(x.double(), w.double(), b.double())) + res.double()
    scale = (x.double().abs() @ w.double().abs().t()) + b.double().abs() + 1e-30
    args = (x.to(dev()), w.to(dev()), b.to(dev()))
    saved = ops.get_gemm_split()
    try:
        ops.set_gemm_split(0, 0)
        native = ops.linear(*args, act=ffi.ACT_GELU, residual=res.to(dev()))
        assert ops.set_gemm_split(products, 0) == (0, 0)
        got = ops.linear(*args, act=ffi.ACT_GELU, residual=res.to(dev()))
        assert ops.get_gemm_split() == (products, 0)
    finally:
        ops.set_gemm_split(*saved)
    e_split = ((got.double().cpu() - ref).abs() / scale).max().item()
    e_native = ((native.double().cpu() - ref).abs() / scale).max().item()
    print(f"split x{products} {M}x{N}x{K}: max err / sum|a||w| = {e_split:.2e} (native f32 MFMA {e_native:.2e})")
    assert e_split <= max(1.5 * e_native, 3e-7)
    assert (got - native).abs().max().item() <= 1e-5 * native.abs().max().item()
    assert not torch.equal(got, native)                   # a different kernel really ran
    with pytest.raises(ffi.AfmError):                      # invalid settings are rejected on the host, state unchanged
        ops.set_gemm_split(5)
    with pytest.raises(ffi.AfmError):
        ops.set_gemm_split(9, -3)
    assert ops.get_gemm_split() == saved
    import ctypes as C                                     # and by the library: the arithmetic is a validated field of the arguments
    bad = ffi.LinearArgs()
    bad.A, bad.W, bad.C, bad.M, bad.N, bad.K, bad.lda, bad.ldw, bad.ldc = args[0].data_ptr(), args[1].data_ptr(), got.data_ptr(), 4, N, K, K, K, N
    bad.arith = 5
    assert ffi.load().afm_linear(C.byref(bad), None) == -1      # AFM_E_BADARG


def test_linear_split_dispatch_is_independent_of_m():
    """The split path is chosen from (N, K) only, so rows computed in a small batch equal the same rows of a large batch bit for bit."""
    x = synth.gaussian("spm_x", (4096, 512)); w = synth.gaussian("spm_w", (1536, 512)) / math.sqrt(512)
    big = ops.linear(x.to(dev()), w.to(dev()))            # 128x128-tile instantiation; the slices below run the 64x64 one
    for rows in (1, 33, 200, 640):
        part = ops.linear(x[:rows].to(dev()), w.to(dev()))
        assert torch.equal(part, big[:rows]), rows


@pytest.mark.parametrize("M,N,K", [(1304, 512, 512), (1304, 512, 1024), (784, 263, 512), (2608, 512, 512), (10432, 512, 512), (97, 96, 64)])
def test_linear_tile_shapes_are_bit_identical(M, N, K):
    """Strong scaling runs B = 4 per GPU (M = 1304 rows): afm_linear then picks other workgroup shapes than at B = 32 (native kernel: 32x32 /
    32x64 tiles; bf16-split kernel: 64x64 tiles whose 256-deep K segments run on separate wave groups of one workgroup).  Every shape of one
    arithmetic adds the same numbers in the same order (K segments summed left to right), so the choice (which depends on M) must not
    change a single bit - the sharding / sub-batch invariance of the loop rests on it."""
    x = synth.gaussian("tile_x", (M, K)).to(dev()); w = (synth.gaussian("tile_w", (N, K)) / math.sqrt(K)).to(dev())
    b = synth.gaussian("tile_b", (N,)).to(dev()); res = synth.gaussian("tile_r", (M, N)).to(dev())
    saved, saved_tune = ops.get_gemm_split(), ops.set_gemm_tune(0)
    try:
        kg = (7,) if K % 256 == 0 and 2 <= K // 256 <= 4 else ()       # split-K form: the K segments of a tile on separate wave groups
        if K // 256 in (2, 4) and K % 256 == 0:
            kg += (10,)                                                  # round 5: the same on three LDS stages (K = 1024: two groups x two segments)
        if K == 1024:
            kg += (11,)
        if K >= 128 and K % 16 == 0:
            kg += (9, 12, 13)                                            # sequential 64x64 on three LDS stages; 64x64 tiles walked by resident workgroups; round 6: 256x128 tiles on 512 threads
        for products, tiles in ((0, (0, 1, 2, 3, 4, 5)), (9, (0, 3, 5) + kg), (6, (0, 3, 5) + kg)):
            ops.set_gemm_split(products, 0)
            outs = []
            for tile in tiles:
                ops.set_gemm_tune(tile << ffi.TUNE_TILE_SHIFT)
                outs.append(ops.linear(x, w, b, act=ffi.ACT_GELU, residual=res))
            for tile, o in zip(tiles[1:], outs[1:]):
                assert torch.equal(o, outs[0]), f"arith {products}: tile {tile} differs from the heuristic's choice on {M}x{N}x{K}"
            ops.set_gemm_tune(ffi.TUNE_NO_DMA)              # register-staged operand path: same k order as the LDS-DMA kernels
            if products == 0:
                assert torch.equal(ops.linear(x, w, b, act=ffi.ACT_GELU, residual=res), outs[0])
            # a shard of the rows computes the same bits as the whole matrix (other tile shape chosen from the smaller M)
            ops.set_gemm_tune(0)
            part = ops.linear(x[: M // 3], w, b, act=ffi.ACT_GELU, residual=res[: M // 3])
            assert torch.equal(part, outs[0][: M // 3])
    finally:
        ops.set_gemm_split(*saved)
        ops.set_gemm_tune(saved_tune)


@pytest.mark.parametrize("products", [6, 9])
@pytest.mark.parametrize("M0,M1", [(5216, 5216), (3136, 3136), (700, 129), (64, 5000)])
def test_linear_pair_is_bit_identical_to_two_launches(products, M0, M1):
    """afm_linear_pair (ABI v7): two independent launches - one sub-batch's out_proj (N = 512, K = 512, bias + residual + row statistics) and the
    other's linear1 (N = 1024, K = 512, folded-LayerNorm input statistics, GELU) - as ONE grid of 128 x 128 tiles.  Same tile program, same
    operands, same order: every output bit equals the two afm_linear calls', for ragged row counts on either side and both arithmetics;
    a pair the form does not take (native-path operand) is refused with AFM_E_UNSUPPORTED, not computed differently."""
    d, ff = 512, 1024
    g = lambda n, *s: synth.gaussian(f"pair_{n}", s).to(dev())
    x0, w0, b0, r0 = g("x0", M0, d), g("w0", d, d) / math.sqrt(d), g("b0", d), g("r0", M0, d)
    x1, w1, b1 = g("x1", M1, d), g("w1", ff, d) / math.sqrt(d), g("b1", ff)
    # statistics of x1's rows as a producer would have written them (mean, M2 per 64-column group), folded-LayerNorm vector for linear1
    grp = x1.view(M1, d // 64, 64)
    st1 = torch.stack([grp.mean(-1), ((grp - grp.mean(-1, keepdim=True)) ** 2).sum(-1)], -1).contiguous()
    fold = g("fold", ff)
    saved = ops.get_gemm_split()
    try:
        ops.set_gemm_split(products, 0)
        st_sep, st_pair = torch.zeros(M0, d // 64, 2, device=dev()), torch.zeros(M0, d // 64, 2, device=dev())
        first = lambda st: dict(x=x0, weight=w0, bias=b0, residual=r0, stat_out=st)
        second = dict(x=x1, weight=w1, bias=b1, act=ffi.ACT_GELU, a_stat=(st1, fold))
        want0, want1 = ops.linear(**first(st_sep)), ops.linear(**second)
        got0, got1 = ops.linear_pair(first(st_pair), second)
        assert torch.equal(got0, want0) and torch.equal(got1, want1) and torch.equal(st_pair, st_sep)
        assert torch.isfinite(got0).all() and torch.isfinite(got1).all()
        with pytest.raises(ffi.AfmError):                       # K = 100 is not on the bf16-split path: no paired form
            ops.linear_pair(dict(x=x0[:, :100].contiguous(), weight=w0[:, :100].contiguous()), second)
    finally:
        ops.set_gemm_split(*saved)


def test_linear_detects_transposed_layouts():
    # asymmetric operands + identity A (guide: always A=I-check with asymmetric B)
    K = N = 64
    w = torch.arange(N * K, dtype=torch.float32).reshape(N, K) / 100.0
    got = ops.linear(torch.eye(K, device=dev()), w.to(dev()))
    report("linear identity", got, w.t().contiguous(), 1e-6)


@pytest.mark.parametrize("act,fn", [(ffi.ACT_GELU, F.gelu), (ffi.ACT_RELU, F.relu), (ffi.ACT_SILU, F.silu)])
def test_linear_epilogues(act, fn):
    M, N, K = 300, 192, 128
    x, w = synth.gaussian("e_x", (M, K)), synth.gaussian("e_w", (N, K)) / math.sqrt(K)
    b, sc, res = synth.gaussian("e_b", (N,)), 1 + 0.1 * synth.gaussian("e_s", (N,)), synth.gaussian("e_r", (M, N))
    tab = synth.gaussian("e_t", (7, N))
    want = fn((x.double() @ w.double().t()) * sc.double() + b.double()) + res.double() + tab.double()[torch.arange(M) % 7]
    got = ops.linear(x.to(dev()), w.to(dev()), b.to(dev()), act=act, scale=sc.to(dev()), residual=res.to(dev()), rowtab=tab.to(dev()))
    report(f"linear epilogue act={act}", got, want.float(), 2e-4)


def test_linear_row_remaps():
    # gather motion tokens out of / scatter into a [B, T, d] token pool (cmdm.py:161,169)
    B, L, T, off, K, N = 3, 5, 9, 4, 64, 32
    pool = synth.gaussian("rm_pool", (B * T, K)); w = synth.gaussian("rm_w", (N, K)) / 8
    want = F.linear(pool.view(B, T, K)[:, off:off + L].reshape(B * L, K), w)
    got = ops.linear(pool.to(dev()), w.to(dev()), rows=B * L, a_map=(L, T, off))
    report("linear a_map gather", got, want, 1e-4)
    x = synth.gaussian("rm_x", (B * L, K))
    out = torch.zeros(B * T, N, device=dev())
    ops.linear(x.to(dev()), w.to(dev()), out=out, c_map=(L, T, off))
    want2 = torch.zeros(B, T, N); want2[:, off:off + L] = F.linear(x, w).view(B, L, N)
    report("linear c_map scatter", out, want2.view(B * T, N), 1e-4)


def test_encoder_layer_golden():
    """One TransformerEncoderLayer (post-LN, key padding mask) against the REFERENCE's output."""
    from oracle import shapes as sh
    g = golden("encoder_layer_T24")
    sd = {k: v.to(dev()) for k, v in sh.weights(sh.encoder_layer("layers.0")).items()}
    p = "layers.0."
    x = g["x"].to(dev()); B, T, D = x.shape
    qkv = ops.linear(x, sd[p + "self_attn.in_proj_weight"], sd[p + "self_attn.in_proj_bias"])
    a = ops.mha(qkv, g["mask"].to(dev()), 8)
    y = ops.linear(a, sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"], residual=x)
    x1 = ops.layernorm(y, sd[p + "norm1.weight"], sd[p + "norm1.bias"])
    h = ops.linear(x1, sd[p + "linear1.weight"], sd[p + "linear1.bias"], act=ffi.ACT_GELU)
    y2 = ops.linear(h, sd[p + "linear2.weight"], sd[p + "linear2.bias"], residual=x1)
    out = ops.layernorm(y2, sd[p + "norm2.weight"], sd[p + "norm2.bias"])
    report("encoder layer vs reference golden", out, g["out"], 1e-4)


@pytest.mark.parametrize("B,T,masked", [(2, 326, True), (1, 190, False), (3, 37, True), (2, 32, False), (1, 420, True)])
def test_mha(B, T, masked):
    H, dh = 8, 64
    qkv = synth.gaussian("mha_qkv", (B, T, 3 * H * dh))
    mask = None
    if masked:
        mask = torch.zeros(B, T, dtype=torch.bool)
        for b in range(B):
            mask[b, T - 1 - 7 * b - T // 3:] = True
        mask[0, 5] = True                                    # a hole in the middle, not only suffix padding
    q, k, v = (z.view(B, T, H, dh).transpose(1, 2).double() for z in qkv.chunk(3, -1))
    s = q @ k.transpose(-1, -2) / math.sqrt(dh)
    if mask is not None:
        s = s.masked_fill(mask[:, None, None, :], float("-inf"))
    want = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, T, H * dh).float()
    got = ops.mha(qkv.to(dev()), None if mask is None else mask.to(dev()), H)
    report(f"mha B{B} T{T} masked={masked}", got, want, 2e-5)


@pytest.mark.parametrize("B,T", [(4, 326), (1, 326), (2, 196), (3, 61), (2, 32), (17, 326), (2, 33)])
def test_mha_workgroup_groupings_are_bit_identical(B, T):
    """grid = (sample, head, query group): groups of 1 / 2 / 4 / 6 / 8 / 12 waves (and the automatic choice, which depends on B) must give the
    same bits - a query row's arithmetic never depends on which workgroup it ran in.  Round 4: nor on whether the two key segments of its
    softmax ran on one wave (one after the other) or on two (codes 102 / 104: workgroups of 2 x 2 / 2 x 4 waves, the small-launch form)."""
    H, dh = 8, 64
    qkv = synth.gaussian("mha_grp", (B, T, 3 * H * dh)).to(dev())
    mask = torch.zeros(B, T, dtype=torch.bool)
    for b in range(B):
        mask[b, T - 1 - 11 * b:] = True
    mask = mask.to(dev())
    for km in (None, mask):
        outs = [ops.mha(qkv, km, H, group_waves=g) for g in (0, 1, 2, 4, 6, 8, 12, 102, 104)]
        for g, o in zip((1, 2, 4, 6, 8, 12, 102, 104), outs[1:]):
            assert torch.equal(o, outs[0]), f"group_waves={g} differs (B={B}, T={T}, masked={km is not None})"
    with pytest.raises(ffi.AfmError):
        ops.mha(qkv, None, H, group_waves=3)


def test_split_reconstructs_f32_exactly():
    """The three-way bf16 split is EXACT: x = x1 + x2 + x3 bit for bit.  A GEMM against the identity returns sum_k A[m][k] I[n][k] = the three
    terms of A[m][n] (times 1 = the only non-zero term of the identity's split) added in f32, so C == A bit for bit iff every split is exact -
    for both operand roles (A x I and I x A^T), over values that stress the residual arithmetic (round 4: the residuals come from
    v_dot2c_f32_bf16): powers of two and their neighbours, numbers one ulp around bf16 rounding ties, tiny and huge magnitudes, both signs."""
    K = 512
    g = torch.Generator().manual_seed(3)
    base = torch.randn(768, K, generator=g)
    edge = torch.tensor([1.0, 1.0 + 2**-7, 1.0 + 2**-8, 1.0 + 2**-8 + 2**-23, 1.0 + 2**-8 - 2**-23, 1.5 - 2**-23, 2.0 - 2**-23, 0.999999, 3.0e-30, 7.0e29, 65504.0,
                         1.0 + 2**-16, 1.0 + 2**-15 + 2**-23, 255.99998, 0.0, 1.17549435e-38 * 1024])
    x = base.clone()
    x[:64] = edge.repeat(64 * K // edge.numel() + 1)[: 64 * K].view(64, K) * torch.where(torch.rand(64, K, generator=g) < 0.5, -1.0, 1.0)
    x[64:128] *= torch.exp2(torch.randint(-60, 60, (64, K), generator=g).float())
    x = x.to(dev())
    eye = torch.eye(K, device=dev())
    got = ops.linear(x, eye)                                  # A = x (activation role), W = I
    assert torch.equal(got, x), f"A-side split inexact: {(got != x).sum().item()} elements, max diff {(got - x).abs().max().item():.3e}"
    got_t = ops.linear(eye, x[:512])                          # A = I, W = x (weight role): C[m][n] = x[n][m]
    assert torch.equal(got_t, x[:512].t()), f"W-side split inexact: {(got_t != x[:512].t()).sum().item()} elements"


def test_linear_row_maps_with_a_hole():
    """ABI v6 (afm_linear_args.a_skip / c_skip): one launch over the time token and the L motion tokens of every sample, skipping the
    n_cond condition tokens between them - layer 0's in_proj of the sampling loop.  Rows are independent: bit-identical to the two
    launches it replaces (M = B and M = B L), for the folded-statistics epilogue's row indexing too."""
    B, L, nc, K, N = 3, 21, 7, 512, 1536
    T = 1 + nc + L
    x = synth.gaussian("hole_x", (B * T, K)).to(dev())
    w, b = synth.gaussian("hole_w", (N, K)).to(dev()) * 0.05, synth.gaussian("hole_b", (N,)).to(dev())
    want = torch.full((B * T, N), 7.0, device=dev())
    ops.linear(x, w, b, out=want, rows=B, a_map=(1, T, 0), c_map=(1, T, 0))
    ops.linear(x, w, b, out=want, rows=B * L, a_map=(L, T, 1 + nc), c_map=(L, T, 1 + nc))
    got = torch.full((B * T, N), 7.0, device=dev())
    ops.linear(x, w, b, out=got, rows=B * (1 + L), a_map=(1 + L, T, 0, 1, nc), c_map=(1 + L, T, 0, 1, nc))
    assert torch.equal(got, want)
    keep = got.view(B, T, N)[:, 1:1 + nc]
    assert (keep == 7.0).all()                               # the condition tokens' rows are not touched
    ref = x.view(B, T, K)[:, [0] + list(range(1 + nc, T))].cpu().double() @ w.cpu().double().T + b.cpu().double()
    report("linear with a hole in the row map", got.view(B, T, N)[:, [0] + list(range(1 + nc, T))], ref.float(), 2e-4)


@pytest.mark.parametrize("B,T,q_first", [(4, 326, 130), (1, 326, 130), (2, 196, 7), (3, 61, 60), (2, 326, 320)])
def test_mha_query_rows_subset_is_bit_identical(B, T, q_first):
    """afm_mha_fwd_rows (ABI v6): the query rows q_first .. T - 1 only (the CMDM's last layer is read on its motion tokens only).  The query
    blocks start at q_first, i.e. rows land in other lanes / waves / workgroups than in the full launch - the rows that are computed must
    carry the same bits for every grouping, and the rows in front must stay untouched.  The ragged key tail (T = 326: 6 keys in the last
    32-key block, second K16 step of P V skipped) and masks that empty the upper half of other blocks ride along."""
    H, dh = 8, 64
    qkv = synth.gaussian("mha_rows", (B, T, 3 * H * dh)).to(dev())
    mask = torch.zeros(B, T, dtype=torch.bool)
    for b in range(B):
        mask[b, T - 1 - 13 * b:] = True
        mask[b, 48:64] = True                                 # block 1 keeps valid keys in its lower half only
    mask[0, 5] = True
    mask = mask.to(dev())
    for km in (None, mask):
        full = ops.mha(qkv, km, H)
        for g in (0, 1, 2, 4, 6, 8, 12, 102, 104):
            part = ops.mha(qkv, km, H, group_waves=g, q_first=q_first)
            assert torch.equal(part[:, q_first:], full[:, q_first:]), f"q_first={q_first} group_waves={g} masked={km is not None}"
            assert not part[:, :q_first].any()
    with pytest.raises(ffi.AfmError):
        ops.mha(qkv, None, H, q_first=T)


def test_mha_softmax_rescale_branch():
    """Force the running max to jump in a late key block (spiked key) - exercises the online-softmax rescale."""
    B, T, H, dh = 1, 160, 8, 64
    qkv = synth.gaussian("mha_spike", (B, T, 3 * H * dh)) * 0.3
    qkv[0, 150, H * dh:2 * H * dh] = qkv[0, 3, :H * dh] * 40.0      # key 150 aligned with query 3, huge score
    q, k, v = (z.view(B, T, H, dh).transpose(1, 2).double() for z in qkv.chunk(3, -1))
    want = (torch.softmax(q @ k.transpose(-1, -2) / 8.0, -1) @ v).transpose(1, 2).reshape(B, T, H * dh).float()
    report("mha rescale", ops.mha(qkv.to(dev()), None, H), want, 2e-5)


@pytest.mark.parametrize("rows,dim", [(10, 512), (1001, 512), (64, 256), (7, 1024), (33, 128)])
def test_layernorm(rows, dim):
    x = synth.gaussian("ln_x", (rows, dim)) * 3 + 1
    w, b = 1 + 0.1 * synth.gaussian("ln_w", (dim,)), synth.gaussian("ln_b", (dim,))
    report(f"layernorm {rows}x{dim}", ops.layernorm(x.to(dev()), w.to(dev()), b.to(dev())), F.layer_norm(x, (dim,), w, b, 1e-5), 2e-5)


def test_ddpm_step_bit_exact():
    """Same float32 expression as gaussian_diffusion.py:222-225,439 -> bit-identical to torch-CPU."""
    from oracle import diffusion_ref as df
    s = df.Schedule(1000)
    B, L, D = 4, 196, 263
    x0, xt, nz = (synth.gaussian(n, (B, L, D)) for n in ("dd_x0", "dd_xt", "dd_nz"))
    t = torch.tensor([999, 500, 1, 0])
    c1 = df._extract(s.posterior_mean_coef1, t, x0.shape); c2 = df._extract(s.posterior_mean_coef2, t, x0.shape)
    lv = df._extract(s.posterior_log_variance_clipped, t, x0.shape)
    nzm = (t != 0).float().view(-1, 1, 1)
    want = (c1 * x0 + c2 * xt) + nzm * torch.exp(0.5 * lv) * nz
    f = lambda a: torch.from_numpy(a).float()[t]
    sig = (t != 0).float() * torch.exp(0.5 * f(s.posterior_log_variance_clipped))
    got = ops.ddpm_step(x0.to(dev()), xt.to(dev()), nz.to(dev()), f(s.posterior_mean_coef1).to(dev()),
                        f(s.posterior_mean_coef2).to(dev()), sig.to(dev()))
    assert torch.equal(got.cpu(), want), (got.cpu() - want).abs().max()


def test_philox_randn_statistics_and_sharding_invariance():
    B, per = 8, 196 * 263
    a = ops.randn((B, per), dev(), seed=7, sample_index0=0, step=3).cpu()
    assert abs(a.mean().item()) < 5e-3 and abs(a.std().item() - 1) < 5e-3
    assert abs(((a ** 4).mean() / a.var() ** 2).item() - 3.0) < 0.05              # kurtosis of a normal
    lo = ops.randn((4, per), dev(), seed=7, sample_index0=0, step=3).cpu()
    hi = ops.randn((4, per), dev(), seed=7, sample_index0=4, step=3).cpu()
    assert torch.equal(torch.cat([lo, hi]), a)                                     # invariant to how B is sharded
    assert not torch.equal(ops.randn((B, per), dev(), seed=7, step=4).cpu(), a)
    assert abs(np.corrcoef(a[0].numpy(), a[1].numpy())[0, 1]) < 0.02


def test_masked_mse():
    B, L, D = 3, 20, 263
    a, b = synth.gaussian("mse_a", (B, L, D)), synth.gaussian("mse_b", (B, L, D))
    m = torch.zeros(B, L, dtype=torch.bool); m[0, 12:] = True; m[2, 3:] = True
    keep = (~m).float().unsqueeze(-1)
    want = (((a - b) ** 2) * keep).sum((1, 2)) / (keep.sum((1, 2)) * D)
    report("masked mse", ops.masked_mse(a.to(dev()), b.to(dev()), m.to(dev())), want, 1e-5)
    report("unmasked mse", ops.masked_mse(a.to(dev()), b.to(dev()), None), ((a - b) ** 2).mean((1, 2)), 1e-5)


def test_empty_and_degenerate_inputs():
    """Empty batches are no-ops; bad arguments are rejected with AfmError instead of launching."""
    d = dev()
    assert ops.linear(torch.zeros(0, 64, device=d), torch.zeros(32, 64, device=d)).shape == (0, 32)
    assert ops.mha(torch.zeros(0, 8, 3 * 512, device=d), None, 8).shape == (0, 8, 512)
    assert ops.layernorm(torch.zeros(0, 512, device=d), torch.ones(512, device=d), torch.zeros(512, device=d)).shape == (0, 512)
    with pytest.raises(ffi.AfmError):                      # head dim other than 64 is unsupported, loudly
        ops.mha(torch.zeros(1, 8, 3 * 256, device=d), None, 8)
    ln6 = ops.layernorm(synth.gaussian("deg_ln", (2, 6)).to(d), torch.ones(6, device=d), torch.zeros(6, device=d))   # any width is supported
    report("LayerNorm dim 6", ln6, torch.nn.functional.layer_norm(synth.gaussian("deg_ln", (2, 6)), (6,)), 1e-5)
    with pytest.raises(ffi.AfmError):                      # CPU tensors never reach a kernel
        ops.linear(torch.zeros(2, 4), torch.zeros(3, 4))
    # a single token / single key attention is the identity on V
    qkv = synth.gaussian("deg_qkv", (2, 1, 3 * 512)).to(d)
    report("mha T=1", ops.mha(qkv, None, 8), qkv[..., 1024:].cpu(), 1e-6)
