"""Tensor-level wrappers of the primitive C-ABI entry points (include/afm_hip.h).

Each function mirrors the PyTorch call the reference makes at that point of the path
(F.linear, F.layer_norm, the attention inside nn.TransformerEncoderLayer, the DDPM update),
takes/returns torch tensors on the GPU and enqueues the HIP kernel on the current stream.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import ffi

RowMap = Tuple[int, ...]   # (group, stride, offset[, skip_after, skip]): row r -> (r // group) * stride + offset + j + (skip if j >= skip_after), j = r % group


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, *, act: int = ffi.ACT_NONE,
           act_post: int = ffi.ACT_NONE,
           scale: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
           rowtab: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
           rows: Optional[int] = None, a_map: Optional[RowMap] = None, c_map: Optional[RowMap] = None,
           rowdot_w: Optional[torch.Tensor] = None, rowdot_out: Optional[torch.Tensor] = None, store: bool = True,
           ln: Optional[Tuple[torch.Tensor, torch.Tensor, float]] = None, ln_out: Optional[torch.Tensor] = None,
           ln_counters: Optional[torch.Tensor] = None, stat_out: Optional[torch.Tensor] = None,
           a_stat: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
           res_stat: Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = None, ln_eps: float = 1e-5, defer: bool = False) -> torch.Tensor:
    """``act_post(act(scale * (x @ weight.T) + bias) + residual + rowtab[row % len(rowtab)])`` on f32 MFMA.

    x [..., K] (or a 2-D row pool when ``a_map`` gathers rows), weight [N, K] as in nn.Linear.
    With ``out`` given (2-D row pool [R, N]) and ``c_map``, rows are scattered into it.
    ``rowdot_w`` [R, N] + ``rowdot_out`` [rows, ceil(N / 64), R]: the fused row-dot epilogue (afm_linear_args.rowdot_*): per 64-column
    group, the dot of the output row with each of the R vectors; ``store=False`` then skips writing the output itself.
    ``ln`` = (gamma, beta, eps) + ``ln_out`` (same row layout as the output): the fused LayerNorm of the output rows (afm_linear_args.ln_*,
    the workgroup finishing the last column tile of a row block normalises it); ``ln_counters`` = zeroed int32 scratch of >= ceil(M / 32)
    words (allocated here when omitted).  Returns the (pre-LayerNorm) output; the normalised rows are in ``ln_out``.
    LayerNorm folded ACROSS launches (afm_linear_args.stat_out / a_stat / res_stat): ``stat_out`` [rows, N / 64, 2] receives (mean, M2) per
    output row and 64-column group; ``a_stat`` = (statistics of the raw input rows, g [N]) with ``weight`` = W * gamma and ``bias`` =
    b + W beta makes this call compute W LN(x) + b from the RAW x; ``res_stat`` = (statistics, gamma, beta) adds LayerNorm(residual)."""
    lib = ffi.load()
    ffi.require_gpu(x, weight)
    x = ffi.f32c(x)
    weight = ffi.f32c(weight)
    K = x.shape[-1]
    N = weight.shape[0]
    assert weight.shape[1] == K, (weight.shape, x.shape)
    M = rows if rows is not None else x.numel() // K
    if out is None:
        out = torch.empty(*(x.shape[:-1] if rows is None else (M,)), N, device=x.device, dtype=torch.float32)
    a = ffi.LinearArgs()
    a.A, a.lda, a.W, a.ldw, a.C, a.ldc = x.data_ptr(), x.stride(-2) if x.dim() > 1 else K, weight.data_ptr(), K, out.data_ptr(), N
    a.M, a.N, a.K = M, N, K
    keep = [x, weight, out]
    for name, t in (("bias", bias), ("scale", scale)):
        if t is not None:
            t = ffi.f32c(t)
            assert t.numel() == N
            keep.append(t)
            setattr(a, name, t.data_ptr())
    if residual is not None:
        residual = ffi.f32c(residual)
        keep.append(residual)
        a.residual, a.ldr = residual.data_ptr(), N
    if rowtab is not None:
        rowtab = ffi.f32c(rowtab)
        keep.append(rowtab)
        a.rowtab, a.rowtab_period = rowtab.data_ptr(), rowtab.shape[0]
    a.act, a.act_post = act, act_post
    if a_map:
        a.a_grp, a.a_stride, a.a_off = a_map[:3]
        if len(a_map) == 5:                       # (group, stride, offset, skip_after, skip): a hole of `skip` rows behind member skip_after - 1
            a.a_skip_after, a.a_skip = a_map[3:]
    if c_map:
        a.c_grp, a.c_stride, a.c_off = c_map[:3]
        if len(c_map) == 5:
            a.c_skip_after, a.c_skip = c_map[3:]
    if rowdot_w is not None:
        rowdot_w = ffi.f32c(rowdot_w)
        R = rowdot_w.shape[0]
        assert rowdot_w.shape[1] == N and 1 <= R <= 8, rowdot_w.shape
        assert rowdot_out is not None and rowdot_out.is_contiguous() and rowdot_out.dtype == torch.float32 and rowdot_out.device == x.device
        assert rowdot_out.numel() >= M * ((N + 63) // 64) * R, (rowdot_out.shape, M, N, R)      # [rows, ceil(N / 64), R]: an undersized buffer is written out of bounds
        keep += [rowdot_w, rowdot_out]
        a.rowdot_w, a.rowdot_out, a.rowdot_n = rowdot_w.data_ptr(), rowdot_out.data_ptr(), rowdot_w.shape[0]
        if not store:
            a.C = None
    if ln is not None:
        g, b, eps = ln
        g, b = ffi.f32c(g), ffi.f32c(b)
        assert ln_out is not None and ln_out.dtype == torch.float32 and ln_out.is_contiguous() and ln_out.shape[-1] == N and g.numel() == N and b.numel() == N
        if ln_counters is None:
            ln_counters = torch.zeros((M + 31) // 32, dtype=torch.int32, device=x.device)
        assert ln_counters.dtype == torch.int32 and ln_counters.numel() >= (M + 31) // 32
        keep += [g, b, ln_out, ln_counters]
        a.ln_gamma, a.ln_beta, a.ln_out, a.ldo, a.ln_eps, a.ln_counters = g.data_ptr(), b.data_ptr(), ln_out.data_ptr(), N, float(eps), ln_counters.data_ptr()
    if stat_out is not None:
        assert stat_out.dtype == torch.float32 and stat_out.is_contiguous() and stat_out.numel() >= out.numel() // N * (N // 64) * 2 and N % 64 == 0
        keep.append(stat_out)
        a.stat_out = stat_out.data_ptr()
    if a_stat is not None:
        st, g = a_stat
        st, g = ffi.f32c(st), ffi.f32c(g)
        assert g.numel() == N and K % 64 == 0
        keep += [st, g]
        a.a_stat, a.a_stat_groups, a.a_fold_g = st.data_ptr(), K // 64, g.data_ptr()
    if res_stat is not None:
        st, g, b = res_stat
        st, g, b = ffi.f32c(st), ffi.f32c(g), ffi.f32c(b)
        assert residual is not None and g.numel() == N and b.numel() == N
        keep += [st, g, b]
        a.res_stat, a.res_gamma, a.res_beta = st.data_ptr(), g.data_ptr(), b.data_ptr()
    if stat_out is not None or a_stat is not None or res_stat is not None:
        a.ln_eps2 = float(ln_eps)
        # the folded-LayerNorm epilogue reads its side inputs 16 bytes at a time and writes stat_out only from that branch: a
        # contiguous-but-offset view (bias[1:]) is refused here as afm_linear refuses it (AFM_E_BADARG), never silently mis-normalised
        for nm in ("C", "residual", "bias", "a_fold_g", "res_gamma", "res_beta"):
            ptr = getattr(a, nm)
            if ptr and ptr % 16:
                raise ffi.AfmError(f"afm_linear with folded LayerNorm statistics: `{nm}` is not 16-byte aligned")
    fill_arith(a)
    if defer:                                           # linear_pair: the argument block instead of the launch
        return a, out, keep
    ffi.check(lib.afm_linear(C.byref(a), ffi.stream_of(x)), "afm_linear")
    return out


def linear_pair(first: dict, second: dict) -> Tuple[torch.Tensor, torch.Tensor]:
    """Two independent `linear` calls (keyword dicts of `linear`) as ONE launch of 128 x 128 tiles (afm_linear_pair): bit-identical to the
    two calls; raises AfmError (AFM_E_UNSUPPORTED) when the pair is not eligible (both on the bf16-split path, one arithmetic, K > 256)."""
    a0, out0, keep0 = linear(**first, defer=True)
    a1, out1, keep1 = linear(**second, defer=True)
    ffi.check(ffi.load().afm_linear_pair(C.byref(a0), C.byref(a1), ffi.stream_of(out0)), "afm_linear_pair")
    del keep0, keep1
    return out0, out1


# ---- arithmetic of the GEMMs: HOST state (this module), written into every afm_linear_args / weight pack that is built.
# The library itself has no switch and reads no environment (ABI v3).  Two settings:
#   sampling / inference operators (`linear`, `mha`, the CMDM / CDM weight packs): AFM_GEMM_SPLIT / AFM_GEMM_SPLIT_MIN_N in the environment of the
#     Python process, else SIX products of the three-way bf16 operand split on every eligible GEMM and in the attention (round 6: the
#     worst-case comparison of tests/test_gpu_arith.py - six products are never above nine and both sit in the error class of an f32 chain);
#   training operators (afm.autograd: every forward and backward GEMM of the tape): AFM_GEMM_SPLIT_TRAIN, else all NINE products (exact f32
#     products) - gradients through ~40 serial batch-statistics BatchNorms are ill-conditioned in float32 (the reference's own f32 backward sits
#     2.8e-2 from its f64 backward on the CDM's PointTrans U-Net), so the tape keeps every bit the operands carry.
DEFAULT_PRODUCTS, DEFAULT_TRAIN_PRODUCTS = 6, 9


def _initial_split():
    import os
    p = os.environ.get("AFM_GEMM_SPLIT")
    products = int(p) if p is not None else DEFAULT_PRODUCTS
    if products not in (0, 1, 6, 9):
        products = 0
    n = os.environ.get("AFM_GEMM_SPLIT_MIN_N")
    return products, max(0, int(n)) if n is not None else 0


def _initial_train_split():
    import os
    p = os.environ.get("AFM_GEMM_SPLIT_TRAIN")
    products = int(p) if p is not None else DEFAULT_TRAIN_PRODUCTS
    return products if products in (0, 6, 9) else 0


_gemm_split = list(_initial_split())
_train_split = [_initial_train_split()]
_gemm_tune = 0
_ARITH = {9: ffi.ARITH_BF16X9, 6: ffi.ARITH_BF16X6, 1: ffi.ARITH_BF16X1}


def gemm_arith() -> Tuple[int, int]:
    """(arith, arith_min_n) for afm_linear_args / afm_c(m)dm_weights from the host setting."""
    products, min_n = _gemm_split
    if products == 0:
        return ffi.ARITH_F32, 0
    return _ARITH[products], min_n


def set_train_gemm_split(products: int) -> int:
    """Arithmetic of the TRAINING operators' GEMMs (afm.autograd): 9 (default, exact f32 products), 6 or 0 (native f32 MFMA).  Returns the previous value."""
    if int(products) not in (0, 6, 9):
        raise ffi.AfmError(f"set_train_gemm_split: products must be 0, 6 or 9 (got {products})")
    prev, _train_split[0] = _train_split[0], int(products)
    return prev


def get_train_gemm_split() -> int:
    return _train_split[0]


def fill_arith_train(a) -> None:
    """afm_linear_args of a training operator: the tape's arithmetic (see the note above), never the sampling setting."""
    a.arith, a.arith_min_n = (ffi.ARITH_F32, 0) if _train_split[0] == 0 else (_ARITH[_train_split[0]], 0)
    a.tune = _gemm_tune


def set_gemm_split(products: int, min_n: Optional[int] = None):
    """Arithmetic of `linear`'s GEMMs and of `mha`: the three-way bf16 operand split on the bf16 matrix pipe with the 6 largest cross products
    (default, every eligible GEMM: K >= 128, K % 16 == 0, aligned) or all 9 (exact f32 products), 0 = native f32 MFMA everywhere.  ``min_n`` moves the N threshold.
    Returns the previous setting in the same form (products, or (products, min_n) when ``min_n`` was given)."""
    if int(products) not in (0, 1, 6, 9):
        raise ffi.AfmError(f"set_gemm_split: products must be 0, 6 or 9 (or 1: plain bf16, informational only) (got {products})")
    if min_n is not None and int(min_n) < 0:
        raise ffi.AfmError(f"set_gemm_split: min_n must be >= 0 (got {min_n})")
    prev = tuple(_gemm_split)
    _gemm_split[0] = int(products)
    if min_n is None:
        return prev[0]
    _gemm_split[1] = int(min_n)
    return prev


def get_gemm_split():
    """Current (products, min_n) of the GEMM arithmetic."""
    return tuple(_gemm_split)


def set_gemm_tune(tune: int) -> int:
    """Bit-neutral performance knobs of afm_linear for experiments (afm_linear_args.tune, AFM_TUNE_*); returns the previous value."""
    global _gemm_tune
    prev, _gemm_tune = _gemm_tune, int(tune)
    return prev


def fill_arith(a) -> None:
    a.arith, a.arith_min_n = gemm_arith()
    a.tune = _gemm_tune


def mha(qkv: torch.Tensor, key_mask: Optional[torch.Tensor], heads: int, group_waves: int = 0, q_first: int = 0) -> torch.Tensor:
    """qkv [B, T, 3*d] (packed in_proj output) -> softmax(QK^T/sqrt(dh) + mask) V, [B, T, d].  ``group_waves``: workgroup shape of
    afm_mha_fwd_grouped (0 = the library's choice); results do not depend on it.  ``q_first`` > 0 (afm_mha_fwd_rows): only the query
    rows q_first .. T - 1 are computed, the leading rows of the result are zeros."""
    lib = ffi.load()
    ffi.require_gpu(qkv)
    qkv = ffi.f32c(qkv)
    B, T, d3 = qkv.shape
    d = d3 // 3
    km = None
    if key_mask is not None:
        km = key_mask.to(torch.uint8).contiguous()
        assert km.shape == (B, T)
    arith = gemm_arith()[0]                             # the attention follows the host's GEMM arithmetic (six products by default)
    if q_first:
        out = torch.zeros(B, T, d, device=qkv.device, dtype=torch.float32)
        ffi.check(lib.afm_mha_fwd_arith(qkv.data_ptr(), ffi.ptr(km), out.data_ptr(), B, T, heads, d // heads, int(q_first), int(group_waves), arith,
                                        ffi.stream_of(qkv)), "afm_mha_fwd_arith")
        return out
    out = torch.empty(B, T, d, device=qkv.device, dtype=torch.float32)
    ffi.check(lib.afm_mha_fwd_arith(qkv.data_ptr(), ffi.ptr(km), out.data_ptr(), B, T, heads, d // heads, 0, int(group_waves), arith,
                                    ffi.stream_of(qkv)), "afm_mha_fwd_arith")
    return out


def clamp_(x: torch.Tensor, lo: float, hi: float) -> torch.Tensor:
    """In-place clamp of a float32 GPU tensor (afm_clamp): `process_xstart` with clip_denoised=True (gaussian_diffusion.py:289-294)."""
    ffi.require_gpu(x)
    assert x.dtype == torch.float32 and x.is_contiguous()
    ffi.check(ffi.load().afm_clamp(x.data_ptr(), x.numel(), float(lo), float(hi), ffi.stream_of(x)), "afm_clamp")
    return x


def mha_cross(q: torch.Tensor, kv: torch.Tensor, key_mask: Optional[torch.Tensor], heads: int) -> torch.Tensor:
    """q [B, Tq, d], kv [B, Tk, 2*d] (packed k | v), key_mask [B, Tk] (True = ignore) -> [B, Tq, d]."""
    lib = ffi.load()
    ffi.require_gpu(q, kv)
    q, kv = ffi.f32c(q), ffi.f32c(kv)
    B, Tq, d = q.shape
    Tk = kv.shape[1]
    assert kv.shape[2] == 2 * d
    out = torch.empty(B, Tq, d, device=q.device, dtype=torch.float32)
    km = None
    if key_mask is not None:
        km = key_mask.to(torch.uint8).contiguous()
        assert km.shape == (B, Tk)
    ffi.check(lib.afm_mha_cross_fwd(q.data_ptr(), kv.data_ptr(), ffi.ptr(km), out.data_ptr(), B, Tq, Tk, heads, d // heads, ffi.stream_of(q)),
              "afm_mha_cross_fwd")
    return out


def layernorm(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float = 1e-5,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = ffi.load()
    ffi.require_gpu(x)
    x = ffi.f32c(x)
    out = torch.empty_like(x) if out is None else out
    dim = x.shape[-1]
    w, b = ffi.f32c(weight), ffi.f32c(bias)
    ffi.check(lib.afm_layernorm(x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), x.numel() // dim, dim, eps,
                                ffi.stream_of(x)), "afm_layernorm")
    return out


def ddpm_step(x0: torch.Tensor, x_t: torch.Tensor, noise: Optional[torch.Tensor], c1: torch.Tensor, c2: torch.Tensor,
              sigma: torch.Tensor, *, seed: int = 0, sample_index0: int = 0, step: int = 0,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x_{t-1} = (c1*x0 + c2*x_t) + sigma*noise, per-sample coefficients [B] (no fma contraction)."""
    lib = ffi.load()
    ffi.require_gpu(x0, x_t)
    x0, x_t = ffi.f32c(x0), ffi.f32c(x_t)
    B = x0.shape[0]
    per = x0.numel() // max(B, 1)
    out = torch.empty_like(x0) if out is None else out
    nz = None if noise is None else ffi.f32c(noise)
    c1, c2, sigma = ffi.f32c(c1), ffi.f32c(c2), ffi.f32c(sigma)
    ffi.check(lib.afm_ddpm_step(x0.data_ptr(), x_t.data_ptr(), ffi.ptr(nz), out.data_ptr(), c1.data_ptr(), c2.data_ptr(),
                                sigma.data_ptr(), B, per, seed & (2**64 - 1), sample_index0, step, ffi.stream_of(x0)),
              "afm_ddpm_step")
    return out


def randn(shape, device, *, seed: int, sample_index0: int = 0, step: int = 0) -> torch.Tensor:
    """Counter-based N(0,1) noise keyed by (seed, global sample index, step, element)."""
    lib = ffi.load()
    out = torch.empty(*shape, device=device, dtype=torch.float32)
    ffi.require_gpu(out)
    B = shape[0]
    ffi.check(lib.afm_randn(out.data_ptr(), B, out.numel() // max(B, 1), seed & (2**64 - 1), sample_index0, step,
                            ffi.stream_of(out)), "afm_randn")
    return out


def masked_mse(target: torch.Tensor, pred: torch.Tensor, frame_mask: Optional[torch.Tensor]) -> torch.Tensor:
    """Per-sample MSE over the un-padded frames: [B, L, D] x2 (+ [B, L] bool, True = padded) -> [B]."""
    lib = ffi.load()
    ffi.require_gpu(target, pred)
    target, pred = ffi.f32c(target), ffi.f32c(pred)
    B, D = target.shape[0], target.shape[-1]
    L = target.numel() // max(B * D, 1)
    km = None if frame_mask is None else frame_mask.reshape(B, L).to(torch.uint8).contiguous()
    out = torch.empty(B, device=target.device, dtype=torch.float32)
    ffi.check(lib.afm_masked_mse(target.data_ptr(), pred.data_ptr(), ffi.ptr(km), out.data_ptr(), B, L, D,
                                 ffi.stream_of(target)), "afm_masked_mse")
    return out
