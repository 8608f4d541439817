"""Training path: torch.autograd Functions whose forward AND backward are the HIP kernels of libafm_hip.so.

torch.autograd is only the tape (it orders the backward calls and accumulates ``.grad`` so that the reference's
``loss.backward(); optimizer.step()`` of utils/training.py:140-155 works unchanged); every tensor operation on the
path is a C-ABI call.  Two granularities:

* ``encoder_layer`` - one post-LN nn.TransformerEncoderLayer (cmdm.py:66-77, train mode incl. its four dropouts) as a
  single Function with a hand-scheduled backward: LayerNorm backward emits the dropout-masked branch gradient,
  the input-gradient GEMMs fuse GELU' / dropout / residual-add in their epilogues, weight gradients run on the
  reduction-major MFMA kernel, attention backward is flash-style from the saved log-sum-exp.
* ``linear`` / ``posenc_dropout`` / ``masked_mse`` - the small stand-alone pieces around the layers (adapters,
  TimestepEmbedder MLP, PositionalEncoding, output projection, loss).

Dropout masks are never stored: forward and backward regenerate them from (seed, mask id, row, col).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import ffi, ops

RowMap = Tuple[int, int, int]


def _st(t: torch.Tensor) -> int:
    return ffi.stream_of(t)


def _gemm(A, W, out, M, N, K, *, lda=None, bias=None, residual=None, act=0, preact=None, dact=0, dact_z=None, drop=None,
          drop_after=0, a_map=None, c_map=None, rowtab=None):
    """Raw afm_linear call: out[M,N] = epilogue(A[M,K] @ W[N,K]^T)."""
    a = ffi.LinearArgs()
    a.A, a.lda, a.W, a.ldw, a.C, a.ldc = A.data_ptr(), (K if lda is None else lda), W.data_ptr(), K, out.data_ptr(), N
    a.M, a.N, a.K = M, N, K
    if bias is not None:
        a.bias = bias.data_ptr()
    if residual is not None:
        a.residual, a.ldr = residual.data_ptr(), N
    if rowtab is not None:
        a.rowtab, a.rowtab_period = rowtab.data_ptr(), rowtab.shape[0]
    a.act = act
    if preact is not None:
        a.preact, a.ldp = preact.data_ptr(), N
    if dact:
        a.dact, a.dact_z, a.ldz = dact, dact_z.data_ptr(), N
    if drop is not None and drop[0] > 0.0:
        a.drop_p, a.drop_seed, a.drop_id, a.drop_after = drop[0], drop[1], drop[2], drop_after
    if a_map:
        a.a_grp, a.a_stride, a.a_off = a_map
    if c_map:
        a.c_grp, a.c_stride, a.c_off = c_map
    ops.fill_arith_train(a)             # the tape's arithmetic: all nine products unless the host says otherwise (afm.ops.set_train_gemm_split)
    ffi.check(ffi.load().afm_linear(C.byref(a), _st(out)), "afm_linear")
    return out


def _transpose(w: torch.Tensor) -> torch.Tensor:
    """[N,K] -> [K,N] on the device (operand of the input-gradient GEMM dX = dY @ W)."""
    out = torch.empty(w.shape[1], w.shape[0], device=w.device, dtype=torch.float32)
    ffi.check(ffi.load().afm_transpose(w.data_ptr(), out.data_ptr(), w.shape[0], w.shape[1], _st(w)), "afm_transpose")
    return out


def _wgrad(dY, X, M, N, K, *, want_bias=True, dy_map: Optional[RowMap] = None, x_map: Optional[RowMap] = None, lddy=None, ldx=None):
    lib = ffi.load()
    dW = torch.empty(N, K, device=dY.device, dtype=torch.float32)
    db = torch.empty(N, device=dY.device, dtype=torch.float32) if want_bias else None
    nbytes = lib.afm_linear_wgrad_workspace_bytes(M, N, K)
    if nbytes < 0:
        ffi.check(int(nbytes), "afm_linear_wgrad_workspace_bytes")
    ws = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=dY.device)
    a = ffi.WgradArgs()
    a.dY, a.lddy, a.X, a.ldx, a.dW, a.lddw = dY.data_ptr(), (N if lddy is None else lddy), X.data_ptr(), (K if ldx is None else ldx), dW.data_ptr(), K
    a.db = ffi.ptr(db)
    a.M, a.N, a.K = M, N, K
    if dy_map:
        a.dy_grp, a.dy_stride, a.dy_off = dy_map
    if x_map:
        a.x_grp, a.x_stride, a.x_off = x_map
    a.ws, a.ws_bytes = ws.data_ptr(), ws.numel()
    ffi.check(lib.afm_linear_wgrad(C.byref(a), _st(dY)), "afm_linear_wgrad")
    return dW, db


def _layernorm(x, g, b, eps=1e-5):
    out = torch.empty_like(x)
    dim = x.shape[-1]
    ffi.check(ffi.load().afm_layernorm(x.data_ptr(), g.data_ptr(), b.data_ptr(), out.data_ptr(), x.numel() // dim, dim, eps, _st(x)),
              "afm_layernorm")
    return out


def _layernorm_bwd(x, g, dy, drop=None, eps=1e-5):
    """-> (dx, dx_dropmasked or dx, dgamma, dbeta)."""
    lib = ffi.load()
    dim = x.shape[-1]
    rows = x.numel() // dim
    dx = torch.empty_like(x)
    use_drop = drop is not None and drop[0] > 0.0
    dxd = torch.empty_like(x) if use_drop else None
    dg = torch.empty(dim, device=x.device, dtype=torch.float32)
    db = torch.empty(dim, device=x.device, dtype=torch.float32)
    nbytes = lib.afm_layernorm_bwd_workspace_bytes(rows, dim)
    ws = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=x.device)
    p, seed, did = drop if use_drop else (0.0, 0, 0)
    ffi.check(lib.afm_layernorm_bwd(x.data_ptr(), g.data_ptr(), dy.data_ptr(), dx.data_ptr(), ffi.ptr(dxd), dg.data_ptr(), db.data_ptr(),
                                    rows, dim, eps, p, seed, did, ws.data_ptr(), ws.numel(), _st(x)), "afm_layernorm_bwd")
    return dx, (dxd if use_drop else dx), dg, db


def _rowop(x, *, rowtab=None, z=None, act=0, drop=None, out=None):
    cols = x.shape[-1]
    rows = x.numel() // cols
    out = torch.empty_like(x) if out is None else out
    p, seed, did = drop if (drop is not None and drop[0] > 0.0) else (0.0, 0, 0)
    ffi.check(ffi.load().afm_rowop(x.data_ptr(), ffi.ptr(rowtab), 0 if rowtab is None else rowtab.shape[0], ffi.ptr(z), act, out.data_ptr(),
                                   rows, cols, p, seed, did, _st(x)), "afm_rowop")
    return out


def _c(t: torch.Tensor) -> torch.Tensor:
    ffi.require_gpu(t)
    return ffi.f32c(t.detach())


# ------------------------------------------------------------------------------------------------ stand-alone linear
class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, act, a_map, rows, residual):
        xc, w = _c(x), _c(weight)
        b = None if bias is None else _c(bias)
        K, N = w.shape[1], w.shape[0]
        M = rows if rows is not None else xc.numel() // K
        out = torch.empty(M, N, device=xc.device, dtype=torch.float32)
        z = torch.empty_like(out) if act else None
        res = None if residual is None else _c(residual).view(M, N)
        _gemm(xc.view(-1, K), w, out, M, N, K, bias=b, act=act, preact=z, a_map=a_map, residual=res)
        ctx.save_for_backward(xc, w, z)
        ctx.cfg = (act, a_map, M, N, K, tuple(x.shape), bias is not None, None if residual is None else tuple(residual.shape))
        return out if rows is not None or a_map else out.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        xc, w, z = ctx.saved_tensors
        act, a_map, M, N, K, xshape, has_bias, res_shape = ctx.cfg
        dy = _c(dy).view(M, N)
        dres = dy.view(res_shape) if res_shape is not None and ctx.needs_input_grad[6] else None     # identity path
        if act:
            dy = _rowop(dy, z=z, act=act)
        dx = dw = db = None
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            dw, db = _wgrad(dy, xc.view(-1, K), M, N, K, want_bias=has_bias, x_map=a_map)
        if ctx.needs_input_grad[0]:
            if a_map:
                dx = torch.zeros(xc.numel() // K, K, device=dy.device, dtype=torch.float32)   # rows outside the gather get 0
            else:
                dx = torch.empty(M, K, device=dy.device, dtype=torch.float32)
            _gemm(dy, _transpose(w), dx, M, K, N, c_map=a_map)
            dx = dx.view(xshape)
        return dx, dw, db, None, None, None, dres


def linear(x, weight, bias=None, *, act: int = ffi.ACT_NONE, a_map: Optional[RowMap] = None, rows: Optional[int] = None,
           residual: Optional[torch.Tensor] = None):
    """act(x @ weight.T + bias) + residual with a HIP backward.  ``a_map`` + ``rows`` gather a strided subset of x's rows
    (the motion tokens of every sample, cmdm.py:169)."""
    return _LinearFn.apply(x, weight, bias, act, a_map, rows, residual)


# ------------------------------------------------------------------------------------------------ PositionalEncoding
class _PosEncDropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pe, drop):
        xc = _c(x)
        ctx.drop = drop
        return _rowop(xc, rowtab=_c(pe), drop=drop)

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        return (_rowop(dy, drop=ctx.drop) if ctx.drop[0] > 0.0 else dy), None, None


def posenc_dropout(x, pe, drop):
    """dropout(x + pe[:T]) over [B, T, d] (modules.py:43-45); pe [T, d]."""
    return _PosEncDropoutFn.apply(x, pe, drop)


# ------------------------------------------------------------------------------------------------ encoder layer
class _EncoderLayerFn(torch.autograd.Function):
    """x [B, T, d] -> post-LN TransformerEncoderLayer(x) with key padding mask; drops = (p, seed, id0): ids id0..id0+3 are
    the attention-probability, dropout1, FFN and dropout2 masks."""

    @staticmethod
    def forward(ctx, x, key_mask, heads, drops, in_w, in_b, out_w, out_b, l1_w, l1_b, l2_w, l2_b, n1_w, n1_b, n2_w, n2_b, act=ffi.ACT_GELU):
        lib = ffi.load()
        xc = _c(x)
        B, T, d = xc.shape
        M, ff = B * T, l1_w.shape[0]
        P = [_c(t) for t in (in_w, in_b, out_w, out_b, l1_w, l1_b, l2_w, l2_b, n1_w, n1_b, n2_w, n2_b)]
        in_w, in_b, out_w, out_b, l1_w, l1_b, l2_w, l2_b, n1_w, n1_b, n2_w, n2_b = P
        p, seed, id0 = drops
        dev = xc.device
        km = None if key_mask is None else key_mask.to(torch.uint8).contiguous()
        qkv = torch.empty(M, 3 * d, device=dev, dtype=torch.float32)
        _gemm(xc, in_w, qkv, M, 3 * d, d, bias=in_b)
        att = torch.empty(M, d, device=dev, dtype=torch.float32)
        lse = torch.empty(B * heads * T, device=dev, dtype=torch.float32)
        ffi.check(lib.afm_mha_fwd_train(qkv.data_ptr(), ffi.ptr(km), att.data_ptr(), lse.data_ptr(), B, T, heads, d // heads, p, seed, id0,
                                        _st(xc)), "afm_mha_fwd_train")
        s1 = torch.empty(M, d, device=dev, dtype=torch.float32)
        _gemm(att, out_w, s1, M, d, d, bias=out_b, residual=xc, drop=(p, seed, id0 + 1))
        x1 = _layernorm(s1, n1_w, n1_b)
        z = torch.empty(M, ff, device=dev, dtype=torch.float32)
        h = torch.empty(M, ff, device=dev, dtype=torch.float32)
        _gemm(x1, l1_w, h, M, ff, d, bias=l1_b, act=act, preact=z, drop=(p, seed, id0 + 2))
        s2 = torch.empty(M, d, device=dev, dtype=torch.float32)
        _gemm(h, l2_w, s2, M, d, ff, bias=l2_b, residual=x1, drop=(p, seed, id0 + 3))
        x2 = _layernorm(s2, n2_w, n2_b)
        ctx.save_for_backward(xc, km, qkv, att, lse, s1, x1, z, h, s2, *P)
        ctx.cfg = (B, T, d, ff, heads, drops, act)
        return x2.view(B, T, d)

    @staticmethod
    def backward(ctx, dx2):
        lib = ffi.load()
        xc, km, qkv, att, lse, s1, x1, z, h, s2, in_w, in_b, out_w, out_b, l1_w, l1_b, l2_w, l2_b, n1_w, n1_b, n2_w, n2_b = ctx.saved_tensors
        B, T, d, ff, heads, (p, seed, id0), act = ctx.cfg
        M = B * T
        dev = xc.device
        dx2 = _c(dx2).view(M, d)
        # ---- FFN half:  x2 = LN2(s2), s2 = x1 + drop3(h W2^T + b2), h = drop2(gelu(z)), z = x1 W1^T + b1
        ds2, ds2d, dn2w, dn2b = _layernorm_bwd(s2, n2_w, dx2, drop=(p, seed, id0 + 3))
        dl2w, dl2b = _wgrad(ds2d, h, M, d, ff)
        dz = torch.empty(M, ff, device=dev, dtype=torch.float32)
        _gemm(ds2d, _transpose(l2_w), dz, M, ff, d, drop=(p, seed, id0 + 2), dact=act, dact_z=z)               # dh o mask o act'(z)
        dl1w, dl1b = _wgrad(dz, x1, M, ff, d)
        dx1 = torch.empty(M, d, device=dev, dtype=torch.float32)
        _gemm(dz, _transpose(l1_w), dx1, M, d, ff, residual=ds2)                                           # + residual path of s2
        # ---- attention half:  x1 = LN1(s1), s1 = x + drop1(att Wo^T + bo), att = MHA(qkv), qkv = x Win^T + bin
        ds1, ds1d, dn1w, dn1b = _layernorm_bwd(s1, n1_w, dx1, drop=(p, seed, id0 + 1))
        dow, dob = _wgrad(ds1d, att, M, d, d)
        datt = torch.empty(M, d, device=dev, dtype=torch.float32)
        _gemm(ds1d, _transpose(out_w), datt, M, d, d)
        dqkv = torch.empty(M, 3 * d, device=dev, dtype=torch.float32)
        ws = torch.empty(B * heads * T, device=dev, dtype=torch.float32)
        ffi.check(lib.afm_mha_bwd(qkv.data_ptr(), ffi.ptr(km), att.data_ptr(), datt.data_ptr(), lse.data_ptr(), dqkv.data_ptr(), B, T, heads,
                                  d // heads, p, seed, id0, ws.data_ptr(), ws.numel() * 4, _st(xc)), "afm_mha_bwd")
        diw, dib = _wgrad(dqkv, xc.view(M, d), M, 3 * d, d)
        dx = torch.empty(M, d, device=dev, dtype=torch.float32)
        _gemm(dqkv, _transpose(in_w), dx, M, d, 3 * d, residual=ds1)
        return (dx.view(B, T, d), None, None, None, diw, dib, dow, dob, dl1w, dl1b, dl2w, dl2b, dn1w, dn1b, dn2w, dn2b, None)


def encoder_layer(x, layer: torch.nn.TransformerEncoderLayer, key_mask, heads: int, drops, act: int = ffi.ACT_GELU):
    """One nn.TransformerEncoderLayer (post-LN, GELU or ReLU, batch_first) forward with the HIP backward attached."""
    return _EncoderLayerFn.apply(x, key_mask, heads, drops, layer.self_attn.in_proj_weight, layer.self_attn.in_proj_bias,
                                 layer.self_attn.out_proj.weight, layer.self_attn.out_proj.bias, layer.linear1.weight, layer.linear1.bias,
                                 layer.linear2.weight, layer.linear2.bias, layer.norm1.weight, layer.norm1.bias, layer.norm2.weight,
                                 layer.norm2.bias, act)


class _DecoderLayerFn(torch.autograd.Function):
    """x [B, T, d], mem [B, n, d] -> post-LN nn.TransformerDecoderLayer(x, mem) (GELU): self-attention (key padding mask), cross-attention
    over the memory (its K | V projections are part of the layer: in_proj rows d .. 3d), feed-forward - the CMDM's `trans_dec` variant
    (cmdm.py:78-113,171-191).  drops = (p, seed, id0): ids id0 .. id0 + 5 are the self-attention probabilities, dropout1, the
    cross-attention probabilities, dropout2, the FFN dropout and dropout3."""

    @staticmethod
    def forward(ctx, x, mem, key_mask, mem_mask, heads, drops, sa_in_w, sa_in_b, sa_out_w, sa_out_b, ca_in_w, ca_in_b, ca_out_w, ca_out_b,
                l1_w, l1_b, l2_w, l2_b, n1_w, n1_b, n2_w, n2_b, n3_w, n3_b):
        lib = ffi.load()
        xc, mc = _c(x), _c(mem)
        B, T, d = xc.shape
        n = mc.shape[1]
        M, Mm, ff = B * T, B * n, l1_w.shape[0]
        P = [_c(t) for t in (sa_in_w, sa_in_b, sa_out_w, sa_out_b, ca_in_w, ca_in_b, ca_out_w, ca_out_b, l1_w, l1_b, l2_w, l2_b, n1_w, n1_b,
                             n2_w, n2_b, n3_w, n3_b)]
        sa_in_w, sa_in_b, sa_out_w, sa_out_b, ca_in_w, ca_in_b, ca_out_w, ca_out_b, l1_w, l1_b, l2_w, l2_b, n1_w, n1_b, n2_w, n2_b, n3_w, n3_b = P
        p, seed, id0 = drops
        dev = xc.device
        km = None if key_mask is None else key_mask.to(torch.uint8).contiguous()
        mm = None if mem_mask is None else mem_mask.to(torch.uint8).contiguous()
        E = lambda r, c: torch.empty(r, c, device=dev, dtype=torch.float32)
        # self-attention half
        qkv = E(M, 3 * d)
        _gemm(xc, sa_in_w, qkv, M, 3 * d, d, bias=sa_in_b)
        att, lse1 = E(M, d), torch.empty(B * heads * T, device=dev, dtype=torch.float32)
        ffi.check(lib.afm_mha_fwd_train(qkv.data_ptr(), ffi.ptr(km), att.data_ptr(), lse1.data_ptr(), B, T, heads, d // heads, p, seed, id0, _st(xc)),
                  "afm_mha_fwd_train")
        s1 = E(M, d)
        _gemm(att, sa_out_w, s1, M, d, d, bias=sa_out_b, residual=xc, drop=(p, seed, id0 + 1))
        x1 = _layernorm(s1, n1_w, n1_b)
        # cross-attention half: q from the tokens, k | v from the memory
        wq, bq, wkv, bkv = ca_in_w[:d], ca_in_b[:d], ca_in_w[d:], ca_in_b[d:]
        qc, kv = E(M, d), E(Mm, 2 * d)
        _gemm(x1, wq, qc, M, d, d, bias=bq)
        _gemm(mc, wkv, kv, Mm, 2 * d, d, bias=bkv)
        catt, lse2 = E(M, d), torch.empty(B * heads * T, device=dev, dtype=torch.float32)
        ffi.check(lib.afm_mha_cross_fwd_train(qc.data_ptr(), kv.data_ptr(), ffi.ptr(mm), catt.data_ptr(), lse2.data_ptr(), B, T, n, heads, d // heads,
                                              p, seed, id0 + 2, _st(xc)), "afm_mha_cross_fwd_train")
        s2 = E(M, d)
        _gemm(catt, ca_out_w, s2, M, d, d, bias=ca_out_b, residual=x1, drop=(p, seed, id0 + 3))
        x2 = _layernorm(s2, n2_w, n2_b)
        # feed-forward
        z, h = E(M, ff), E(M, ff)
        _gemm(x2, l1_w, h, M, ff, d, bias=l1_b, act=ffi.ACT_GELU, preact=z, drop=(p, seed, id0 + 4))
        s3 = E(M, d)
        _gemm(h, l2_w, s3, M, d, ff, bias=l2_b, residual=x2, drop=(p, seed, id0 + 5))
        x3 = _layernorm(s3, n3_w, n3_b)
        ctx.save_for_backward(xc, mc, km, mm, qkv, att, lse1, s1, x1, qc, kv, catt, lse2, s2, x2, z, h, s3, *P)
        ctx.cfg = (B, T, n, d, ff, heads, drops)
        return x3.view(B, T, d)

    @staticmethod
    def backward(ctx, dx3):
        lib = ffi.load()
        (xc, mc, km, mm, qkv, att, lse1, s1, x1, qc, kv, catt, lse2, s2, x2, z, h, s3, sa_in_w, sa_in_b, sa_out_w, sa_out_b, ca_in_w, ca_in_b,
         ca_out_w, ca_out_b, l1_w, l1_b, l2_w, l2_b, n1_w, n1_b, n2_w, n2_b, n3_w, n3_b) = ctx.saved_tensors
        B, T, n, d, ff, heads, (p, seed, id0) = ctx.cfg
        M, Mm = B * T, B * n
        dev = xc.device
        E = lambda r, c: torch.empty(r, c, device=dev, dtype=torch.float32)
        dx3 = _c(dx3).view(M, d)
        # ---- feed-forward:  x3 = LN3(s3), s3 = x2 + drop5(h W2^T + b2), h = drop4(gelu(z)), z = x2 W1^T + b1
        ds3, ds3d, dn3w, dn3b = _layernorm_bwd(s3, n3_w, dx3, drop=(p, seed, id0 + 5))
        dl2w, dl2b = _wgrad(ds3d, h, M, d, ff)
        dz = E(M, ff)
        _gemm(ds3d, _transpose(l2_w), dz, M, ff, d, drop=(p, seed, id0 + 4), dact=ffi.ACT_GELU, dact_z=z)
        dl1w, dl1b = _wgrad(dz, x2, M, ff, d)
        dx2 = E(M, d)
        _gemm(dz, _transpose(l1_w), dx2, M, d, ff, residual=ds3)
        # ---- cross-attention:  x2 = LN2(s2), s2 = x1 + drop3(catt Wo^T + bo), catt = MHA(qc, kv), qc = x1 Wq^T + bq, kv = mem Wkv^T + bkv
        ds2, ds2d, dn2w, dn2b = _layernorm_bwd(s2, n2_w, dx2, drop=(p, seed, id0 + 3))
        dcow, dcob = _wgrad(ds2d, catt, M, d, d)
        dcatt = E(M, d)
        _gemm(ds2d, _transpose(ca_out_w), dcatt, M, d, d)
        dqc, dkv = E(M, d), E(Mm, 2 * d)
        ws = torch.empty(B * heads * T, device=dev, dtype=torch.float32)
        ffi.check(lib.afm_mha_cross_bwd(qc.data_ptr(), kv.data_ptr(), ffi.ptr(mm), catt.data_ptr(), dcatt.data_ptr(), lse2.data_ptr(), dqc.data_ptr(),
                                        dkv.data_ptr(), B, T, n, heads, d // heads, p, seed, id0 + 2, ws.data_ptr(), ws.numel() * 4, _st(xc)),
                  "afm_mha_cross_bwd")
        wq, wkv = ca_in_w[:d], ca_in_w[d:]
        dwq, dbq = _wgrad(dqc, x1, M, d, d)
        dwkv, dbkv = _wgrad(dkv, mc.view(Mm, d), Mm, 2 * d, d)
        dmem = E(Mm, d)
        _gemm(dkv, _transpose(wkv), dmem, Mm, d, 2 * d)
        dx1 = E(M, d)
        _gemm(dqc, _transpose(wq), dx1, M, d, d, residual=ds2)
        # ---- self-attention:  x1 = LN1(s1), s1 = x + drop1(att Wo^T + bo), att = MHA(qkv), qkv = x Win^T + bin
        ds1, ds1d, dn1w, dn1b = _layernorm_bwd(s1, n1_w, dx1, drop=(p, seed, id0 + 1))
        dow, dob = _wgrad(ds1d, att, M, d, d)
        datt = E(M, d)
        _gemm(ds1d, _transpose(sa_out_w), datt, M, d, d)
        dqkv = E(M, 3 * d)
        ffi.check(lib.afm_mha_bwd(qkv.data_ptr(), ffi.ptr(km), att.data_ptr(), datt.data_ptr(), lse1.data_ptr(), dqkv.data_ptr(), B, T, heads,
                                  d // heads, p, seed, id0, ws.data_ptr(), ws.numel() * 4, _st(xc)), "afm_mha_bwd")
        diw, dib = _wgrad(dqkv, xc.view(M, d), M, 3 * d, d)
        dx = E(M, d)
        _gemm(dqkv, _transpose(sa_in_w), dx, M, d, 3 * d, residual=ds1)
        return (dx.view(B, T, d), dmem.view(B, n, d), None, None, None, None, diw, dib, dow, dob, torch.cat((dwq, dwkv), 0), torch.cat((dbq, dbkv), 0),
                dcow, dcob, dl1w, dl1b, dl2w, dl2b, dn1w, dn1b, dn2w, dn2b, dn3w, dn3b)


def decoder_layer(x, mem, layer: torch.nn.TransformerDecoderLayer, key_mask, mem_mask, heads: int, drops):
    """One nn.TransformerDecoderLayer (post-LN, GELU, batch_first) forward with the HIP backward attached."""
    sa, ca = layer.self_attn, layer.multihead_attn
    return _DecoderLayerFn.apply(x, mem, key_mask, mem_mask, heads, drops, sa.in_proj_weight, sa.in_proj_bias, sa.out_proj.weight, sa.out_proj.bias,
                                 ca.in_proj_weight, ca.in_proj_bias, ca.out_proj.weight, ca.out_proj.bias, layer.linear1.weight, layer.linear1.bias,
                                 layer.linear2.weight, layer.linear2.bias, layer.norm1.weight, layer.norm1.bias, layer.norm2.weight, layer.norm2.bias,
                                 layer.norm3.weight, layer.norm3.bias)


# ------------------------------------------------------------------------------------------------ LayerNorm / self-attention / Perceiver attention
class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        xc, g, b = _c(x), _c(weight), _c(bias)
        ctx.save_for_backward(xc, g)
        ctx.eps = eps
        return _layernorm(xc, g, b, eps)

    @staticmethod
    def backward(ctx, dy):
        xc, g = ctx.saved_tensors
        dx, _, dg, db = _layernorm_bwd(xc, g, _c(dy), eps=ctx.eps)
        return dx, dg, db, None


def layer_norm(x, ln: torch.nn.LayerNorm):
    """nn.LayerNorm over the last dimension with the HIP backward."""
    return _LayerNormFn.apply(x, ln.weight, ln.bias, float(ln.eps))


class _SelfAttentionFn(torch.autograd.Function):
    """Packed qkv [B, T, 3*C] -> softmax(QK^T/sqrt(dh)) V [B, T, C] (dh = 64), attention dropout by counter hash."""

    @staticmethod
    def forward(ctx, qkv, heads, drop):
        lib = ffi.load()
        q = _c(qkv)
        B, T, c3 = q.shape
        Cn = c3 // 3
        out = torch.empty(B, T, Cn, device=q.device, dtype=torch.float32)
        lse = torch.empty(B * heads * T, device=q.device, dtype=torch.float32)
        ffi.check(lib.afm_mha_fwd_train(q.data_ptr(), None, out.data_ptr(), lse.data_ptr(), B, T, heads, Cn // heads, drop[0], drop[1], drop[2],
                                        _st(q)), "afm_mha_fwd_train")
        ctx.save_for_backward(q, out, lse)
        ctx.cfg = (B, T, heads, Cn, drop)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = ffi.load()
        q, out, lse = ctx.saved_tensors
        B, T, heads, Cn, drop = ctx.cfg
        dout = _c(dout)
        dqkv = torch.empty_like(q)
        ws = torch.empty(B * heads * T, device=q.device, dtype=torch.float32)
        ffi.check(lib.afm_mha_bwd(q.data_ptr(), None, out.data_ptr(), dout.data_ptr(), lse.data_ptr(), dqkv.data_ptr(), B, T, heads, Cn // heads,
                                  drop[0], drop[1], drop[2], ws.data_ptr(), ws.numel() * 4, _st(q)), "afm_mha_bwd")
        return dqkv, None, None


def self_attention(qkv, heads: int, drop=(0.0, 0, 0)):
    return _SelfAttentionFn.apply(qkv, heads, drop)


class _FewQueryAttentionFn(torch.autograd.Function):
    """Q [B, 2, C] over K, V [B, N, C] (ContactPerceiver encoder cross-attention, modules.py:301-381)."""

    @staticmethod
    def forward(ctx, q, k, v, heads, drop):
        lib = ffi.load()
        q, k, v = _c(q), _c(k), _c(v)
        B, N, Cn = k.shape
        P = torch.empty(B, heads * 2, N, device=q.device, dtype=torch.float32)
        O = torch.empty(B, 2, Cn, device=q.device, dtype=torch.float32)
        ws = torch.empty(max(int(lib.afm_xq_workspace_bytes(B, N, Cn)), 16), dtype=torch.uint8, device=q.device)
        ffi.check(lib.afm_xq_attention_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), P.data_ptr(), O.data_ptr(), B, N, heads, Cn, drop[0], drop[1],
                                           drop[2], ws.data_ptr(), ws.numel(), _st(q)), "afm_xq_attention_fwd")
        ctx.save_for_backward(q, k, v, P)
        ctx.cfg = (B, N, heads, Cn, drop)
        return O

    @staticmethod
    def backward(ctx, dO):
        lib = ffi.load()
        q, k, v, P = ctx.saved_tensors
        B, N, heads, Cn, drop = ctx.cfg
        dO = _c(dO)
        dS = torch.empty_like(P)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        ws = torch.empty(max(int(lib.afm_xq_workspace_bytes(B, N, Cn)), 16), dtype=torch.uint8, device=q.device)
        ffi.check(lib.afm_xq_attention_bwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), P.data_ptr(), dO.data_ptr(), dS.data_ptr(), dq.data_ptr(),
                                           dk.data_ptr(), dv.data_ptr(), B, N, heads, Cn, drop[0], drop[1], drop[2], ws.data_ptr(), ws.numel(),
                                           _st(q)), "afm_xq_attention_bwd")
        return dq, dk, dv, None, None


def few_query_attention(q, k, v, heads: int, drop=(0.0, 0, 0)):
    return _FewQueryAttentionFn.apply(q, k, v, heads, drop)


class _FewKeyAttentionFn(torch.autograd.Function):
    """Q [B, N, C] over K, V [B, 2, C] (ContactPerceiver decoder cross-attention)."""

    @staticmethod
    def forward(ctx, q, k, v, heads, drop):
        lib = ffi.load()
        q, k, v = _c(q), _c(k), _c(v)
        B, N, Cn = q.shape
        O = torch.empty_like(q)
        ffi.check(lib.afm_xk_attention_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), O.data_ptr(), B, N, heads, Cn, drop[0], drop[1], drop[2],
                                           _st(q)), "afm_xk_attention_fwd")
        ctx.save_for_backward(q, k, v)
        ctx.cfg = (B, N, heads, Cn, drop)
        return O

    @staticmethod
    def backward(ctx, dO):
        lib = ffi.load()
        q, k, v = ctx.saved_tensors
        B, N, heads, Cn, drop = ctx.cfg
        dO = _c(dO)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        ws = torch.empty(max(int(lib.afm_xk_workspace_bytes(B, N, Cn)), 16), dtype=torch.uint8, device=q.device)
        ffi.check(lib.afm_xk_attention_bwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), dO.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), B, N,
                                           heads, Cn, drop[0], drop[1], drop[2], ws.data_ptr(), ws.numel(), _st(q)), "afm_xk_attention_bwd")
        return dq, dk, dv, None, None


def few_key_attention(q, k, v, heads: int, drop=(0.0, 0, 0)):
    return _FewKeyAttentionFn.apply(q, k, v, heads, drop)


class _SegmentMeanFn(torch.autograd.Function):
    """x [B, N, C] -> mean over the N points of every sample [B, C] (PointSceneMLP scene feature, cdm.py:35)."""

    @staticmethod
    def forward(ctx, x):
        xc = _c(x)
        B, N, Cn = xc.shape
        out = torch.empty(B, Cn, device=xc.device, dtype=torch.float32)
        ffi.check(ffi.load().afm_segment_mean(xc.data_ptr(), out.data_ptr(), B, N, Cn, _st(xc)), "afm_segment_mean")
        ctx.dims = (B, N, Cn)
        return out

    @staticmethod
    def backward(ctx, dy):
        B, N, Cn = ctx.dims
        return (_c(dy) / N).view(B, 1, Cn).expand(B, N, Cn)


def segment_mean(x):
    return _SegmentMeanFn.apply(x)


# ------------------------------------------------------------------------------------------------ loss
class _MaskedMseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, target, pred, frame_mask):
        t, p = _c(target), _c(pred)
        B, D = t.shape[0], t.shape[-1]
        L = t.numel() // max(B * D, 1)
        km = None if frame_mask is None else frame_mask.reshape(B, L).to(torch.uint8).contiguous()
        out = torch.empty(B, device=t.device, dtype=torch.float32)
        ffi.check(ffi.load().afm_masked_mse(t.data_ptr(), p.data_ptr(), ffi.ptr(km), out.data_ptr(), B, L, D, _st(t)), "afm_masked_mse")
        ctx.save_for_backward(t, p, km)
        ctx.dims = (B, L, D)
        return out

    @staticmethod
    def backward(ctx, dloss):
        t, p, km = ctx.saved_tensors
        B, L, D = ctx.dims
        dpred = torch.empty_like(p)
        dl = _c(dloss)
        ffi.check(ffi.load().afm_masked_mse_bwd(t.data_ptr(), p.data_ptr(), ffi.ptr(km), dl.data_ptr(), dpred.data_ptr(), B, L, D, _st(t)),
                  "afm_masked_mse_bwd")
        return None, dpred, None


def masked_mse(target, pred, frame_mask):
    """Per-sample masked MSE [B] (gaussian_diffusion.py:815-818) with gradient to ``pred``."""
    return _MaskedMseFn.apply(target, pred, frame_mask)


# ------------------------------------------------------------------------------------------------ optimiser
def adamw_step(params, state: dict, *, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0) -> None:
    """torch.optim.AdamW semantics (utils/training.py:48-53) for every parameter with a gradient in ONE kernel launch
    (afm_adamw_multi); ``state`` holds the step count and the per-parameter moments and is created on first use."""
    lib = ffi.load()
    live = [p for p in params if p.grad is not None]
    if not live:
        return
    ffi.require_gpu(*live)
    state["step"] = state.get("step", 0) + 1
    rows = []
    keep = []
    for p in live:
        assert p.is_contiguous() and p.dtype == torch.float32
        st = state.get(p)
        if st is None:
            st = (torch.zeros_like(p, memory_format=torch.contiguous_format), torch.zeros_like(p, memory_format=torch.contiguous_format))
            state[p] = st
        g = ffi.f32c(p.grad)
        keep.append(g)
        rows.append((p.data_ptr(), g.data_ptr(), st[0].data_ptr(), st[1].data_ptr(), p.numel()))
    table = torch.tensor(rows, dtype=torch.int64).to(live[0].device, non_blocking=True)       # 5 x int64 per tensor = afm_adamw_tensor
    ffi.check(lib.afm_adamw_multi(table.data_ptr(), len(rows), max(r[4] for r in rows), lr, betas[0], betas[1], eps, weight_decay,
                                  state["step"], _st(live[0])), "afm_adamw_multi")
    state["_keep"] = (table, keep)          # alive until the next step's launch is enqueued behind this one
