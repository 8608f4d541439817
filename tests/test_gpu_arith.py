"""`-m gpu`: WORST-CASE comparison of afm_linear's arithmetics - the six-product bf16 split (x6), the exact nine-product
split (x9) and the native f32 MFMA kernel against float64, next to the arithmetic the reference's CPU path is built from: an UNFUSED float32
multiply + add chain, products rounded to f32 and added left to right (numpy, no FMA contraction).  Not an rms over Gaussians: adversarial
input families, worst output element of each.

    err(kernel) = max over outputs of |kernel - float64| / sum_k |x_k||w_k|        (the classical normalisation of a dot product's error bound)

What the six-product form drops per product x w is x2 w3 + x3 w2 + x3 w3 with |x2| <= 2^-8 |x|, |x3| <= 2^-16 |x| (the split rounds to
nearest): at most 2^-23 |x w|, two f32 roundings of that product.  The question the decision hangs on is whether that ever shows above what
f32 arithmetic loses anyway.  Families: Gaussian baseline; catastrophic cancellation (the sum is ~1e-7 of sum|x||w|, in an adjacent-pairs and
in a far-apart arrangement); 2^+-60 dynamic range inside a row; mantissas SEARCHED to maximise the dropped terms with every product of one
sign (the dropped parts add up coherently: the worst case for x6's bias); operands with subnormal split terms; K = 128 ... 4096.

Measured on MI355X (round 6, profiles/r06_arith_worstcase.json; the asserts below are what that table supports):
  * x6 is NEVER above x9: equal to three digits in 20 of 28 rows, slightly lower in the rest - the dropped products are invisible even where
    they add up coherently (worst_mantissa_one_sign: 5.6e-7 / 8.2e-7 / 4.5e-7 / 3.6e-7 for both forms).
  * Against the unfused f32 chain: x6 (and x9) are BELOW it in 24 of 28 rows (3x - 10x below at K >= 1024: the matrix pipe adds 16 products
    per accumulation where the chain rounds after every one) and above it in four: gaussian / range_2pm60 / range_both_2pm30 at K = 128
    (1.17x / 1.47x / 1.16x) and cancel_adjacent at K = 4096 (1.30x, at an error level of 4e-9) - in every one of them x9, with its EXACT
    products, is above the chain by the same amount and in the first the native f32 MFMA instruction is too: the excess is the accumulation
    order of the matrix pipe, not the dropped products.  So the literal rule "max err(x6) <= max err(chain)" fails exactly where it fails
    for the exact-product forms; the rule that separates the arithmetics - x6 vs x9 - holds everywhere.  Decision: six products are the
    sampling default (afm.ops.DEFAULT_PRODUCTS, AFM_ARITH_DEFAULT), nine stay on the training tape.
  * Domain of BOTH split forms: a split term below 2^-126 is a bf16 subnormal and is flushed to zero (the hardware's conversion / matrix
    pipe), i.e. operands below ~2^-110 lose their trailing terms and f32-subnormal operands vanish: an ABSOLUTE error of at most 2^-125
    per operand element times the other operand - invisible next to O(1) activations, but not f32's gradual underflow.  The native f32
    kernel (AFM_ARITH_F32) has no such limit.  Pinned below."""
import json
import math
import os

import numpy as np
import pytest
import torch

from afm import ffi, ops
from gpu_util import dev

pytestmark = pytest.mark.gpu

M, N = 96, 64


def _rng(tag):
    return np.random.default_rng(sum(ord(c) * (i + 1) for i, c in enumerate(tag)))          # (hash() is salted per process)


def bf16_rne(x):
    """float32 -> nearest bf16 (ties to even), returned as float32: what v_cvt_pk_bf16_f32 does to a normal number."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32)
    return ((u + (np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1)))) & np.uint32(0xFFFF0000)).view(np.float32)


def split3(x):
    x1 = bf16_rne(x); r = x - x1
    x2 = bf16_rne(r); r = r - x2
    return x1, x2, bf16_rne(r)


_POOL = None


def _worst_mantissa(rng, shape):
    """f32 values whose second AND third split terms are positive and as large as they get (searched, not guessed: of 2^20 random mantissas
    in [1, 2) the 256 with the largest x2 x3 among those with x2 > 0, x3 > 0 - |x2| -> 2^-9 |x|, |x3| -> 2^-17 |x|), times 2^-3 .. 2^3.  With both
    operands drawn from this pool every dropped product x2 w3, x3 w2, x3 w3 is positive: the six-product form's error is a coherent bias."""
    global _POOL
    if _POOL is None:
        r = np.random.default_rng(7)
        c = (np.uint32(0x3F800000) | r.integers(0, 1 << 23, size=1 << 20).astype(np.uint32)).view(np.float32)
        _, x2, x3 = split3(c)
        score = np.where((x2 > 0) & (x3 > 0), x2.astype(np.float64) * x3.astype(np.float64), -1.0)
        _POOL = c[np.argsort(score)[-256:]]
    return _POOL[rng.integers(0, _POOL.size, size=shape)] * np.exp2(rng.integers(-3, 4, size=shape)).astype(np.float32)


def family(name, K):
    """-> (x [M, K], w [N, K]) float32 numpy arrays."""
    rng = _rng(f"{name}{K}")
    g = lambda *s: rng.standard_normal(s).astype(np.float32)
    if name == "gaussian":
        return g(M, K), g(N, K) / np.float32(math.sqrt(K))
    if name == "cancel_adjacent":
        # x_{2i+1} w_{2i+1} = -(1 + d) x_{2i} w_{2i}, d ~ 1e-7 ... 1e-6: neighbours cancel, the sum is ~1e-7 of sum |x||w|
        x, w = g(M, K), g(N, K)
        x[:, 1::2] = x[:, 0::2]
        w[:, 1::2] = -w[:, 0::2] * (1.0 + rng.integers(1, 8, size=(N, K // 2)).astype(np.float32) * np.float32(2.0**-23))
        return x, w
    if name == "cancel_far":
        # the cancelling partner sits K / 2 terms away: a chain carries the large partial sum through K / 2 small additions first
        x, w = g(M, K), g(N, K)
        h = K // 2
        x[:, h:] = x[:, :h]
        w[:, h:] = -w[:, :h] * (1.0 + rng.integers(1, 8, size=(N, h)).astype(np.float32) * np.float32(2.0**-23))
        x[:, :8] *= np.float32(2.0**20); x[:, h:h + 8] *= np.float32(2.0**20)            # a few dominant pairs on top
        return x, w
    if name == "range_2pm60":
        return g(M, K) * np.exp2(rng.integers(-60, 61, size=(M, K))).astype(np.float32), g(N, K)
    if name == "range_both_2pm30":
        return (g(M, K) * np.exp2(rng.integers(-30, 31, size=(M, K))).astype(np.float32),
                g(N, K) * np.exp2(rng.integers(-30, 31, size=(N, K))).astype(np.float32))
    if name == "worst_mantissa_one_sign":
        return _worst_mantissa(rng, (M, K)), _worst_mantissa(rng, (N, K))
    if name == "worst_mantissa_signed":
        s = lambda *sh: np.where(rng.random(sh) < 0.5, np.float32(-1), np.float32(1))
        return _worst_mantissa(rng, (M, K)) * s(M, K), _worst_mantissa(rng, (N, K)) * s(N, K)
    if name == "subnormal_residuals":
        # |x| ~ 2^-118: x3 ~ 2^-134 lies below bf16's (= f32's) smallest normal 2^-126; w ~ 2^+50 keeps every product a normal f32
        return g(M, K) * np.float32(2.0**-118), g(N, K) * np.float32(2.0**50)
    if name == "subnormal_operands":
        return g(M, K) * np.float32(2.0**-100) * np.float32(2.0**-40), g(N, K) * np.float32(2.0**100)
    raise KeyError(name)


def f32_chain(x, w):
    """The unfused float32 chain: acc = fl(acc + fl(x_k w_k)), k = 0 .. K-1 (numpy float32 arithmetic: one rounding per operation)."""
    acc = np.zeros((x.shape[0], w.shape[0]), np.float32)
    for k in range(x.shape[1]):
        acc = acc + np.multiply.outer(x[:, k], w[:, k])
    return acc


def measure(name, K):
    x, w = family(name, K)
    with np.errstate(all="ignore"):
        ref = x.astype(np.float64) @ w.astype(np.float64).T
        scale = np.abs(x).astype(np.float64) @ np.abs(w).astype(np.float64).T
        chain = f32_chain(x, w)
    scale = np.maximum(scale, np.finfo(np.float64).tiny)
    out = {"chain_f32": float((np.abs(chain.astype(np.float64) - ref) / scale).max())}
    xd, wd = torch.from_numpy(x).to(dev()), torch.from_numpy(w).to(dev())
    saved = ops.get_gemm_split()
    try:
        for tag, products in (("native_f32_mfma", 0), ("x9", 9), ("x6", 6)):
            ops.set_gemm_split(products, 0)
            got = ops.linear(xd, wd).double().cpu().numpy()
            assert np.isfinite(got).all(), (name, K, tag)
            out[tag] = float((np.abs(got - ref) / scale).max())
    finally:
        ops.set_gemm_split(*saved)
    out["sum_over_scale_median"] = float(np.median(np.abs(ref) / scale))
    return out


FAMILIES = ["gaussian", "cancel_adjacent", "cancel_far", "range_2pm60", "range_both_2pm30", "worst_mantissa_one_sign", "worst_mantissa_signed"]
KS = [128, 512, 1024, 4096]
_table = {}


@pytest.mark.parametrize("K", KS)
@pytest.mark.parametrize("name", FAMILIES)
def test_six_products_worst_case(name, K):
    """Per family and K: (1) x6 is not above x9 - the dropped products never show; (2) both split forms stay in the error class of f32
    arithmetic: within a factor two of the unfused f32 chain's worst element (one bit) and of the native f32 MFMA kernel's, and - the literal
    rule of VERDICT r5 item 4 - at or below the chain wherever the exact-product form x9 is."""
    r = measure(name, K)
    _table[f"{name} K={K}"] = r
    r["x6_le_chain"] = r["x6"] <= r["chain_f32"]
    print(f"[arith] {name:26s} K={K:5d}  chain {r['chain_f32']:.3e}  native {r['native_f32_mfma']:.3e}  x9 {r['x9']:.3e}  x6 {r['x6']:.3e}  "
          f"(|sum| / sum|x||w| median {r['sum_over_scale_median']:.1e}){'' if r['x6_le_chain'] else '   x6 > chain'}")
    assert r["x6"] <= 1.02 * r["x9"] + 2.0**-32, f"x6 {r['x6']:.3e} above x9 {r['x9']:.3e}: the dropped products show"
    assert r["x6"] <= 2.0 * max(r["chain_f32"], r["native_f32_mfma"]), f"x6 {r['x6']:.3e} outside the f32 error class (chain {r['chain_f32']:.3e})"
    if r["x9"] <= r["chain_f32"]:
        assert r["x6"] <= r["chain_f32"], f"x6 {r['x6']:.3e} above the unfused f32 chain {r['chain_f32']:.3e} where x9 is not"
    # and in absolute terms: the dropped products are bounded by 2^-23 per product whatever the accumulation does
    assert r["x6"] <= r["x9"] + 2.0**-23 * 1.01


@pytest.mark.parametrize("name", ["subnormal_residuals", "subnormal_operands"])
def test_split_forms_below_the_bf16_normal_range(name):
    """The split's stated domain: a split term below 2^-126 is a bf16 SUBNORMAL and the hardware flushes it - measured: |x| ~ 2^-118 loses its
    third term (relative error 2.4e-6 where the chain has 2.6e-7), f32-subnormal operands vanish entirely, for x9 and x6 ALIKE; the native
    f32 MFMA kernel keeps both.  What is asserted is the bound DESIGN section 2 states: an absolute error of at most 2^-125 per operand element
    (every flushed term is below 2^-126, at most two of an element's three), times the other operand - on top of ordinary f32 rounding."""
    K = 512
    x, w = family(name, K)
    ref = x.astype(np.float64) @ w.astype(np.float64).T
    scale = np.abs(x).astype(np.float64) @ np.abs(w).astype(np.float64).T
    bound = 2.0**-125 * (np.ones_like(x, np.float64) @ np.abs(w).astype(np.float64).T + np.abs(x).astype(np.float64) @ np.ones_like(w, np.float64).T) + 2.0**-21 * scale
    xd, wd = torch.from_numpy(x).to(dev()), torch.from_numpy(w).to(dev())
    saved = ops.get_gemm_split()
    r = {}
    try:
        for tag, products in (("native_f32_mfma", 0), ("x9", 9), ("x6", 6)):
            ops.set_gemm_split(products, 0)
            got = ops.linear(xd, wd).double().cpu().numpy()
            assert np.isfinite(got).all()
            r[tag] = float((np.abs(got - ref) / scale).max())
            if products:
                assert (np.abs(got - ref) <= bound).all(), f"{tag}: flushed-term bound exceeded by {float((np.abs(got - ref) / bound).max()):.2f}x"
    finally:
        ops.set_gemm_split(*saved)
    with np.errstate(all="ignore"):
        r["chain_f32"] = float((np.abs(f32_chain(x, w).astype(np.float64) - ref) / scale).max())
    _table[f"{name} K={K}"] = r
    print(f"[arith] {name:26s} K={K:5d}  chain {r['chain_f32']:.3e}  native {r['native_f32_mfma']:.3e}  x9 {r['x9']:.3e}  x6 {r['x6']:.3e}   (relative to sum|x||w|)")
    assert r["x6"] <= 1.02 * r["x9"] + 2.0**-32                        # six products are no worse than nine out here
    assert r["native_f32_mfma"] <= 2.0**-21                            # the f32 MFMA kernel has no such domain limit


def test_zz_write_arith_table():
    """Not a check: stores the table the tests above measured (gpurun_out/arith/arith_worstcase.json -> profiles/r06_arith_worstcase.json)."""
    if not _table:
        pytest.skip("run together with the measuring tests")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = os.path.join(root, "gpurun_out", "arith")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "arith_worstcase.json"), "w") as f:
        json.dump({"normalisation": "max over outputs of |result - float64| / sum_k |x_k||w_k|", "M": M, "N": N, "rows": _table}, f, indent=1)
