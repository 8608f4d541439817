"""`-m gpu`: randomised shape sweeps (hypothesis) of the primitive kernels against float64 torch - ragged M/N/K (including the
non-vectorised and K-tail paths of the GEMM, row maps, any-width LayerNorm, odd attention lengths)."""
import pytest
import torch
from hypothesis import given, settings, strategies as st

from afm import autograd as AG
from afm import ffi, ops
from gpu_util import dev

pytestmark = pytest.mark.gpu
SET = dict(max_examples=40, deadline=None, derandomize=True)      # fixed example set: the round-end run must be reproducible


def _rand(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g)


@settings(**SET)
@given(M=st.integers(1, 300), N=st.integers(1, 200), K=st.integers(1, 200), act=st.sampled_from([0, 1, 2, 3]), seed=st.integers(0, 10**6),
       use_res=st.booleans(), use_bias=st.booleans())
def test_linear_random_shapes(M, N, K, act, seed, use_res, use_bias):
    x, w = _rand((M, K), seed), _rand((N, K), seed + 1) / K ** 0.5
    b = _rand((N,), seed + 2) if use_bias else None
    r = _rand((M, N), seed + 3) if use_res else None
    y = x.double() @ w.double().t() + (b.double() if use_bias else 0)
    y = {0: lambda v: v, 1: lambda v: torch.nn.functional.gelu(v), 2: torch.relu, 3: torch.nn.functional.silu}[act](y)
    if use_res:
        y = y + r.double()
    got = ops.linear(x.to(dev()), w.to(dev()), None if b is None else b.to(dev()), act=act, residual=None if r is None else r.to(dev()))
    assert (got.cpu().double() - y).abs().max().item() <= 2e-5 * max(1.0, y.abs().max().item())


@settings(**SET)
@given(M=st.integers(1, 3000), N=st.integers(1, 130), K=st.integers(1, 130), seed=st.integers(0, 10**6))
def test_wgrad_random_shapes(M, N, K, seed):
    dy, x = _rand((M, N), seed), _rand((M, K), seed + 1)
    dW, db = AG._wgrad(dy.to(dev()), x.to(dev()), M, N, K)
    ref = dy.double().t() @ x.double()
    assert (dW.cpu().double() - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    assert (db.cpu().double() - dy.double().sum(0)).abs().max().item() <= 2e-5 * max(1.0, dy.double().sum(0).abs().max().item())


@settings(**SET)
@given(rows=st.integers(1, 200), dim=st.integers(2, 700), seed=st.integers(0, 10**6))
def test_layernorm_random_shapes(rows, dim, seed):
    x = _rand((rows, dim), seed) * 2 + 0.5
    g, b = 1 + 0.1 * _rand((dim,), seed + 1), 0.1 * _rand((dim,), seed + 2)
    ref = torch.nn.functional.layer_norm(x.double(), (dim,), g.double(), b.double(), 1e-5)
    got = ops.layernorm(x.to(dev()), g.to(dev()), b.to(dev()))
    # a row of nearly equal values (tiny variance next to eps) amplifies the f32 rounding of x - mean by rstd: allow that much
    rstd = (x.double().var(dim=1, unbiased=False) + 1e-5).rsqrt()
    tol = 2e-5 + 8 * 2.0 ** -24 * x.double().abs().amax(dim=1) * rstd * g.double().abs().max()
    assert ((got.cpu().double() - ref).abs().amax(dim=1) <= tol).all()


@settings(max_examples=20, deadline=None, derandomize=True)
@given(B=st.integers(1, 3), T=st.integers(1, 150), H=st.sampled_from([1, 2, 8]), seed=st.integers(0, 10**6), masked=st.booleans())
def test_attention_random_lengths(B, T, H, seed, masked):
    d = 64 * H
    qkv = _rand((B, T, 3 * d), seed)
    mask = None
    if masked and T > 1:
        mask = torch.zeros(B, T, dtype=torch.bool)
        mask[0, T - max(1, T // 3):] = True
    q, k, v = [t.double().view(B, T, H, 64).transpose(1, 2) for t in qkv.split(d, dim=-1)]
    s = q @ k.transpose(-1, -2) / 8.0
    if mask is not None:
        s = s.masked_fill(mask[:, None, None, :], float("-inf"))
    ref = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, T, d)
    got = ops.mha(qkv.to(dev()), None if mask is None else mask.to(dev()), H)
    assert (got.cpu().double() - ref).abs().max().item() <= 2e-5


@settings(max_examples=25, deadline=None, derandomize=True)
@given(B=st.integers(1, 3), n=st.integers(1, 700), frac=st.floats(0.05, 1.0), k=st.sampled_from([3, 8, 16]), seed=st.integers(0, 10**6),
       dup=st.booleans())
def test_fps_knn_random_sizes_bit_exact(B, n, frac, k, seed, dup):
    """FPS / kNN indices vs the oracle on ragged sizes (n < k, m not a multiple of the workgroup, duplicated points)."""
    from afm import pointops
    from oracle import pointops_ref as po
    m = max(1, int(n * frac))
    p = _rand((B * n, 3), seed)
    if dup and n > 4:
        p[1::3] = p[0::3][: p[1::3].shape[0]]                      # exact duplicates -> ties in both operators
    o = (torch.arange(1, B + 1, dtype=torch.int32) * n)
    no = (torch.arange(1, B + 1, dtype=torch.int32) * m)
    idx_ref = po.furthest_sampling(p, o, no)
    idx = pointops.furthest_point_sampling(p.to(dev()), B, n, m)
    assert torch.equal(idx.cpu().int(), idx_ref.int())
    q = p[idx_ref.long()]
    ki_ref, d_ref = po.knn_query(k, p, q, o, no)
    ki, d2 = pointops.knn(k, p.to(dev()), q.to(dev()), B, n, m)
    assert torch.equal(ki.cpu().int(), ki_ref.int())
    assert torch.equal(d2.cpu(), d_ref)
