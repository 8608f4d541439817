"""`-m gpu`: the training (backward) path on the HIP kernels vs the CPU oracle (torch autograd over the functional
restatement, oracle/train_ref.py) and vs gradients of the REAL reference (tests/golden/cmdm_training_grads.npz).

Tolerances: f32 MFMA products are exact f32; differences come from summation order.  Gradients are compared after
scaling by the reference tensor's max |value| (relative 1e-4 .. 5e-4: reductions over up to ~10^4 rows)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from afm import autograd as AG
from afm import ffi, synth
from afm.base import create_model_and_diffusion
from conftest import golden
from gpu_util import dev, load_named_weights, report
from test_gpu_cmdm import cmdm_cfg

pytestmark = pytest.mark.gpu


def rel(name, got, want, tol):
    scale = max(want.detach().abs().max().item(), 1e-12)
    return report(name + " (scaled)", got.detach().cpu() / scale, want.detach().cpu() / scale, tol)


def g(name, shape):
    return synth.gaussian(name, shape)


# ------------------------------------------------------------------------------------------------ primitives
@pytest.mark.parametrize("M,N,K", [(652, 512, 512), (10432, 1536, 512), (300, 263, 512), (300, 512, 263), (32, 512, 512), (7, 5, 3), (0, 8, 8),
                                   # point-cloud linears: millions of grouped rows, a few channels (stream kernel, no LDS)
                                   (300001, 32, 3), (100000, 4, 32), (70000, 3, 3), (50000, 64, 64), (40000, 256, 3), (40000, 16, 128),
                                   (20000, 32, 256), (30000, 64, 35), (8000, 128, 67),
                                   # dense rows: the flat-staged kernel (csrc/train.hip wgrad_flat_kernel), ragged tails, every tile-count family
                                   (4096, 3, 3), (4097, 4, 4), (99999, 8, 8), (65536, 64, 64), (33001, 35, 64), (12345, 67, 128), (10001, 131, 67),
                                   (9000, 256, 3), (9001, 8, 256), (5000, 96, 100), (70000, 16, 16), (6000, 192, 40)])
def test_linear_wgrad(M, N, K):
    dy, x = g("wg_dy", (max(M, 1), N))[:M].contiguous(), g("wg_x", (max(M, 1), K))[:M].contiguous()
    dW, db = AG._wgrad(dy.to(dev()), x.to(dev()), M, N, K)
    rel(f"dW {M}x{N}x{K}", dW, dy.double().t() @ x.double(), 2e-5)
    rel(f"db {M}x{N}", db, dy.double().sum(0), 2e-5) if M else report("db empty", db, torch.zeros(N), 0.0)
    dW2, _ = AG._wgrad(dy.to(dev()), x.to(dev()), M, N, K)
    assert torch.equal(dW, dW2), "weight gradient is not deterministic"


def test_linear_wgrad_row_maps():
    """Token-subset rows (motion tokens of every sample) on both operands."""
    B, T, L, N, K = 3, 50, 20, 263, 512
    dy, x = g("wgm_dy", (B * L, N)), g("wgm_x", (B * T, K))
    dW, db = AG._wgrad(dy.to(dev()), x.to(dev()), B * L, N, K, x_map=(L, T, T - L))
    xs = x.view(B, T, K)[:, T - L:, :].reshape(B * L, K)
    rel("dW with x row map", dW, dy.double().t() @ xs.double(), 2e-5)


def test_transpose():
    for r, c in ((512, 1536), (263, 512), (33, 7)):
        w = g("tr", (r, c)).to(dev())
        assert torch.equal(AG._transpose(w), w.t().contiguous())


@pytest.mark.parametrize("rows,dim", [(10432, 512), (37, 256), (5, 1024)])
def test_layernorm_backward(rows, dim):
    x, dy = g("lnb_x", (rows, dim)), g("lnb_dy", (rows, dim))
    gam, bet = 1.0 + 0.1 * g("lnb_g", (dim,)), 0.1 * g("lnb_b", (dim,))
    xr, gr, br = x.double().requires_grad_(True), gam.double().requires_grad_(True), bet.double().requires_grad_(True)
    F.layer_norm(xr, (dim,), gr, br, 1e-5).backward(dy.double())
    dx, dxd, dg, db = AG._layernorm_bwd(x.to(dev()), gam.to(dev()), dy.to(dev()))
    assert dxd is dx
    rel("LN dx", dx, xr.grad, 2e-5)
    rel("LN dgamma", dg, gr.grad, 2e-5)
    rel("LN dbeta", db, br.grad, 2e-5)


def _attn_ref(qkv, mask, H, keep=None):
    """softmax(QK^T/sqrt(dh) + mask) V in float64; keep [B,H,T,T] multiplies P before P V (dropout)."""
    B, T, d3 = qkv.shape
    d = d3 // 3
    q, k, v = [t.view(B, T, H, d // H).transpose(1, 2) for t in qkv.split(d, dim=-1)]
    s = q @ k.transpose(-1, -2) / (d // H) ** 0.5
    if mask is not None:
        s = s.masked_fill(mask[:, None, None, :], float("-inf"))
    p = torch.softmax(s, -1)
    lse = torch.logsumexp(s, -1)
    if keep is not None:
        p = p * keep
    return (p @ v).transpose(1, 2).reshape(B, T, d), lse


def _mha_train(qkv, km, H, drop=(0.0, 0, 0)):
    B, T, d3 = qkv.shape
    d = d3 // 3
    out = torch.empty(B, T, d, device=qkv.device)
    lse = torch.empty(B, H, T, device=qkv.device)
    ffi.check(ffi.load().afm_mha_fwd_train(qkv.data_ptr(), ffi.ptr(km), out.data_ptr(), lse.data_ptr(), B, T, H, d // H, drop[0], drop[1],
                                           drop[2], ffi.stream_of(qkv)), "afm_mha_fwd_train")
    return out, lse


def _mha_bwd(qkv, km, out, dout, lse, H, drop=(0.0, 0, 0)):
    B, T, d3 = qkv.shape
    dqkv = torch.empty_like(qkv)
    ws = torch.empty(B * H * T, device=qkv.device)
    ffi.check(ffi.load().afm_mha_bwd(qkv.data_ptr(), ffi.ptr(km), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), dqkv.data_ptr(), B, T, H,
                                     d3 // 3 // H, drop[0], drop[1], drop[2], ws.data_ptr(), ws.numel() * 4, ffi.stream_of(qkv)), "afm_mha_bwd")
    return dqkv


@pytest.mark.parametrize("B,T,H,masked", [(2, 326, 8, True), (3, 70, 8, True), (1, 33, 2, False), (2, 400, 4, True)])
def test_attention_backward(B, T, H, masked):
    d = 64 * H
    qkv = g("ab_qkv", (B, T, 3 * d))
    dout = g("ab_do", (B, T, d))
    mask = None
    if masked:
        mask = torch.zeros(B, T, dtype=torch.bool)
        mask[0, T - 11:] = True
        if B > 1:
            mask[1, T // 2:] = True
    qr = qkv.double().requires_grad_(True)
    o_ref, lse_ref = _attn_ref(qr, mask, H)
    o_ref.backward(dout.double())
    km = None if mask is None else mask.to(torch.uint8).to(dev())
    out, lse = _mha_train(qkv.to(dev()), km, H)
    report("attention out (train fwd)", out, o_ref, 2e-5)
    report("attention lse", lse, lse_ref, 2e-5)
    dqkv = _mha_bwd(qkv.to(dev()), km, out, dout.to(dev()), lse, H)
    for i, n in enumerate("QKV"):
        rel(f"attention d{n}", dqkv[..., i * d:(i + 1) * d], qr.grad[..., i * d:(i + 1) * d], 5e-5)
    assert torch.equal(dqkv, _mha_bwd(qkv.to(dev()), km, out, dout.to(dev()), lse, H)), "attention backward is not deterministic"


def test_attention_dropout_forward_backward_consistent():
    """With V = one-hot rows the forward output IS the dropped probability matrix, which reveals the keep mask; the general
    forward / backward must then equal the float64 reference evaluated with that same mask."""
    B, T, H, p = 2, 64, 2, 0.25
    d = 64 * H
    drop = (p, 1234567, 40)
    qkv = g("ad_qkv", (B, T, 3 * d))
    probe = qkv.clone()
    probe[..., 2 * d:] = torch.eye(64).repeat(1, H).unsqueeze(0)              # V_h = I for every head
    pd, _ = _mha_train(probe.to(dev()), None, H, drop)
    p0, _ = _mha_train(probe.to(dev()), None, H)
    pd, p0 = pd.cpu().view(B, T, H, 64).transpose(1, 2), p0.cpu().view(B, T, H, 64).transpose(1, 2)     # [B,H,Tq,Tk]
    keep = torch.where(pd > 0, torch.full_like(pd, 1 / (1 - p)), torch.zeros_like(pd))
    report("dropped P == P * keep", pd, p0 * keep, 1e-6)
    rate = (keep > 0).float().mean().item()
    assert abs(rate - (1 - p)) < 0.02, rate
    dout = g("ad_do", (B, T, d))
    qr = qkv.double().requires_grad_(True)
    o_ref, _ = _attn_ref(qr, None, H, keep.double())
    o_ref.backward(dout.double())
    out, lse = _mha_train(qkv.to(dev()), None, H, drop)
    report("attention out with dropout", out, o_ref, 2e-5)
    dqkv = _mha_bwd(qkv.to(dev()), None, out, dout.to(dev()), lse, H, drop)
    rel("attention dqkv with dropout", dqkv, qr.grad, 5e-5)
    out2, _ = _mha_train(qkv.to(dev()), None, H, (p, 1234568, 40))
    assert not torch.equal(out, out2), "a different seed must give a different mask"


def test_epilogue_dropout_and_rowop_share_masks():
    """The GEMM epilogue, afm_rowop and afm_layernorm_bwd regenerate the same (seed, id, row, col) mask."""
    M, N, K, p = 300, 512, 64, 0.1
    drop = (p, 99, 7)
    x, w = g("ed_x", (M, K)).to(dev()), g("ed_w", (N, K)).to(dev())
    plain = torch.empty(M, N, device=dev())
    AG._gemm(x, w, plain, M, N, K)
    dropped = torch.empty(M, N, device=dev())
    AG._gemm(x, w, dropped, M, N, K, drop=drop)
    assert torch.equal(dropped, AG._rowop(plain, drop=drop))
    keep = (dropped != 0).float().mean().item()
    assert abs(keep - (1 - p)) < 0.01, keep
    report("kept values scaled by 1/(1-p)", dropped, torch.where(dropped != 0, plain / (1 - p), torch.zeros_like(plain)), 1e-5)
    dy = g("ed_dy", (M, N)).to(dev())
    gam = torch.ones(N, device=dev())
    dx, dxd, _, _ = AG._layernorm_bwd(plain, gam, dy, drop=drop)
    assert torch.equal(dxd, AG._rowop(dx, drop=drop))


def test_stand_alone_linear_and_posenc_grads():
    B, T, L, K, N = 2, 30, 12, 64, 263
    x = g("sl_x", (B, T, K))
    w, b = g("sl_w", (N, K)) * 0.1, g("sl_b", (N,)) * 0.1
    dy = g("sl_dy", (B * L, N))
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    (F.silu(F.linear(xr[:, T - L:, :], wr, br)).reshape(B * L, N) * dy.double()).sum().backward()
    xg, wg, bg = (t.to(dev()).requires_grad_(True) for t in (x, w, b))
    y = AG.linear(xg.view(B * T, K), wg, bg, act=ffi.ACT_SILU, a_map=(L, T, T - L), rows=B * L)
    (y * dy.to(dev())).sum().backward()
    rel("linear dx (row-gathered)", xg.grad, xr.grad, 2e-5)
    rel("linear dW", wg.grad, wr.grad, 2e-5)
    rel("linear db", bg.grad, br.grad, 2e-5)
    pe = g("sl_pe", (T, K)).to(dev())
    xg2 = x.to(dev()).requires_grad_(True)
    AG.posenc_dropout(xg2, pe, (0.0, 0, 0)).sum().backward()
    assert torch.equal(xg2.grad, torch.ones_like(xg2))


# ------------------------------------------------------------------------------------------------ encoder layer
def test_encoder_layer_grads_vs_torch():
    torch.manual_seed(0)
    B, T, d, H = 2, 70, 512, 8
    layer = torch.nn.TransformerEncoderLayer(d_model=d, nhead=H, dim_feedforward=1024, dropout=0.1, activation="gelu", batch_first=True).eval()
    load_named_weights(layer)
    x, dy = g("el_x", (B, T, d)), g("el_dy", (B, T, d))
    mask = torch.zeros(B, T, dtype=torch.bool)
    mask[1, 50:] = True
    ref = torch.nn.TransformerEncoderLayer(d_model=d, nhead=H, dim_feedforward=1024, dropout=0.1, activation="gelu", batch_first=True).double().eval()
    ref.load_state_dict({k: v.double() for k, v in layer.state_dict().items()})
    xr = x.double().requires_grad_(True)
    yr = ref(xr, src_key_padding_mask=mask)          # grad enabled -> torch takes its unfused (autograd) path
    (yr * dy.double()).sum().backward()
    layer = layer.to(dev())
    xg = x.to(dev()).requires_grad_(True)
    y = AG.encoder_layer(xg, layer, mask.to(dev()), H, (0.0, 0, 0))
    report("encoder layer forward", y, yr, 5e-5)
    (y * dy.to(dev())).sum().backward()
    rel("encoder layer dx", xg.grad, xr.grad, 1e-4)
    for (n, p_), (_, pr) in zip(layer.named_parameters(), ref.named_parameters()):
        rel(f"encoder layer d{n}", p_.grad, pr.grad, 1e-4)


# ------------------------------------------------------------------------------------------------ whole model
@pytest.fixture(scope="module")
def cmdm_train():
    model, diff = create_model_and_diffusion(cmdm_cfg(), device=dev())
    load_named_weights(model)
    model = model.to(dev())
    model.contact_encoder.requires_grad_(False)
    return model, diff


def _digest(gr):
    flat = gr.detach().reshape(-1).cpu()
    sample = flat if flat.numel() <= 2048 else flat[:: flat.numel() // 1024][:1024]
    return sample, flat.double().abs().sum().item()


def test_training_losses_backward_vs_reference_gradients(cmdm_train):
    """diffusion.training_losses(...)['loss'].mean().backward() (utils/training.py:140-152) on the HIP path vs the
    gradients the real reference produced for the same inputs (eval mode: dropout off)."""
    model, diff = cmdm_train
    model.eval()
    gf, gg = golden("cmdm_forward_N1024_L16"), golden("cmdm_training_grads")
    x0, tn = synth.gaussian("train_x0", (2, 16, 263)).to(dev()), synth.gaussian("train_noise", (2, 16, 263)).to(dev())
    kw = dict(c_text_feat=gf["text_feat"].to(dev()), c_cont_emb=gf["cont_emb"].to(dev()), x_mask=gf["x_mask"].to(dev()))
    model.zero_grad()
    terms = diff.training_losses(model, x0, gg["t"].to(dev()), model_kwargs=kw, noise=tn)
    report("training loss", terms["loss"], gg["loss"], 2e-5)
    terms["loss"].mean().backward()
    names = [k[2:] for k in gg if k.startswith("g/")]
    params = dict(model.named_parameters())
    worst = 0.0
    for n in names:
        assert params[n].grad is not None, n
        sample, abs_sum = _digest(params[n].grad)
        scale = max(gg["g/" + n].abs().max().item(), 1e-9)
        err = ((sample - gg["g/" + n]).abs().max() / scale).item()
        worst = max(worst, err)
        assert err <= 5e-4, f"{n}: scaled grad err {err:.3e}"
        assert abs(abs_sum - gg["s/" + n][1].item()) <= 5e-4 * gg["s/" + n][1].item() + 1e-9, n
    print(f"[parity] 72 trunk gradients vs the reference's backward: worst scaled err {worst:.3e}")
    for n, p_ in model.named_parameters():
        if n.startswith("contact_encoder."):
            assert p_.grad is None


def test_training_losses_backward_with_encoder_and_full_shapes(cmdm_train):
    """B=4, L=196 (T=326 tokens incl. 128 contact tokens from the HIP SceneMapEncoder) vs the oracle's autograd."""
    from oracle import diffusion_ref as df
    from oracle import shapes as sh
    from oracle import train_ref as tr
    model, diff = cmdm_train
    model.eval()
    B, L, N = 4, 196, 1024
    x0, tn = synth.gaussian("tf_x0", (B, L, 263)), synth.gaussian("tf_noise", (B, L, 263))
    text = synth.text_feature(B)
    cont = synth.gaussian("tf_cont", (B, 128, 256)) * 0.5
    x_mask = synth.frame_mask(B, L)
    t = torch.tensor([3, 250, 640, 999])
    sd = sh.weights(sh.cmdm())
    loss_ref, grads = tr.cmdm_loss_and_grads(sd, df.Schedule(1000), x0, t, tn, text, cont, x_mask)
    kw = dict(c_text_feat=text.to(dev()), c_cont_emb=cont.to(dev()), x_mask=x_mask.to(dev()))
    model.zero_grad()
    terms = diff.training_losses(model, x0.to(dev()), t.to(dev()), model_kwargs=kw, noise=tn.to(dev()))
    report("training loss (B=4, L=196)", terms["loss"], loss_ref, 5e-5)
    terms["loss"].mean().backward()
    params = dict(model.named_parameters())
    worst = 0.0
    for n, gr in grads.items():
        scale = max(gr.abs().max().item(), 1e-9)
        err = ((params[n].grad.cpu() - gr).abs().max() / scale).item()
        worst = max(worst, err)
        assert err <= 5e-4, f"{n}: scaled grad err {err:.3e}"
    print(f"[parity] {len(grads)} trunk gradients at B=4, L=196 vs oracle autograd: worst scaled err {worst:.3e}")


def test_train_mode_dropout_is_reproducible_and_trains(cmdm_train):
    """Train mode (all dropouts on): same seed -> identical loss / grads; a few fused-AdamW steps reduce the loss."""
    model, diff = cmdm_train
    gf = golden("cmdm_forward_N1024_L16")
    x0 = synth.gaussian("train_x0", (2, 16, 263)).to(dev())
    kw = dict(c_text_feat=gf["text_feat"].to(dev()), c_cont_emb=gf["cont_emb"].to(dev()), x_mask=gf["x_mask"].to(dev()))
    t = torch.tensor([17, 803], device=dev())
    state0 = {k: v.clone() for k, v in model.state_dict().items()}
    model.train()

    def run():
        torch.manual_seed(5)
        model._drop_calls, diff._loss_calls = 0, 0
        model.zero_grad()
        terms = diff.training_losses(model, x0, t, model_kwargs=kw)
        terms["loss"].mean().backward()
        return terms["loss"].detach().clone(), model.motion_layer.weight.grad.clone()
    l1, g1 = run()
    l2, g2 = run()
    assert torch.equal(l1, l2) and torch.equal(g1, g2)
    model.eval()
    with torch.no_grad():
        diff._loss_calls = 0
        l_eval = diff.training_losses(model, x0, t, model_kwargs=kw)["loss"]
    assert not torch.allclose(l1, l_eval), "dropout had no effect in train mode"
    model.train()
    opt_state, losses = {}, []
    params = [p_ for p_ in model.parameters() if p_.requires_grad]
    for it in range(8):
        model.zero_grad()
        diff._loss_calls = 0                      # same noise every iteration: the loss must go down
        terms = diff.training_losses(model, x0, t, model_kwargs=kw)
        terms["loss"].mean().backward()
        losses.append(terms["loss"].mean().item())
        AG.adamw_step(params, opt_state, lr=1e-4)
    print("[train] losses:", [round(v, 4) for v in losses])
    assert losses[-1] < losses[0]
    model.load_state_dict(state0)
    model.eval()


def test_adamw_matches_torch():
    p0, gr = g("aw_p", (1000,)), g("aw_g", (1000,))
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    mine = torch.nn.Parameter(p0.clone().to(dev()))
    st = {}
    for i in range(3):
        ref.grad = gr * (i + 1)
        opt.step()
        mine.grad = (gr * (i + 1)).to(dev())
        AG.adamw_step([mine], st, lr=1e-3, weight_decay=0.01)
    report("AdamW 3 steps", mine.data, ref.data, 1e-6)


# ------------------------------------------------------------------------------------------------ point-cloud branch (train-mode BatchNorm)
from afm import autograd_points as AP      # noqa: E402
from afm import scene as S                 # noqa: E402


@pytest.mark.parametrize("rows,C,relu,res", [(20000, 32, True, False), (4097, 3, True, False), (3000, 256, False, True), (50, 512, True, True)])
def test_batch_norm_train_forward_backward(rows, C, relu, res):
    x, dy = g("bn_x", (rows, C)) * 1.5 + 0.3, g("bn_dy", (rows, C))
    r = g("bn_r", (rows, C)) if res else None
    bn = torch.nn.BatchNorm1d(C)
    with torch.no_grad():
        bn.weight.copy_(1 + 0.1 * g("bn_w", (C,))); bn.bias.copy_(0.1 * g("bn_b", (C,)))
    ref = torch.nn.BatchNorm1d(C).double()
    ref.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in bn.state_dict().items()})
    ref.train()
    xr = x.double().requires_grad_(True)
    rr = None if r is None else r.double().requires_grad_(True)
    yr = ref(xr) + (0 if rr is None else rr)
    yr = torch.relu(yr) if relu else yr
    (yr * dy.double()).sum().backward()
    bn = bn.to(dev()).train()
    xg = x.to(dev()).requires_grad_(True)
    rg = None if r is None else r.to(dev()).requires_grad_(True)
    y = AP.batch_norm(xg, bn, relu=relu, residual=rg)
    report("BN train forward", y, yr, 2e-5)
    (y * dy.to(dev())).sum().backward()
    rel("BN dx", xg.grad, xr.grad, 5e-5)
    rel("BN dgamma", bn.weight.grad, ref.weight.grad, 5e-5)
    rel("BN dbeta", bn.bias.grad, ref.bias.grad, 5e-5)
    if res:
        rel("BN dresidual", rg.grad, rr.grad, 1e-6)
    report("BN running_mean", bn.running_mean, ref.running_mean, 1e-5)
    report("BN running_var", bn.running_var, ref.running_var, 1e-5)
    assert int(bn.num_batches_tracked) == 1
    # frozen BatchNorm (eval): running statistics, gradients still flow
    bn.eval(); ref.eval()
    xg.grad = None; bn.weight.grad = None; xr.grad = None; ref.weight.grad = None
    y2 = AP.batch_norm(xg, bn, relu=relu, residual=None if rg is None else rg.detach())
    y2r = ref(xr) + (0 if rr is None else rr.detach())
    y2r = torch.relu(y2r) if relu else y2r
    report("BN eval forward", y2, y2r, 2e-5)
    (y2 * dy.to(dev())).sum().backward(); (y2r * dy.double()).sum().backward()
    rel("BN eval dx", xg.grad, xr.grad, 5e-5)
    rel("BN eval dgamma", bn.weight.grad, ref.weight.grad, 5e-5)


def test_scatter_adds_are_deterministic_segmented_sums():
    """Round 6: the backward of every row gather / neighbour grouping and of the 3-NN interpolation is a segmented sum over
    the inverse of the index list (afm_scatter_plan + afm_segment_sum_rows): no f32 atomics, so repeated runs produce the SAME bits - with
    heavily shared destinations (a coarse point that is the neighbour of hundreds of fine points), empty destinations, and an index list in
    arbitrary order.  Checked against float64 index_add_, against the atomic entry points they replace (same numbers, other order), and
    bit for bit across repeats; the plan itself against a stable sort of the index list."""
    import ctypes as C
    lib = ffi.load()
    gen = torch.Generator().manual_seed(7)
    n_dst, m, k, Cn = 700, 3000, 16, 40
    idx = torch.randint(0, n_dst - 50, (m * k,), generator=gen, dtype=torch.int32)      # the last 50 destinations stay empty
    idx[: m * k // 4] = idx[: m * k // 4] % 5                                            # five destinations with thousands of entries each
    src = g("ss_src", (m * k, 3 + Cn))
    want = torch.zeros(n_dst, Cn, dtype=torch.float64).index_add_(0, idx.long(), src[:, 3:].double())
    idx_d, src_d = idx.to(dev()), src.to(dev())
    plan = AP.scatter_plan(idx_d, n_dst)
    assert AP.scatter_plan(idx_d, n_dst) is plan                                         # cached per index tensor
    off, ent = plan[: n_dst + 1].cpu().long(), plan[n_dst + 1: n_dst + 1 + m * k].cpu().long()
    order = torch.sort(idx.long(), stable=True).indices
    assert torch.equal(ent, order) and torch.equal(off[1:] - off[:-1], torch.bincount(idx.long(), minlength=n_dst))
    outs = [AP._segment_sum(src_d, 3, idx_d, n_dst, Cn) for _ in range(3)]
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    rel("segmented scatter-add vs float64 index_add_", outs[0], want, 2e-6)
    atomic = torch.zeros(n_dst, Cn, device=dev())
    ffi.check(lib.afm_scatter_add_rows(src_d.data_ptr(), 3 + Cn, 3, idx_d.data_ptr(), atomic.data_ptr(), m * k, Cn, None), "afm_scatter_add_rows")
    rel("segmented vs atomic scatter-add", outs[0], atomic.double(), 2e-6)
    # interpolation backward: k = 3 weighted entries per fine point
    nf, mc, c3 = 5000, 300, 32
    idx3 = torch.randint(0, mc, (nf, 3), generator=gen, dtype=torch.int32).to(dev())
    d2 = (torch.rand(nf, 3, generator=gen) + 1e-3).to(dev())
    dout = g("ss_dout", (nf, c3)).to(dev())
    a = [AP._segment_sum(dout, 0, idx3, mc, c3, row_div=3, d2=d2) for _ in range(2)]
    assert torch.equal(a[0], a[1])
    atom = torch.empty(mc, c3, device=dev())
    ffi.check(lib.afm_interpolate_bwd(dout.data_ptr(), idx3.data_ptr(), d2.data_ptr(), atom.data_ptr(), nf, mc, c3, 3, None), "afm_interpolate_bwd")
    rel("segmented vs atomic interpolation backward", a[0], atom.double(), 2e-6)
    w = 1.0 / (d2.double().sqrt() + 1e-8)
    w = w / w.sum(1, keepdim=True)
    want3 = torch.zeros(mc, c3, dtype=torch.float64, device=dev())
    for j in range(3):
        want3.index_add_(0, idx3[:, j].long(), dout.double() * w[:, j:j + 1])
    rel("interpolation backward vs float64", a[0], want3, 2e-6)
    # broadcast_rows: the sum over a sample's n rows (afm_group_sum), repeatable
    rows = g("ss_rows", (4, 24)).to(dev()).requires_grad_(True)
    dyb = g("ss_dyb", (4 * 513, 24)).to(dev())
    grads = []
    for _ in range(2):
        rows.grad = None
        (AP.broadcast_rows(rows, 513) * dyb).sum().backward()
        grads.append(rows.grad.clone())
    assert torch.equal(grads[0], grads[1])
    rel("broadcast_rows backward", grads[0], dyb.double().view(4, 513, 24).sum(1), 2e-6)


def test_two_identical_training_steps_of_the_scene_branch_are_bit_identical():
    """DESIGN section 2 lists bit-identity as an invariant; with the segmented scatter-adds it holds for the TRAINING step of the scene branch
    too: SceneMapEncoder in train mode (batch-statistics BatchNorm, kNN grouping backward, vector attention), forward
    + backward twice from the same weights on the same data - every gradient tensor and the output equal bit for bit."""
    from afm import scene as S
    from gpu_util import load_named_weights
    xyz, con = synth.scene_cloud(4, 1024, seed=23).to(dev()), synth.contact_map(4, 1024, seed=23).to(dev())
    dy = g("det_dy", (4, 16, 256)).to(dev())
    runs = []
    for _ in range(2):
        enc = S.SceneMapEncoder(point_feat_dim=6, planes=[32, 64, 128, 256], blocks=[2, 2, 2, 2], num_points=1024)
        load_named_weights(enc)
        enc = enc.to(dev()).train()
        out = enc(xyz, con)
        (out * dy).sum().backward()
        runs.append((out.detach().clone(), {n: p.grad.clone() for n, p in enc.named_parameters() if p.grad is not None}))
    assert torch.equal(runs[0][0], runs[1][0])
    assert runs[0][1].keys() == runs[1][1].keys() and len(runs[0][1]) > 100
    diff = [n for n in runs[0][1] if not torch.equal(runs[0][1][n], runs[1][1][n])]
    assert not diff, f"{len(diff)} gradient tensors differ between two identical steps: {diff[:4]}"


def test_gather_group_and_max_ops():
    n, m, k, C = 500, 120, 16, 24
    feat, xyz = g("gg_f", (n, C)), g("gg_p", (n, 3))
    idx = torch.randint(0, n, (m, k), generator=torch.Generator().manual_seed(1), dtype=torch.int32)
    new_xyz = xyz[idx[:, 0].long()]
    dy = g("gg_dy", (m * k, 3 + C))
    fr = feat.double().requires_grad_(True)
    gr = torch.cat([xyz.double()[idx.long().view(-1)] - new_xyz.double().repeat_interleave(k, 0), fr[idx.long().view(-1)]], 1)
    mx_r = gr.view(m, k, 3 + C).max(1).values
    ((gr * dy.double()).sum() + (mx_r ** 2).sum()).backward()
    fg = feat.to(dev()).requires_grad_(True)
    gg = AP.group_points(xyz.to(dev()), new_xyz.to(dev()), fg, idx.to(dev()).view(-1), k)
    report("group_points", gg, gr, 1e-6)
    mx = AP.group_max(gg, k)
    report("group_max", mx, mx_r, 1e-6)
    ((gg * dy.to(dev())).sum() + (mx ** 2).sum()).backward()
    rel("group_points/max dfeat (segmented scatter-add)", fg.grad, fr.grad, 1e-5)
    f2 = feat.to(dev()).requires_grad_(True)
    (AP.gather(f2, idx.to(dev()).view(-1)) * dy[:, 3:].to(dev())).sum().backward()
    f2r = feat.double().requires_grad_(True)
    (f2r[idx.long().view(-1)] * dy[:, 3:].double()).sum().backward()
    rel("gather backward", f2.grad, f2r.grad, 1e-5)


@pytest.mark.parametrize("m,k,C,share", [(300, 16, 64, 8), (200, 16, 32, 1), (130, 8, 64, 2), (77, 16, 512, 4)])
def test_vector_attention_glue_ops(m, k, C, share):
    """share_planes = 8 is the reference's; small share_planes make the fused kernels' weight slab as large as their value slab - their
    workgroup shrinks until its LDS fits 64 KB (share_planes = 1, k = 16, C = 32 asked for 100 KB and failed to launch)."""
    kg, pr, vg = g("va_kg", (m * k, C)), g("va_pr", (m * k, C)), g("va_vg", (m * k, C))
    q, w2 = g("va_q", (m, C)), g("va_w2", (m * k, C // share))
    d0, d1 = g("va_d0", (m * k, C)), g("va_d1", (m, C))
    T = lambda t: t.double().requires_grad_(True)
    kgr, prr, vgr, qr, w2r = T(kg), T(pr), T(vg), T(q), T(w2)
    w0r = kgr.view(m, k, C) - qr[:, None, :] + prr.view(m, k, C)
    swr = torch.softmax(w2r.view(m, k, C // share), 1)
    outr = ((vgr + prr).view(m, k, share, C // share) * swr[:, :, None, :]).sum(1).view(m, C)
    ((w0r.reshape(m * k, C) * d0.double()).sum() + (outr * d1.double()).sum()).backward()
    G = lambda t: t.to(dev()).requires_grad_(True)
    kgg, prg, vgg, qg, w2g = G(kg), G(pr), G(vg), G(q), G(w2)
    w0 = AP.pt_w0(kgg, qg, prg, k)
    out = AP.pt_aggregate(vgg, prg, w2g, k, share)
    report("pt_w0", w0, w0r.reshape(m * k, C), 1e-6)
    report("pt_aggregate", out, outr, 1e-5)
    ((w0 * d0.to(dev())).sum() + (out * d1.to(dev())).sum()).backward()
    for name, a, b in (("dkg", kgg, kgr), ("dpr", prg, prr), ("dvg", vgg, vgr), ("dq", qg, qr), ("dw2", w2g, w2r)):
        rel("vector attention " + name, a.grad, b.grad, 2e-5)


def _zero_grad_name(name):
    return "transformer2." in name and name.split("transformer2.")[1] in ("linear_q.bias", "linear_k.bias", "linear_v.bias", "linear_p.0.bias",
                                                                           "linear_p.3.bias", "linear_w.2.bias", "linear_w.5.bias")


def test_scene_encoder_train_mode_vs_reference():
    """SceneMapEncoder under .train(): forward, every parameter gradient and the BatchNorm running statistics vs the real
    reference (tests/golden/scene_encoder_train_N1024.npz, FPS / kNN from our own operators on both sides)."""
    gi, gt = golden("scene_map_encoder_N1024"), golden("scene_encoder_train_N1024")
    enc = S.SceneMapEncoder(point_feat_dim=6, planes=[32, 64, 128, 256], blocks=[2, 2, 2, 2], num_points=gi["xyz"].shape[1])
    load_named_weights(enc)
    enc = enc.to(dev()).train()
    out = enc(gi["xyz"].to(dev()), gi["contact"].to(dev()))
    report("SceneMapEncoder train-mode forward", out, gt["out"], 2e-4)
    dy = synth.gaussian("scene_train_dy", tuple(gt["out"].shape)).to(dev())
    (out * dy).sum().backward()
    worst = 0.0
    params = dict(enc.named_parameters())
    names = [k[2:] for k in gt if k.startswith("g/")]
    assert len(names) == 124
    for n in names:
        sample, _ = _digest(params[n].grad)
        if _zero_grad_name(n):
            assert sample.abs().max().item() < 5e-3, n
            continue
        scale = max(gt["g/" + n].abs().max().item(), 1e-3)
        err = ((sample - gt["g/" + n]).abs().max() / scale).item()
        worst = max(worst, err)
        assert err <= 2e-3, f"{n}: scaled grad err {err:.3e}"
    print(f"[parity] 124 scene-encoder gradients (train-mode BatchNorm) vs the reference: worst scaled err {worst:.3e}")
    mods = dict(enc.named_modules())
    for bn in ("enc1.0.bn", "enc2.1.transformer2.linear_w.0", "enc4.1.bn3"):
        report(f"running_mean {bn}", mods[bn].running_mean, gt["rm/" + bn], 1e-5)
        report(f"running_var {bn}", mods[bn].running_var, gt["rv/" + bn], 1e-4)


def test_cmdm_trains_end_to_end_including_scene_encoder():
    """Whole CMDM (scene encoder not frozen) in train mode: losses fall over a few AdamW steps, every parameter gets a
    gradient - the reference's TrainLoop step (utils/training.py:138-155)."""
    model, diff = create_model_and_diffusion(cmdm_cfg(num_points=1024), device=dev())
    load_named_weights(model)
    model = model.to(dev()).train()
    gf = golden("cmdm_forward_N1024_L16")
    x0 = synth.gaussian("train_x0", (2, 16, 263)).to(dev())
    kw = dict(c_text_feat=gf["text_feat"].to(dev()), c_pc_xyz=gf["xyz"].to(dev()), c_pc_contact=gf["contact"].to(dev()),
              x_mask=gf["x_mask"].to(dev()))
    t = torch.tensor([17, 803], device=dev())
    params = [p_ for n_, p_ in model.named_parameters() if p_.requires_grad and not n_.startswith("text_model")]
    opt = torch.optim.AdamW(params, lr=1e-4)          # the reference's own optimiser works on our parameters / gradients
    losses = []
    for it in range(6):
        opt.zero_grad()
        diff._loss_calls = 0
        terms = diff.training_losses(model, x0, t, model_kwargs=kw)
        terms["loss"].mean().backward()
        losses.append(terms["loss"].mean().item())
        opt.step()
    assert all(p_.grad is not None for p_ in params)
    assert all(torch.isfinite(p_.grad).all() for p_ in params)
    print("[train] full-model losses:", [round(v, 4) for v in losses])
    assert losses[-1] < losses[0]


# ------------------------------------------------------------------------------------------------ CDM (Perceiver) training
def _xattn_ref(q, k, v, H, keep=None):
    B, Tq, C = q.shape
    sp = lambda z: z.view(z.shape[0], z.shape[1], H, C // H).transpose(1, 2)
    qh, kh, vh = sp(q), sp(k), sp(v)
    a = torch.softmax(qh @ kh.transpose(-1, -2) / (C // H) ** 0.5, -1)
    if keep is not None:
        a = a * keep
    return (a @ vh).transpose(1, 2).reshape(B, Tq, C), a


@pytest.mark.parametrize("B,N,H,C", [(2, 8192, 8, 512), (3, 300, 8, 512), (2, 1000, 8, 256)])
def test_few_query_attention_fwd_bwd(B, N, H, C):
    q, k, v, dO = g("fq_q", (B, 2, C)), g("fq_k", (B, N, C)) * 0.5, g("fq_v", (B, N, C)), g("fq_do", (B, 2, C))
    T = lambda t: t.double().requires_grad_(True)
    qr, kr, vr = T(q), T(k), T(v)
    o_ref, _ = _xattn_ref(qr, kr, vr, H)
    (o_ref * dO.double()).sum().backward()
    G = lambda t: t.to(dev()).requires_grad_(True)
    qg, kg, vg = G(q), G(k), G(v)
    o = AG.few_query_attention(qg, kg, vg, H)
    report("few-query attention out", o, o_ref, 2e-5)
    (o * dO.to(dev())).sum().backward()
    rel("few-query dQ", qg.grad, qr.grad, 5e-5)
    rel("few-query dK", kg.grad, kr.grad, 5e-5)
    rel("few-query dV", vg.grad, vr.grad, 5e-5)


@pytest.mark.parametrize("B,N,H,C", [(2, 8192, 8, 256), (3, 301, 8, 256), (2, 500, 8, 512)])
def test_few_key_attention_fwd_bwd(B, N, H, C):
    q, k, v, dO = g("fk_q", (B, N, C)), g("fk_k", (B, 2, C)), g("fk_v", (B, 2, C)), g("fk_do", (B, N, C))
    T = lambda t: t.double().requires_grad_(True)
    qr, kr, vr = T(q), T(k), T(v)
    o_ref, _ = _xattn_ref(qr, kr, vr, H)
    (o_ref * dO.double()).sum().backward()
    G = lambda t: t.to(dev()).requires_grad_(True)
    qg, kg, vg = G(q), G(k), G(v)
    o = AG.few_key_attention(qg, kg, vg, H)
    report("few-key attention out", o, o_ref, 2e-5)
    (o * dO.to(dev())).sum().backward()
    rel("few-key dQ", qg.grad, qr.grad, 5e-5)
    rel("few-key dK", kg.grad, kr.grad, 5e-5)
    rel("few-key dV", vg.grad, vr.grad, 5e-5)


def test_perceiver_attention_dropout_consistent():
    """Same trick as for the encoder layers: a one-hot V exposes the dropped probabilities, i.e. the keep mask; forward and
    backward with that mask must equal float64."""
    B, H, p = 2, 8, 0.3
    # few-query: N = 64 keys, C = 512 (dh = 64): V_h = identity over the keys
    N, C = 64, 512
    drop = (p, 4242, 3)
    q, k, dO = g("pd_q", (B, 2, C)), g("pd_k", (B, N, C)) * 0.5, g("pd_do", (B, 2, C))
    eye = torch.eye(64).repeat(1, H).unsqueeze(0).expand(B, N, C).contiguous()
    pd = AG.few_query_attention(q.to(dev()), k.to(dev()), eye.to(dev()), H, drop).cpu().view(B, 2, H, 64).transpose(1, 2)     # [B,H,2,N]
    p0 = AG.few_query_attention(q.to(dev()), k.to(dev()), eye.to(dev()), H).cpu().view(B, 2, H, 64).transpose(1, 2)
    keep = torch.where(pd > 0, torch.full_like(pd, 1 / (1 - p)), torch.zeros_like(pd))
    report("few-query dropped P == P * keep", pd, p0 * keep, 1e-6)
    v = g("pd_v", (B, N, C))
    T = lambda t: t.double().requires_grad_(True)
    qr, kr, vr = T(q), T(k), T(v)
    o_ref, _ = _xattn_ref(qr, kr, vr, H, keep.double())
    (o_ref * dO.double()).sum().backward()
    G = lambda t: t.to(dev()).requires_grad_(True)
    qg, kg, vg = G(q), G(k), G(v)
    o = AG.few_query_attention(qg, kg, vg, H, drop)
    report("few-query out with dropout", o, o_ref, 2e-5)
    (o * dO.to(dev())).sum().backward()
    for n_, a_, b_ in (("dQ", qg, qr), ("dK", kg, kr), ("dV", vg, vr)):
        rel("few-query dropout " + n_, a_.grad, b_.grad, 5e-5)
    # few-key: 2 keys, C = 256 (dh = 32): V rows = two one-hot vectors per head expose both probabilities
    N, C = 500, 256
    q, k, dO = g("pk_q", (B, N, C)), g("pk_k", (B, 2, C)), g("pk_do", (B, N, C))
    onehot = torch.zeros(B, 2, C)
    for h in range(H):
        onehot[:, 0, h * 32] = 1.0; onehot[:, 1, h * 32 + 1] = 1.0
    pd = AG.few_key_attention(q.to(dev()), k.to(dev()), onehot.to(dev()), H, drop).cpu().view(B, N, H, 32)[..., :2].permute(0, 2, 1, 3)   # [B,H,N,2]
    p0 = AG.few_key_attention(q.to(dev()), k.to(dev()), onehot.to(dev()), H).cpu().view(B, N, H, 32)[..., :2].permute(0, 2, 1, 3)
    keep = torch.where(pd > 0, torch.full_like(pd, 1 / (1 - p)), torch.zeros_like(pd))
    report("few-key dropped P == P * keep", pd, p0 * keep, 1e-6)
    assert abs((keep > 0).float().mean().item() - (1 - p)) < 0.03
    v = g("pk_v", (B, 2, C))
    qr, kr, vr = T(q), T(k), T(v)
    o_ref, _ = _xattn_ref(qr, kr, vr, H, keep.double())
    (o_ref * dO.double()).sum().backward()
    qg, kg, vg = G(q), G(k), G(v)
    o = AG.few_key_attention(qg, kg, vg, H, drop)
    report("few-key out with dropout", o, o_ref, 2e-5)
    (o * dO.to(dev())).sum().backward()
    for n_, a_, b_ in (("dQ", qg, qr), ("dK", kg, kr), ("dV", vg, vr)):
        rel("few-key dropout " + n_, a_.grad, b_.grad, 5e-5)


def _cdm_model():
    from afm.config import load_config
    cfg = load_config("text_to_motion_contact_gen", "cdm", ["model.arch=Perceiver", "model.scene_model.use_scene_model=False",
                                                           "model.input_feats=6", "model.text_model.max_length=20", "diffusion.steps=500"])
    model, diff = create_model_and_diffusion(cfg, device=dev())
    load_named_weights(model)
    return model.to(dev()), diff


def test_cdm_training_losses_backward_vs_reference_gradients():
    """CDM (Perceiver): training_losses(...)['loss'].mean().backward() on the HIP path vs the real reference's gradients."""
    model, diff = _cdm_model()
    model.eval()
    gf, gg = golden("cdm_forward_N256"), golden("cdm_training_grads")
    x0, tn = synth.gaussian("cdm_train_x0", (2, 256, 6)).to(dev()), synth.gaussian("cdm_train_noise", (2, 256, 6)).to(dev())
    kw = dict(c_text_feat=gf["text_feat"].to(dev()), c_pc_xyz=gf["xyz"].to(dev()))
    model.zero_grad()
    terms = diff.training_losses(model, x0, gg["t"].to(dev()), model_kwargs=kw, noise=tn)
    report("CDM training loss", terms["loss"], gg["loss"], 2e-5)
    terms["loss"].mean().backward()
    params = dict(model.named_parameters())
    names = [k[2:] for k in gg if k.startswith("g/")]
    assert len(names) == 82
    worst = 0.0
    for n in names:
        sample, _ = _digest(params[n].grad)
        scale = max(gg["g/" + n].abs().max().item(), 1e-4)
        err = ((sample - gg["g/" + n]).abs().max() / scale).item()
        worst = max(worst, err)
        assert err <= 1e-3, f"{n}: scaled grad err {err:.3e}"
    print(f"[parity] 82 CDM gradients vs the reference's backward: worst scaled err {worst:.3e}")


@pytest.mark.parametrize("arch,tag,train,count", [("PointTrans", "cdm_pointtrans_training_grads", True, 278),
                                                   ("PointTransV2", "cdm_pointtransv2_training_grads", False, 302)])
def test_cdm_pointtrans_training_vs_reference_gradients(arch, tag, train, count):
    """`model.arch=PointTrans` (train() mode: every BatchNorm of the U-Net on batch statistics) and `PointTransV2` (eval() mode: its
    bottleneck encoder layer carries a dropout torch draws from its own generator): training_losses(...)['loss'].mean().backward() on
    the HIP operator graph vs the gradients of the REAL reference (oracle/make_goldens_train.py --pointtrans_train)."""
    from afm.config import load_config
    cfg = load_config("text_to_motion_contact_gen", "cdm", ["model.input_feats=6", "model.scene_model.use_scene_model=False", f"model.arch={arch}",
                                                           "task.dataset.num_points=1024", "model.text_model.max_length=20", "diffusion.steps=500"])
    model, diff = create_model_and_diffusion(cfg, device=dev())
    load_named_weights(model)
    model = model.to(dev())
    model.train() if train else model.eval()
    gg, gf = golden(tag), golden("cdm_forward_N256")
    x0, tn = synth.gaussian("cdm_pt_train_x0", (2, 1024, 6)).to(dev()), synth.gaussian("cdm_pt_train_noise", (2, 1024, 6)).to(dev())
    kw = dict(c_text_feat=gf["text_feat"].to(dev()), c_pc_xyz=synth.scene_cloud(2, 1024, seed=16).to(dev()))
    model.zero_grad()
    terms = diff.training_losses(model, x0, gg["t"].to(dev()), model_kwargs=kw, noise=tn)
    report(f"CDM {arch} training loss", terms["loss"], gg["loss"], 2e-4)
    terms["loss"].mean().backward()
    params = dict(model.named_parameters())
    names = [k[2:] for k in gg if k.startswith("g/")]
    assert len(names) == count
    # train mode: ~40 batch-statistics BatchNorms in series (down to 32 rows per batch at the bottleneck) make some gradients ill-conditioned
    # in float32 - the reference's own f32 and f64 runs differ by up to 2.8e-2 (scaled) on them - so the train-mode golden carries the
    # reference's FLOAT64 gradients (g64/) next to its float32 ones and the HIP result is held to the float64 ones
    errs, own, tol = [], [], 2e-3
    for n in names:
        assert params[n].grad is not None, n
        sample, _ = _digest(params[n].grad)
        if train and _zero_grad_name(n):            # a bias in front of a batch-statistics BatchNorm: its true gradient is zero
            assert sample.abs().max().item() < 5e-3, n
            continue
        want = gg["g64/" + n] if train else gg["g/" + n]
        scale = max(want.abs().max().item(), 1e-3)
        errs.append((((sample - want).abs().max() / scale).item(), n))
        if train:                                     # what the reference's own float32 run is away from its float64 run on this tensor
            own.append(((gg["g/" + n] - want).abs().max() / scale).item())
    errs.sort(reverse=True)
    p90 = errs[len(errs) // 10][0]
    print(f"[parity] {count} CDM {arch} gradients vs the reference's {'float64 ' if train else ''}backward: worst scaled err {errs[0][0]:.2e} "
          f"({errs[0][1]}), 90th percentile {p90:.2e}" + (f"; the reference's own f32 run vs its f64 run: worst {max(own):.2e}, "
                                                           f"90th percentile {sorted(own, reverse=True)[len(own) // 10]:.2e}" if train else ""))
    if train:       # the same noise class as the reference's own float32 run: no tensor further from float64 than its worst one, the bulk tight
        assert errs[0][0] <= max(own), f"{errs[0][1]}: scaled grad err {errs[0][0]:.3e} vs float64 exceeds the reference's own worst f32 distance {max(own):.3e}"
        assert p90 <= 3e-3, f"90th percentile of the scaled gradient errors {p90:.3e}"
    else:
        assert errs[0][0] <= tol, f"{errs[0][1]}: scaled grad err {errs[0][0]:.3e}"
    if train:
        mods = dict(model.named_modules())
        for bn in ("contact_model.enc1.0.bn", "contact_model.dec2.0.linear2.1", "contact_model.ctx.1"):
            report(f"running_mean {bn}", mods[bn].running_mean, gg["rm/" + bn], 2e-5)
            report(f"running_var {bn}", mods[bn].running_var, gg["rv/" + bn], 2e-4)


@pytest.mark.parametrize("B,T,n,heads", [(2, 37, 300, 2), (3, 198, 1024, 8), (1, 5, 33, 1)])
def test_decoder_layer_fwd_bwd_vs_torch_float64(B, T, n, heads):
    """AG.decoder_layer (self-attention with a key padding mask, cross-attention over a masked memory with Tq != Tk - afm_mha_cross_fwd_train /
    afm_mha_cross_bwd -, feed-forward; post-LN, GELU) against nn.TransformerDecoderLayer in float64 on the CPU: output and every gradient."""
    d = 64 * heads
    torch.manual_seed(5)
    layer = torch.nn.TransformerDecoderLayer(d_model=d, nhead=heads, dim_feedforward=2 * d, dropout=0.0, activation="gelu", batch_first=True)
    for name, p_ in layer.named_parameters():
        p_.data = synth.gaussian("dl_" + name, tuple(p_.shape)) * (2.0 / d ** 0.5 if p_.dim() > 1 else 0.1) + (1.0 if "norm" in name and "weight" in name else 0.0)
    x, mem, dy = g("dl_x", (B, T, d)), g("dl_mem", (B, n, d)), g("dl_dy", (B, T, d))
    km = torch.zeros(B, T, dtype=torch.bool); km[0, T - T // 3:] = True
    mm = torch.zeros(B, n, dtype=torch.bool); mm[B - 1, n // 2:] = True
    ref = torch.nn.TransformerDecoderLayer(d_model=d, nhead=heads, dim_feedforward=2 * d, dropout=0.0, activation="gelu", batch_first=True).double()
    ref.load_state_dict({k: v.double() for k, v in layer.state_dict().items()})
    xr, mr = x.double().requires_grad_(True), mem.double().requires_grad_(True)
    out_r = ref(xr, mr, tgt_key_padding_mask=km, memory_key_padding_mask=mm)
    valid = ~km                                        # rows of padded queries are not compared (nothing reads them)
    (out_r * dy.double() * valid[..., None]).sum().backward()
    layer = layer.to(dev())
    xg, mg = x.to(dev()).requires_grad_(True), mem.to(dev()).requires_grad_(True)
    out = AG.decoder_layer(xg, mg, layer, km.to(dev()), mm.to(dev()), heads, (0.0, 0, 0))
    report(f"decoder layer forward B={B} T={T} n={n}", out.detach().cpu()[valid], out_r.detach()[valid].float(), 1e-4)
    (out * (dy * valid[..., None]).to(dev())).sum().backward()
    rel("decoder layer dx", xg.grad.cpu()[valid], xr.grad[valid], 1e-4)
    rel("decoder layer dmem", mg.grad, mr.grad, 1e-4)
    pr = dict(ref.named_parameters())
    for name, p_ in layer.named_parameters():
        rel(f"decoder layer d{name}", p_.grad, pr[name].grad, 1e-4)


def test_cross_attention_dropout_mask_is_the_same_in_forward_and_backward():
    """afm_mha_cross_fwd_train / afm_mha_cross_bwd with attention dropout: V = identity over Tk = dh = 64 keys makes the output row the dropped
    probability row itself, which recovers the keep mask; forward and all three gradients are then checked against float64 with THAT mask."""
    lib = ffi.load()
    B, Tq, Tk, H, p = 2, 40, 64, 1, 0.25
    q, k, dO = g("xd_q", (B, Tq, 64)), g("xd_k", (B, Tk, 64)), g("xd_do", (B, Tq, 64))
    v = torch.eye(64).expand(B, Tk, 64).contiguous()
    kv = torch.cat((k, v), -1).contiguous()
    qd, kvd = q.to(dev()), kv.to(dev())
    out, lse = torch.empty(B, Tq, 64, device=dev()), torch.empty(B * H * Tq, device=dev())
    ffi.check(lib.afm_mha_cross_fwd_train(qd.data_ptr(), kvd.data_ptr(), None, out.data_ptr(), lse.data_ptr(), B, Tq, Tk, H, 64, p, 1234, 7,
                                          ffi.stream_of(qd)), "afm_mha_cross_fwd_train")
    P = torch.softmax(q.double() @ k.double().transpose(1, 2) / 8.0, -1)
    M = torch.where(out.cpu().double() > 0.5 * P, torch.full_like(P, 1.0 / (1.0 - p)), torch.zeros_like(P))
    kept = (M > 0).double().mean().item()
    assert abs(kept - (1 - p)) < 0.05, kept
    report("cross-attention dropped probabilities", out, (P * M).float(), 2e-6)
    dq, dkv = torch.empty_like(qd), torch.empty_like(kvd)
    ws = torch.empty(B * H * Tq, device=dev())
    dOd = dO.to(dev())
    ffi.check(lib.afm_mha_cross_bwd(qd.data_ptr(), kvd.data_ptr(), None, out.data_ptr(), dOd.data_ptr(), lse.data_ptr(), dq.data_ptr(), dkv.data_ptr(),
                                    B, Tq, Tk, H, 64, p, 1234, 7, ws.data_ptr(), ws.numel() * 4, ffi.stream_of(qd)), "afm_mha_cross_bwd")
    dOr, Vr = dO.double(), v.double()
    dV = (P * M).transpose(1, 2) @ dOr
    dP = (dOr @ Vr.transpose(1, 2)) * M
    dS = P * (dP - (dP * P).sum(-1, keepdim=True))
    rel("cross-attention dropout dQ", dq, dS @ k.double() / 8.0, 2e-5)
    rel("cross-attention dropout dK", dkv[..., :64], dS.transpose(1, 2) @ q.double() / 8.0, 2e-5)
    rel("cross-attention dropout dV", dkv[..., 64:], dV, 2e-5)


def test_cmdm_trans_dec_training_vs_reference_gradients():
    """`model.arch=trans_dec` under autograd: training_losses(...)['loss'].mean().backward() through the multi-scale SceneMapEncoderDecoder,
    kv_mappling, the self-attention stacks and the decoder layers vs the 424 gradients of the REAL reference (eval mode: dropout off,
    BatchNorm on running statistics; oracle/make_goldens_train.py --trans_dec_train)."""
    cfg = cmdm_cfg()
    cfg.model.arch = "trans_dec"
    model, diff = create_model_and_diffusion(cfg, device=dev())
    load_named_weights(model)
    model = model.to(dev()).eval()
    gf, gg = golden("cmdm_forward_N1024_L16"), golden("cmdm_trans_dec_training_grads")
    x0, tn = synth.gaussian("train_x0", (2, 16, 263)).to(dev()), synth.gaussian("train_noise", (2, 16, 263)).to(dev())
    kw = dict(c_text_feat=gf["text_feat"].to(dev()), c_pc_xyz=gf["xyz"].to(dev()), c_pc_contact=gf["contact"].to(dev()), x_mask=gf["x_mask"].to(dev()))
    model.zero_grad()
    terms = diff.training_losses(model, x0, gg["t"].to(dev()), model_kwargs=kw, noise=tn)
    report("trans_dec training loss", terms["loss"], gg["loss"], 5e-5)
    terms["loss"].mean().backward()
    names = [k[2:] for k in gg if k.startswith("g/")]
    assert len(names) == 424
    params = dict(model.named_parameters())
    errs = []
    for n in names:
        assert params[n].grad is not None, n
        sample, _ = _digest(params[n].grad)
        scale = max(gg["g/" + n].abs().max().item(), 1e-4)
        errs.append((((sample - gg["g/" + n]).abs().max() / scale).item(), n))
    errs.sort(reverse=True)
    print(f"[parity] 424 CMDM trans_dec gradients vs the reference's backward: worst scaled errs " + ", ".join(f"{e:.2e} {n}" for e, n in errs[:3]))
    assert errs[0][0] <= 2e-3, f"{errs[0][1]}: scaled grad err {errs[0][0]:.3e}"


def test_cdm_train_mode_full_size_and_learning():
    """N = 8192 points, train mode (attention dropout on): reproducible with a fixed seed, loss falls under fused AdamW."""
    model, diff = _cdm_model()
    model.train()
    B, N = 4, 8192
    x0 = synth.gaussian("cdm_tf_x0", (B, N, 6)).to(dev())
    kw = dict(c_text_feat=synth.text_feature(B).to(dev()), c_pc_xyz=synth.scene_cloud(B, N).to(dev()))
    t = torch.tensor([5, 120, 333, 499], device=dev())

    def run():
        torch.manual_seed(9)
        model._drop_calls, diff._loss_calls = 0, 0
        model.zero_grad()
        terms = diff.training_losses(model, x0, t, model_kwargs=kw)
        terms["loss"].mean().backward()
        return terms["loss"].detach().clone(), model.contact_layer.weight.grad.clone()
    l1, g1 = run()
    l2, g2 = run()
    assert torch.equal(l1, l2) and torch.equal(g1, g2)
    params = [p_ for n_, p_ in model.named_parameters() if p_.requires_grad and not n_.startswith("text_model")]
    st, losses = {}, []
    for it in range(8):
        model.zero_grad()
        diff._loss_calls = 0
        terms = diff.training_losses(model, x0, t, model_kwargs=kw)
        terms["loss"].mean().backward()
        losses.append(terms["loss"].mean().item())
        AG.adamw_step(params, st, lr=1e-4)
    print("[train] CDM losses:", [round(v, 4) for v in losses])
    assert losses[-1] < losses[0]


# ------------------------------------------------------------------------------------------------ CDM 'MLP' arch (config default)
@pytest.mark.parametrize("rows,dim", [(513, 646), (100, 38), (7, 1022)])
def test_layernorm_any_width_forward_backward(rows, dim):
    from afm import ops
    x, dy = g("lg_x", (rows, dim)), g("lg_dy", (rows, dim))
    gam, bet = 1.0 + 0.1 * g("lg_g", (dim,)), 0.1 * g("lg_b", (dim,))
    xr, gr, br = x.double().requires_grad_(True), gam.double().requires_grad_(True), bet.double().requires_grad_(True)
    yr = F.layer_norm(xr, (dim,), gr, br, 1e-5)
    yr.backward(dy.double())
    report("LN (any width) forward", ops.layernorm(x.to(dev()), gam.to(dev()), bet.to(dev())), yr, 2e-5)
    dx, _, dg, db = AG._layernorm_bwd(x.to(dev()), gam.to(dev()), dy.to(dev()))
    rel("LN (any width) dx", dx, xr.grad, 2e-5)
    rel("LN (any width) dgamma", dg, gr.grad, 2e-5)
    rel("LN (any width) dbeta", db, br.grad, 2e-5)


def test_cdm_mlp_arch_forward_and_gradients_vs_reference():
    from afm.config import load_config
    cfg = load_config("text_to_motion_contact_gen", "cdm", ["model.input_feats=6", "task.dataset.use_openscene=True", "diffusion.steps=500",
                                                           "model.text_model.max_length=20"])
    assert cfg.model.arch == "MLP"
    model, diff = create_model_and_diffusion(cfg, device=dev())
    load_named_weights(model)
    model = model.to(dev()).eval()
    gf, gm = golden("cdm_forward_N256"), golden("cdm_mlp_N256")
    kw = dict(c_text_feat=gf["text_feat"].to(dev()), c_pc_xyz=gf["xyz"].to(dev()), c_pc_feat=synth.gaussian("cdm_pc_feat", (2, 256, 32)).to(dev()))
    with torch.no_grad():
        out = model(gf["x"].to(dev()), gm["t"].to(dev()), **kw)
    report("CDM 'MLP' arch forward vs reference", out, gm["out"], 2e-4)
    x0, tn = synth.gaussian("cdm_train_x0", (2, 256, 6)).to(dev()), synth.gaussian("cdm_train_noise", (2, 256, 6)).to(dev())
    model.zero_grad()
    terms = diff.training_losses(model, x0, gm["t_train"].to(dev()), model_kwargs=kw, noise=tn)
    report("CDM 'MLP' training loss", terms["loss"], gm["loss"], 5e-5)
    terms["loss"].mean().backward()
    params = dict(model.named_parameters())
    worst = 0.0
    names = [k[2:] for k in gm if k.startswith("g/")]
    for n in names:
        sample, _ = _digest(params[n].grad)
        scale = max(gm["g/" + n].abs().max().item(), 1e-4)
        err = ((sample - gm["g/" + n]).abs().max() / scale).item()
        worst = max(worst, err)
        assert err <= 1e-3, f"{n}: {err:.3e}"
    print(f"[parity] {len(names)} gradients of the CDM 'MLP' arch vs the reference's backward: worst scaled err {worst:.3e}")
