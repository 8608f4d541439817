"""`-m gpu`: CMDM denoiser + DDPM loop on the HIP path vs reference goldens and the CPU oracle.
Tolerances (f32 MFMA, exact-f32 products): one forward 2e-4 abs on O(1) outputs; 20-step loop with
shared noise 1e-3."""
import pytest
import torch

from afm import synth
from afm.config import to_config
from afm.base import create_model, create_model_and_diffusion, create_gaussian_diffusion
from conftest import golden
from gpu_util import dev, load_named_weights, report

pytestmark = pytest.mark.gpu


def cmdm_cfg(num_points=1024, steps=1000, respacing="", input_feats=263, data_repr="h3d", time_emb_dim=512):
    return to_config(dict(
        model=dict(name="CMDM", input_feats=input_feats, data_repr=data_repr, time_emb_dim=time_emb_dim,
                   contact_model=dict(contact_type="contact_cont_joints", contact_joints=[0, 10, 11, 12, 20, 21],
                                      planes=[32, 64, 128, 256], num_points=num_points, blocks=[2, 2, 2, 2]),
                   text_model=dict(version="ViT-B/32", max_length=20), arch="trans_enc", latent_dim=512,
                   mask_motion=True, num_layers=[1, 1, 1, 1, 1], num_heads=8, dropout=0.1, dim_feedforward=1024),
        diffusion=dict(predict_xstart=True, steps=steps, noise_schedule="cosine", timestep_respacing=respacing,
                       rescale_timesteps=False, loss_type="MSE", learn_sigma=False, sigma_small=True)))


@pytest.fixture(scope="module")
def cmdm():
    cfg = cmdm_cfg()
    model, diff = create_model_and_diffusion(cfg, device=dev())
    load_named_weights(model)
    return model.to(dev()).eval(), diff


def _kw(g, with_encoder=False):
    kw = dict(c_text_feat=g["text_feat"].to(dev()), x_mask=g["x_mask"].to(dev()), info_dummy=[0, 1])
    if with_encoder:
        kw.update(c_pc_xyz=g["xyz"].to(dev()), c_pc_contact=g["contact"].to(dev()))
    else:
        kw.update(c_cont_emb=g["cont_emb"].to(dev()))
    return kw


def test_forward_vs_reference_golden(cmdm):
    model, _ = cmdm
    g = golden("cmdm_forward_N1024_L16")
    out = model(g["x"].to(dev()), g["t"].to(dev()), **_kw(g))
    report("CMDM forward (cont_emb given) vs reference", out, g["out"], 2e-4)


@pytest.mark.parametrize("tt", [999, 500, 1, 0])
def test_p_sample_vs_reference_golden(cmdm, tt):
    model, diff = cmdm
    g, gs = golden("cmdm_forward_N1024_L16"), golden(f"cmdm_p_sample_t{tt}")
    t = torch.tensor([tt, tt], device=dev())
    out = diff.p_sample(model, gs["x"].to(dev()), t, clip_denoised=False, model_kwargs=_kw(g), noise=gs["noise"].to(dev()))
    report(f"p_sample t={tt} pred_xstart", out["pred_xstart"], gs["pred_xstart"], 2e-4)
    report(f"p_sample t={tt} sample", out["sample"], gs["sample"], 2e-4)


@pytest.mark.parametrize("steps,resp,tag", [(1000, "5", "r5"), (20, "", "T20")])
def test_sample_loop_vs_reference_golden(cmdm, steps, resp, tag):
    """Native loop (afm_cmdm_sample_loop, fused DDPM epilogue) and the generic python loop both
    reproduce the reference's p_sample_loop with the recorded per-step noise."""
    model, _ = cmdm
    diff = create_gaussian_diffusion(cmdm_cfg(steps=steps, respacing=resp))
    g = golden("cmdm_forward_N1024_L16")
    want = golden(f"cmdm_loop_{tag}")["sample"]
    nz = torch.stack([synth.gaussian(f"loop_{tag}_{j}", (2, 16, 263)) for j in range(diff.num_timesteps)]).to(dev())
    xT = synth.gaussian(f"loop_{tag}_xT", (2, 16, 263)).to(dev())
    native = diff.p_sample_loop(model, (2, 16, 263), noise=xT, clip_denoised=False, model_kwargs=_kw(g), step_noise=nz)
    report(f"native loop {tag}", native, want, 1e-3)
    assert torch.equal(xT.cpu(), synth.gaussian(f"loop_{tag}_xT", (2, 16, 263)))      # caller's x_T untouched
    generic = None
    for out in diff.p_sample_loop_progressive(model, (2, 16, 263), noise=xT, clip_denoised=False, model_kwargs=_kw(g), step_noise=nz):
        generic = out["sample"]
    report(f"generic loop {tag}", generic, want, 1e-3)
    report(f"native vs generic {tag}", native, generic.cpu(), 1e-5)
    # test.py:94-101 passes progress=True: the sliced native loop is bit-identical, with recorded noise and with Philox noise
    sliced = diff.p_sample_loop(model, (2, 16, 263), noise=xT, clip_denoised=False, model_kwargs=_kw(g), step_noise=nz, progress=True)
    assert torch.equal(native, sliced)
    if tag == "r5":
        d70 = create_gaussian_diffusion(cmdm_cfg(steps=1000, respacing="70"))
        kwi = dict(_kw(g), info_set_split=["test", "test"], c_text=["a", "b"])
        a = d70.p_sample_loop(model, (2, 16, 263), clip_denoised=False, noise=None, model_kwargs=kwi, seed=5, sample_index0=3)
        b = d70.p_sample_loop(model, (2, 16, 263), clip_denoised=False, noise=None, model_kwargs=kwi, seed=5, sample_index0=3, progress=True)
        assert torch.equal(a, b)


def test_training_losses_vs_reference_golden(cmdm):
    model, diff = cmdm
    g, gl = golden("cmdm_forward_N1024_L16"), golden("cmdm_training_losses")
    x0, tn = synth.gaussian("train_x0", (2, 16, 263)).to(dev()), synth.gaussian("train_noise", (2, 16, 263)).to(dev())
    with torch.no_grad():
        l = diff.training_losses(model, x0, gl["t"].to(dev()), model_kwargs=_kw(g), noise=tn)
    report("training loss (masked)", l["loss"], gl["loss_masked"], 1e-4)


def test_full_size_vs_oracle():
    """BASELINE config [1] shape (B=2 here for oracle time): L=196, 128 contact tokens, T=326."""
    from oracle import denoiser_ref as dr, shapes as sh
    cfg = cmdm_cfg(num_points=8192)
    model, _ = create_model_and_diffusion(cfg, device=dev())
    load_named_weights(model)
    model = model.to(dev()).eval()
    B, L = 2, 196
    x = synth.gaussian("full_x", (B, L, 263)); t = torch.tensor([999, 17])
    cont = synth.gaussian("full_cont", (B, 128, 256)); text = synth.text_feature(B)
    mask = synth.frame_mask(B, L, seed=3)
    want = dr.cmdm_forward(sh.weights(sh.cmdm()), x, t, text, x_mask=mask, cont_emb=cont)
    got = model(x.to(dev()), t.to(dev()), c_text_feat=text.to(dev()), c_cont_emb=cont.to(dev()), x_mask=mask.to(dev()))
    report("CMDM forward L=196 T=326 vs oracle", got, want, 3e-4)


def test_split_bf16_gemm_modes_track_native_f32_over_a_sampling_run():
    """afm_linear_set_split(9 | 6) vs 0: the exact three-way bf16 operand split on the bf16 matrix pipe is f32 arithmetic in another
    summation order, so a 200-step sampling run (full-size tokens, shared Philox noise) stays within f32 drift of the native run."""
    from afm import ops as afm_ops
    cfg = cmdm_cfg(num_points=8192, steps=1000, respacing="200")
    model, diff = create_model_and_diffusion(cfg, device=dev())
    load_named_weights(model)
    model = model.to(dev()).eval()
    B, L = 4, 196
    kw = dict(c_text_feat=synth.text_feature(B).to(dev()), c_cont_emb=synth.gaussian("full_cont", (B, 128, 256)).to(dev()),
              x_mask=synth.frame_mask(B, L, seed=3).to(dev()))
    runs = {}
    saved = afm_ops.get_gemm_split()
    try:
        for products in (0, 9, 6):                       # every eligible GEMM (min_n = 0), not only the default's wide ones
            afm_ops.set_gemm_split(products, 0)
            runs[products] = diff.p_sample_loop(model, (B, L, 263), clip_denoised=False, model_kwargs=kw, seed=21)
    finally:
        afm_ops.set_gemm_split(*saved)
    valid = ~kw["x_mask"]
    scale = runs[0][valid].abs().max().item()
    for products in (9, 6):
        drift = (runs[products] - runs[0])[valid].abs().max().item()
        print(f"split x{products}: max |x_0 - native| after 200 steps = {drift:.2e} (max |x_0| = {scale:.2f})")
        assert drift <= 2e-4 * max(scale, 1.0)
    assert not torch.equal(runs[9], runs[0])             # the mode really switched kernels


def test_sharding_invariance_of_native_loop(cmdm):
    """Philox noise is keyed by the global sample index: sampling 2 samples at once equals sampling
    them one by one with sample_index0 = 0 / 1 (what rank r does for its shard)."""
    model, _ = cmdm
    diff = create_gaussian_diffusion(cmdm_cfg(steps=1000, respacing="4"))
    g = golden("cmdm_forward_N1024_L16")
    kw = _kw(g)
    both = diff.p_sample_loop(model, (2, 16, 263), clip_denoised=False, model_kwargs=kw, seed=11)
    parts = []
    for b in range(2):
        kwb = {k: (v[b:b + 1] if isinstance(v, torch.Tensor) else v) for k, v in kw.items()}
        parts.append(diff.p_sample_loop(model, (1, 16, 263), clip_denoised=False, model_kwargs=kwb, seed=11, sample_index0=b))
    report("sharded vs whole", torch.cat(parts), both.cpu(), 1e-5)


def test_forward_with_scene_encoder_vs_reference_golden(cmdm):
    """Whole CMDM.forward including the SceneMapEncoder (FPS, kNN, set abstraction, vector attention)."""
    model, _ = cmdm
    g = golden("cmdm_forward_N1024_L16")
    out = model(g["x"].to(dev()), g["t"].to(dev()), **_kw(g, with_encoder=True))
    report("CMDM forward (full, with contact encoder) vs reference", out, g["out"], 3e-4)
    model.hoist_conditions = False                 # "faithful" mode recomputes the conditions per call
    out2 = model(g["x"].to(dev()), g["t"].to(dev()), **_kw(g, with_encoder=True))
    model.hoist_conditions = True
    assert torch.equal(out.cpu(), out2.cpu())      # hoisting is exactly output-neutral


def test_sub_batch_streams_are_bit_identical(cmdm):
    """Splitting the batch over HIP side streams (tail filling) must not change a single bit."""
    model, _ = cmdm
    diff = create_gaussian_diffusion(cmdm_cfg(steps=1000, respacing="6"))
    g = golden("cmdm_forward_N1024_L16")
    kw = _kw(g)
    outs = []
    model.loop_streams_auto = False              # take the stream count literally (the automatic rule uses one stream below B = 16)
    for n in (1, 2):
        model.loop_streams = n
        outs.append(diff.p_sample_loop(model, (2, 16, 263), clip_denoised=False, model_kwargs=kw, seed=5).cpu())
    model.loop_streams, model.loop_streams_auto = 2, True
    assert torch.equal(outs[0], outs[1])


def test_two_stream_loop_repeats_at_the_headline_shape():
    """Round 2 (profiles/r02_decfold_nondeterminism.md): a kernel can be bit-exact on one stream and wrong once in a while when a second
    stream's kernels share the chip - a single run-twice check passes most of the time.  The headline loop (B = 32, L = 196, N = 8192)
    runs as two sub-batch streams: ten 100-step loops against the single-stream result, bit for bit."""
    B, L, N = 32, 196, 8192
    model = create_model(cmdm_cfg(num_points=N), device=dev()); load_named_weights(model); model = model.to(dev()).eval()
    diff = create_gaussian_diffusion(cmdm_cfg(steps=1000, respacing="100"))
    kw = dict(c_text_feat=synth.text_feature(B).to(dev()), c_pc_xyz=synth.scene_cloud(B, N, seed=3).to(dev()),
              c_pc_contact=synth.contact_map(B, N).to(dev()), x_mask=synth.frame_mask(B, L, seed=2).to(dev()))

    def run(streams):
        model.loop_streams, model.loop_streams_auto = streams, False
        return diff.p_sample_loop(model, (B, L, 263), clip_denoised=False, model_kwargs=kw, seed=9).clone()

    ref = run(1)
    assert torch.isfinite(ref).all()
    for r in range(10):
        junk = torch.randn(32 << 20, device=dev()) if r % 2 else None      # vary allocator state and stream timing between the loops
        out = run(2)
        del junk
        bad = (out != ref).flatten(1).any(1).nonzero().flatten().tolist()
        assert not bad, f"loop {r}: samples {bad} differ from the single-stream result (max {(out - ref).abs().max().item():.2e})"


def test_config0_100_step_loop_vs_oracle():
    """BASELINE configs[0]: CMDM t2m_contact_motion config, synthetic B=4, L=60, D=263, 100 DDPM steps,
    shared explicit noise, HIP native loop vs the CPU oracle loop (drift over 100 sequential steps)."""
    from oracle import denoiser_ref as dr, diffusion_ref as df, shapes as sh
    cfg = cmdm_cfg(num_points=8192, steps=100)
    model, diff = create_model_and_diffusion(cfg, device=dev())
    load_named_weights(model)
    model = model.to(dev()).eval()
    B, L = 4, 60
    text, cont = synth.text_feature(B), synth.gaussian("c0_cont", (B, 128, 256))
    mask = synth.frame_mask(B, L, seed=9)
    xT = synth.gaussian("c0_xT", (B, L, 263))
    nz = [synth.gaussian(f"c0_nz{j}", (B, L, 263)) for j in range(100)]
    sd = sh.weights(sh.cmdm())
    want = df.p_sample_loop(df.Schedule(100), lambda x, t, **k: dr.cmdm_forward(sd, x, t, text, x_mask=mask, cont_emb=cont), xT, nz)
    got = diff.p_sample_loop(model, (B, L, 263), noise=xT.to(dev()), clip_denoised=False, step_noise=torch.stack(nz).to(dev()),
                             model_kwargs=dict(c_text_feat=text.to(dev()), c_cont_emb=cont.to(dev()), x_mask=mask.to(dev())))
    report("config[0] 100-step loop vs oracle", got, want, 1e-3)


def test_full_size_properties():
    """BASELINE configs[1] size (B=32, L=196, T=326), where the oracle is too slow to run in a test: size-independent
    properties of the denoiser - determinism, batch-permutation equivariance (samples are independent), and padded
    frames cannot influence the valid frames of the same sample."""
    cfg = cmdm_cfg(num_points=8192)
    model, _ = create_model_and_diffusion(cfg, device=dev())
    load_named_weights(model)
    model = model.to(dev()).eval()
    B, L = 32, 196
    x = synth.gaussian("fp_x", (B, L, 263)).to(dev()); t = torch.arange(B, device=dev()) * 31
    text, cont = synth.text_feature(B).to(dev()), synth.gaussian("fp_cont", (B, 128, 256)).to(dev())
    mask = synth.frame_mask(B, L, seed=4).to(dev())
    kw = dict(c_text_feat=text, c_cont_emb=cont, x_mask=mask)
    out = model(x, t, **kw)
    assert torch.isfinite(out).all()
    assert torch.equal(out, model(x, t, **kw))                                        # deterministic
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(0)).to(dev())
    outp = model(x[perm].contiguous(), t[perm].contiguous(), c_text_feat=text[perm].contiguous(),
                 c_cont_emb=cont[perm].contiguous(), x_mask=mask[perm].contiguous())
    assert torch.equal(outp, out[perm])                                                # sample independence, bit-exact
    x2 = x.clone()
    x2[mask] = 1234.5                                                                  # garbage in the padded frames
    out2 = model(x2, t, **kw)
    valid = ~mask
    assert torch.equal(out2[valid], out[valid])                                        # masked keys carry exactly zero weight


def test_forward_with_condition_switches_vs_reference_golden(cmdm):
    """c_text_mask / c_text_erase / c_pc_mask / c_pc_erase (training-time augmentations, cmdm.py:142-155) in eval mode."""
    model, diff = cmdm
    g, gm = golden("cmdm_forward_N1024_L16"), golden("cmdm_forward_cond_masks")
    sw = {k: gm[k].to(dev()) for k in ("c_text_mask", "c_text_erase", "c_pc_mask", "c_pc_erase")}
    x, t = g["x"].to(dev()), g["t"].to(dev())
    report("CMDM forward, all four switches", model(x, t, **_kw(g), **sw), gm["out_all"], 2e-4)
    report("CMDM forward, c_text_mask", model(x, t, **_kw(g), c_text_mask=sw["c_text_mask"]), gm["out_text_mask"], 2e-4)
    report("CMDM forward, c_pc_erase", model(x, t, **_kw(g), c_pc_erase=sw["c_pc_erase"]), gm["out_pc_erase"], 2e-4)
    # the per-operator composition equals the fused forward when no switch is set
    with torch.no_grad():
        report("composed forward == fused forward", model.forward_train(x, t, **_kw(g)), model(x, t, **_kw(g)), 2e-5)


def test_trans_dec_forward_and_sampling_vs_reference_golden():
    """`model.arch=trans_dec` (cmdm.py:78-113,171-191): self-attention stacks interleaved with cross-attention over the
    multi-scale scene memories (N/64 ... N points); cross-attention runs on the generalised flash kernel."""
    cfg = cmdm_cfg()
    cfg.model.arch = "trans_dec"
    model, diff = create_model_and_diffusion(cfg, device=dev())
    load_named_weights(model)
    model = model.to(dev()).eval()
    g = golden("cmdm_forward_N1024_L16")
    kw = dict(c_text_feat=g["text_feat"].to(dev()), c_pc_xyz=g["xyz"].to(dev()), c_pc_contact=g["contact"].to(dev()), x_mask=g["x_mask"].to(dev()))
    with torch.no_grad():
        out = model(g["x"].to(dev()), g["t"].to(dev()), **kw)
    valid = ~g["x_mask"]            # padded frames: the reference's nested-tensor fast path zero-fills them (see the oracle test)
    report("CMDM trans_dec forward vs reference (un-padded frames)", out.cpu()[valid], golden("cmdm_trans_dec_N1024_L16")["out"][valid], 5e-4)
    d5 = create_gaussian_diffusion(cmdm_cfg(respacing="3"))
    s = d5.p_sample_loop(model, (2, 16, 263), clip_denoised=False, model_kwargs=kw, seed=5)      # step-by-step path (no native loop)
    assert torch.isfinite(s).all() and s.shape == (2, 16, 263)


def test_cross_attention_kernel_vs_float64():
    from afm import ops
    B, Tq, Tk, H = 2, 198, 2048, 8
    d = 64 * H
    q, kv = synth.gaussian("xa_q", (B, Tq, d)), synth.gaussian("xa_kv", (B, Tk, 2 * d))
    mask = torch.zeros(B, Tk, dtype=torch.bool); mask[1, 1500:] = True
    sp = lambda z: z.double().view(B, z.shape[1], H, 64).transpose(1, 2)
    sc = sp(q) @ sp(kv[..., :d]).transpose(-1, -2) / 8.0
    sc = sc.masked_fill(mask[:, None, None, :], float("-inf"))
    ref = (torch.softmax(sc, -1) @ sp(kv[..., d:])).transpose(1, 2).reshape(B, Tq, d)
    report("cross-attention (Tq=198, Tk=2048)", ops.mha_cross(q.to(dev()), kv.to(dev()), mask.to(dev()), H), ref, 2e-5)
    Tk = 8192
    kv = synth.gaussian("xa_kv8", (1, Tk, 2 * d))
    sp1 = lambda z: z.double().view(1, z.shape[1], H, 64).transpose(1, 2)
    sc = sp1(q[:1]) @ sp1(kv[..., :d]).transpose(-1, -2) / 8.0
    ref = (torch.softmax(sc, -1) @ sp1(kv[..., d:])).transpose(1, 2).reshape(1, Tq, d)
    report("cross-attention (Tk=8192, >64 KB LDS)", ops.mha_cross(q[:1].to(dev()), kv.to(dev()), None, H), ref, 2e-5)


def test_default_seed_draws_fresh_noise_per_call(cmdm):
    """test.py:95-102 calls p_sample_loop(model, shape, noise=None, ...) k_sample times without a seed and expects k
    DIFFERENT samples (th.randn advances the global RNG); reproducible from torch.manual_seed like the reference."""
    model, _ = cmdm
    g = golden("cmdm_forward_N1024_L16")
    kw = _kw(g)

    def three():
        torch.manual_seed(2023)
        diff = create_gaussian_diffusion(cmdm_cfg(steps=1000, respacing="4"))
        return [diff.p_sample_loop(model, (2, 16, 263), clip_denoised=False, model_kwargs=kw).cpu() for _ in range(3)]

    a, b = three(), three()
    assert not torch.equal(a[0], a[1]) and not torch.equal(a[1], a[2]) and not torch.equal(a[0], a[2])
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    assert (a[0] - a[1]).abs().max() > 1e-2


def test_condition_cache_is_not_fooled_by_a_recycled_address(cmdm):
    """a dataloader loop frees batch k's device tensors; the caching allocator gives batch k+1 the same address with
    version 0 and the same shape.  The condition tokens must be recomputed for the new batch."""
    model, _ = cmdm
    g = golden("cmdm_forward_N1024_L16")
    x, t = g["x"].to(dev()), g["t"].to(dev())
    base_kw = _kw(g, with_encoder=True)
    hits = 0
    for trial in range(6):
        xyz = base_kw["c_pc_xyz"].clone()
        kw = dict(base_kw, c_pc_xyz=xyz)
        ptr = xyz.data_ptr()
        out_a = model(x, t, **kw).clone()
        del xyz, kw
        xyz2 = (base_kw["c_pc_xyz"] * 0.5 + 0.1).contiguous()        # another scene, same shape
        hits += int(xyz2.data_ptr() == ptr)
        kw2 = dict(base_kw, c_pc_xyz=xyz2)
        out_b = model(x, t, **kw2)
        model.hoist_conditions = False
        want_b = model(x, t, **kw2)
        model.hoist_conditions = True
        assert torch.equal(out_b, want_b), f"trial {trial}: stale condition tokens reused"
        assert not torch.equal(out_a, out_b)
        del xyz2, kw2
    print(f"[cache] allocator handed back the freed address in {hits}/6 trials (held keys make that impossible: expected 0)")


def test_1000_step_drift_vs_oracle():
    """VERDICT r1 #4a / SURVEY 8c: the FULL 1000-step chain (gaussian_diffusion.py:442-536) at the headline frame count (L = 196,
    T = 326 tokens, B = 2) with shared recorded noise, HIP native loop vs the CPU oracle loop.  Stated tolerance: 1e-3 abs after 1000
    sequential steps (values are O(1..4)); the drift after 10 / 100 / 1000 executed steps is printed."""
    from oracle import denoiser_ref as dr, diffusion_ref as df, shapes as sh
    cfg = cmdm_cfg(num_points=8192, steps=1000)
    model, diff = create_model_and_diffusion(cfg, device=dev())
    load_named_weights(model)
    model = model.to(dev()).eval()
    B, L, n = 2, 196, 1000
    text, cont = synth.text_feature(B), synth.gaussian("d1k_cont", (B, 128, 256))
    mask = synth.frame_mask(B, L, seed=12)
    xT = synth.gaussian("d1k_xT", (B, L, 263))
    gen = torch.Generator().manual_seed(20260927)
    nz = torch.randn(n, B, L, 263, generator=gen)
    sd = sh.weights(sh.cmdm())
    s = df.Schedule(n)
    marks, want = (10, 100, 1000), {}
    torch.set_num_threads(min(32, torch.get_num_threads()))
    img = xT
    with torch.no_grad():
        for j, i in enumerate(range(n - 1, -1, -1)):
            img = df.p_sample(s, lambda x, t, **k: dr.cmdm_forward(sd, x, t, text, x_mask=mask, cont_emb=cont), img,
                              torch.tensor([i] * B), nz[j])["sample"]
            if j + 1 in marks:
                want[j + 1] = img.clone()
    snaps = {10: None, 100: None}
    got = diff.p_sample_loop(model, (B, L, 263), noise=xT.to(dev()), clip_denoised=False, step_noise=nz.to(dev()), snapshots=snaps,
                             model_kwargs=dict(c_text_feat=text.to(dev()), c_cont_emb=cont.to(dev()), x_mask=mask.to(dev())))
    snaps[1000] = got
    valid = ~mask                               # padded frames are never read (loss mask / m_len cut); parity is stated on valid frames
    for k in marks:
        d = (snaps[k].cpu() - want[k])[valid].abs().max().item()
        print(f"[drift] after {k:4d} executed steps: max|HIP - oracle| = {d:.3e} (max|x| = {want[k][valid].abs().max().item():.2f})")
    report("1000-step loop vs oracle (valid frames)", snaps[1000].cpu()[valid], want[1000][valid], 1e-3)
    report("100-step prefix vs oracle (valid frames)", snaps[100].cpu()[valid], want[100][valid], 1e-3)


def test_bf16_one_product_drift_is_measured_and_fails_the_f32_bar():
    """VERDICT r1 #9 (informational): AFM_ARITH_BF16X1 keeps only the product of the leading bf16 terms - what a plain bf16 GEMM with
    f32 accumulation computes.  Same 1000-step chain, same noise, against the default exact-split arithmetic (itself ~4e-6 from the
    oracle, test above): the drift after 10 / 100 / 1000 steps is printed, and it must EXCEED the 1e-3 parity tolerance - the reason the
    library never selects it."""
    from afm import ops
    cfg = cmdm_cfg(num_points=8192, steps=1000)
    model, diff = create_model_and_diffusion(cfg, device=dev())
    load_named_weights(model)
    model = model.to(dev()).eval()
    B, L = 2, 196
    kw = dict(c_text_feat=synth.text_feature(B).to(dev()), c_cont_emb=synth.gaussian("d1k_cont", (B, 128, 256)).to(dev()),
              x_mask=synth.frame_mask(B, L, seed=12).to(dev()))
    saved = ops.get_gemm_split()
    chains = {}
    try:
        for products in (9, 1):
            ops.set_gemm_split(products, 0)
            snaps = {10: None, 100: None}
            snaps[1000] = diff.p_sample_loop(model, (B, L, 263), clip_denoised=False, model_kwargs=kw, seed=5, snapshots=snaps)
            chains[products] = snaps
    finally:
        ops.set_gemm_split(*saved)
    valid = ~kw["x_mask"]
    drift = {k: (chains[1][k] - chains[9][k])[valid].abs().max().item() for k in (10, 100, 1000)}
    for k, d in drift.items():
        print(f"[bf16x1] after {k:4d} executed steps: max|bf16 one-product - default| = {d:.3e} "
              f"(max|x| = {chains[9][k][valid].abs().max().item():.2f})")
    assert all(torch.isfinite(chains[1][k]).all() for k in chains[1])
    assert drift[1000] > 1e-3, "a plain-bf16 GEMM inside the f32 tolerance would be worth shipping; it is not"


def test_small_batch_loop_is_bit_identical_to_the_large_batch():
    """Strong scaling gives each GPU 4 (or 1) of the job's samples: the GEMM tile shapes and attention groupings chosen for the small
    launch differ from the B = 32 launch's, the bits must not (T = 326, full-size layers, Philox noise keyed by global sample)."""
    cfg = cmdm_cfg(num_points=8192, steps=1000, respacing="6")
    model, diff = create_model_and_diffusion(cfg, device=dev())
    load_named_weights(model)
    model = model.to(dev()).eval()
    B, L = 16, 196
    kw = dict(c_text_feat=synth.text_feature(B).to(dev()), c_cont_emb=synth.gaussian("sb_cont", (B, 128, 256)).to(dev()),
              x_mask=synth.frame_mask(B, L, seed=3).to(dev()))
    whole = diff.p_sample_loop(model, (B, L, 263), clip_denoised=False, model_kwargs=kw, seed=21)
    for nb in (4, 1):
        parts = []
        for b0 in range(0, B, nb):
            kwb = {k: v[b0:b0 + nb].contiguous() for k, v in kw.items()}
            parts.append(diff.p_sample_loop(model, (nb, L, 263), clip_denoised=False, model_kwargs=kwb, seed=21, sample_index0=b0))
            if nb == 1 and b0 >= 3:
                break
        got = torch.cat(parts)
        assert torch.equal(got, whole[: got.shape[0]]), f"shards of {nb} samples differ from the batch of {B}"


def test_last_layer_attention_on_the_motion_queries_only_is_bit_neutral():
    """Round 4: the last encoder layer's attention computes the L motion tokens' query rows only (afm_mha_fwd_rows, q_first = 2 + n_groups;
    nothing reads the other rows).  `model.all_queries` restores the full launch: a sampling loop and a single forward must not change by a
    bit, with ragged frame masks, at a large and at a small batch."""
    cfg = cmdm_cfg(num_points=8192, steps=1000, respacing="8")
    model, diff = create_model_and_diffusion(cfg, device=dev())
    load_named_weights(model)
    model = model.to(dev()).eval()
    for B, L in ((16, 196), (4, 196), (3, 60)):
        kw = dict(c_text_feat=synth.text_feature(B).to(dev()), c_cont_emb=synth.gaussian("aq_cont", (B, 128, 256)).to(dev()),
                  x_mask=synth.frame_mask(B, L, seed=7).to(dev()))
        x = synth.gaussian("aq_x", (B, L, 263)).to(dev())
        t = torch.arange(B, device=dev()) * 37 % 1000
        outs = {}
        for allq in (False, True):
            model.all_queries = allq
            outs[allq] = (diff.p_sample_loop(model, (B, L, 263), clip_denoised=False, model_kwargs=kw, seed=44).clone(), model(x, t, **kw).clone())
        model.all_queries = False
        assert torch.isfinite(outs[False][0]).all()
        assert torch.equal(outs[False][0], outs[True][0]) and torch.equal(outs[False][1], outs[True][1]), f"B={B}, L={L}"
        # round 4: the per-step prologue launch is gone (time tokens and the K-padded x_t ride on the step's first / last GEMM): same bits as
        # with the launch, also when the chain is cut into slices (progress bar) and continued
        model.no_riders = True
        with_launch = diff.p_sample_loop(model, (B, L, 263), clip_denoised=False, model_kwargs=kw, seed=44).clone()
        model.no_riders = False
        sliced = diff.p_sample_loop(model, (B, L, 263), clip_denoised=False, model_kwargs=kw, seed=44, snapshots={3: None, 5: None})
        assert torch.equal(with_launch, outs[False][0]) and torch.equal(sliced, outs[False][0]), f"riders, B={B}, L={L}"


def test_clip_denoised_native_loop_matches_the_step_by_step_path(cmdm):
    """`clip_denoised=True` is the reference's DEFAULT argument of p_sample_loop (gaussian_diffusion.py:442-449; test.py passes False): the native
    loop clamps pred_xstart to [-1, 1] inside the fused DDPM update (ABI v6).  Against the REFERENCE's own loop (golden `cmdm_loop_r5_clip`:
    default arguments, recorded noise) and against the step-by-step composition (model -> afm_clamp -> DDPM update kernel)."""
    model, _ = cmdm
    g = golden("cmdm_forward_N1024_L16")
    diff = create_gaussian_diffusion(cmdm_cfg(respacing="5"))
    B, L = 2, 16
    kw = _kw(g)
    x_T = 2.0 * synth.gaussian("loop_r5_xT", (B, L, 263))
    nz = [synth.gaussian(f"loop_r5_{j}", (B, L, 263)) for j in range(5)]
    native = diff.p_sample_loop(model, (B, L, 263), noise=x_T.to(dev()), model_kwargs=kw, step_noise=torch.stack(nz).to(dev()))          # default: clip on
    report("clip_denoised=True (default): native loop vs reference", native, golden("cmdm_loop_r5_clip")["sample"], 1e-3)
    steps = None
    for out in diff.p_sample_loop_progressive(model, (B, L, 263), noise=x_T.to(dev()), clip_denoised=True, model_kwargs=kw,
                                              step_noise=[z.to(dev()) for z in nz]):
        steps = out
    report("clip_denoised: native loop vs step-by-step", native, steps["sample"], 2e-5)
    assert steps["pred_xstart"].abs().max().item() <= 1.0
    unclipped = diff.p_sample_loop(model, (B, L, 263), noise=x_T.to(dev()), clip_denoised=False, model_kwargs=kw, step_noise=torch.stack(nz).to(dev()))
    assert (unclipped - native).abs().max().item() > 1e-2           # the clamp is live on this input


def test_paired_launch_loop_is_bit_identical():
    """Round 6: with `model.pair_launch` the two-stream loop computes sub-batch A's out_proj and sub-batch B's linear1 of every
    layer in ONE 128 x 128-tile launch (afm_linear_pair; two cross-stream edges per layer).  Same tile program on the same operands: the
    sampling result must not change by a bit - against the unpaired two-stream loop and the single-stream loop, at the headline shape
    (B = 32: 16 + 16 samples), an uneven split (B = 17: 9 + 8) and repeated runs (the cross-stream edges are the new hazard)."""
    L, N = 196, 8192
    model = create_model(cmdm_cfg(num_points=N), device=dev()); load_named_weights(model); model = model.to(dev()).eval()
    diff = create_gaussian_diffusion(cmdm_cfg(steps=1000, respacing="30"))
    for B in (32, 17):
        kw = dict(c_text_feat=synth.text_feature(B).to(dev()), c_cont_emb=synth.gaussian("pl_cont", (B, 128, 256)).to(dev()),
                  x_mask=synth.frame_mask(B, L, seed=6).to(dev()))

        def run(streams, pair):
            model.loop_streams, model.loop_streams_auto, model.pair_launch = streams, False, pair
            return diff.p_sample_loop(model, (B, L, 263), clip_denoised=False, model_kwargs=kw, seed=14).clone()
        try:
            ref1, ref2 = run(1, False), run(2, False)
            assert torch.isfinite(ref1).all() and torch.equal(ref1, ref2)
            for r in range(4):
                out = run(2, True)
                bad = (out != ref1).flatten(1).any(1).nonzero().flatten().tolist()
                assert not bad, f"B={B} run {r}: samples {bad} differ with the paired launch (max {(out - ref1).abs().max().item():.2e})"
            assert torch.equal(run(1, True), ref1)                      # one stream: the flag changes nothing
        finally:
            model.loop_streams, model.loop_streams_auto, model.pair_launch = 2, True, False


def test_two_stream_loop_soak():
    """Round 3: 50 two-stream 100-step loops of the headline CMDM shape (B = 32, L = 196, N = 8192) against the
    single-stream result, bit for bit - the harness that caught round 2's lat_decfold defect (profiles/r02_decfold_nondeterminism.md)."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    from loop_determinism_probe import probe_cmdm
    bad = probe_cmdm(50, dev())
    assert not bad, f"{len(bad)} of 50 two-stream loops differ from the single-stream result: {bad[:4]}"


def test_fused_layernorm_loop_is_bit_identical_to_separate_launches():
    """Round 3: norm1 / norm2 can run inside the out_proj / linear2 GEMMs (last-arriver LayerNorm, `model.fused_layernorm`; opt-in: it
    measured slower than the separate launches).  Same arithmetic as the separate launch: a sampling loop must not change by a bit, on
    one stream and on two, at a large and at a small batch."""
    cfg = cmdm_cfg(num_points=8192, steps=1000, respacing="12")
    model, diff = create_model_and_diffusion(cfg, device=dev())
    load_named_weights(model)
    model = model.to(dev()).eval()
    for B, L in ((16, 196), (3, 60)):
        kw = dict(c_text_feat=synth.text_feature(B).to(dev()), c_cont_emb=synth.gaussian("fl_cont", (B, 128, 256)).to(dev()),
                  x_mask=synth.frame_mask(B, L, seed=5).to(dev()))
        outs = {}
        for fused in (True, False):
            for streams in (1, 2):
                model.fused_layernorm, model.no_ln_fold = fused, not fused       # reference: separate LayerNorm launches
                model.loop_streams, model.loop_streams_auto = streams, False
                outs[(fused, streams)] = diff.p_sample_loop(model, (B, L, 263), clip_denoised=False, model_kwargs=kw, seed=33).clone()
        model.fused_layernorm = model.no_ln_fold = False
        ref = outs[(False, 1)]
        assert torch.isfinite(ref).all()
        for key, o in outs.items():
            assert torch.equal(o, ref), f"B={B}: fused={key[0]} streams={key[1]} differs by {(o - ref).abs().max().item():.3e}"


def test_folded_layernorm_loop_tracks_the_separate_launches():
    """Round 3: in eval mode norm1 / norm2 are folded across the kernel boundaries (no LayerNorm launches, statistics-carrying GEMM
    epilogues; `model.no_ln_fold` restores the launches).  A re-association of the same arithmetic: a 100-step loop at the headline shape
    stays within 1e-4 of the separate-launch loop on |x| <= ~4, and sub-batch streams / small batches stay bit-identical in the folded form."""
    cfg = cmdm_cfg(num_points=8192, steps=1000, respacing="100")
    model, diff = create_model_and_diffusion(cfg, device=dev())
    load_named_weights(model)
    model = model.to(dev()).eval()
    B, L = 8, 196
    kw = dict(c_text_feat=synth.text_feature(B).to(dev()), c_cont_emb=synth.gaussian("lf_cont", (B, 128, 256)).to(dev()),
              x_mask=synth.frame_mask(B, L, seed=6).to(dev()))
    w = model._weights()
    assert w.motion_layer_wg and w.layer[0].lin1_wg and w.layer[1].in_proj_wg and not w.layer[0].in_proj_wg
    run = lambda **k: diff.p_sample_loop(model, (B, L, 263), clip_denoised=False, model_kwargs=dict(kw, **k), seed=44).clone()
    folded = run()
    try:
        model.no_ln_fold = True
        separate = run()
    finally:
        model.no_ln_fold = False
    assert torch.isfinite(folded).all() and not torch.equal(folded, separate)
    report("100-step loop: LayerNorm folded across launches vs separate launches", folded, separate.cpu(), 1e-4)
    model.loop_streams, model.loop_streams_auto = 2, False
    try:
        assert torch.equal(run(), folded), "two sub-batch streams"
    finally:
        model.loop_streams, model.loop_streams_auto = 2, True
    kw3 = {k: v[:3].contiguous() for k, v in kw.items()}
    small = diff.p_sample_loop(model, (3, L, 263), clip_denoised=False, model_kwargs=kw3, seed=44)
    assert torch.equal(small, folded[:3]), "a shard of 3 samples"
    model.train()
    try:
        assert not model._weights().motion_layer_wg, "training mode keeps the plain weights (they change every step)"
    finally:
        model.eval()
