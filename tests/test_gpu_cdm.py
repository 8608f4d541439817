"""`-m gpu`: CDM / ContactPerceiver denoiser on the HIP path vs the reference goldens and the CPU oracle.
The encoder/decoder cross-attentions are evaluated in folded form (no K/V over the N points), i.e. a
re-association of the same f32 arithmetic: tolerance 2e-4 abs on O(1) outputs."""
import pytest
import torch

from afm import synth
from afm.base import create_gaussian_diffusion, create_model
from afm.config import to_config
from conftest import golden
from gpu_util import dev, load_named_weights, report

pytestmark = pytest.mark.gpu


def cdm_cfg(num_points=256, point_feats=False, steps=500, respacing=""):
    sm = dict(name="PointTransformerSeg", use_scene_model=point_feats, use_color=False, use_openscene=point_feats,
              num_points=num_points, point_feat_dim=32, pretrained_weight="", freeze=True)
    return to_config(dict(
        model=dict(name="CDM", input_feats=6, data_repr="contact_cont_joints", time_emb_dim=128,
                   text_model=dict(version="ViT-B/32", max_length=20), scene_model=sm, arch="Perceiver",
                   arch_perceiver=dict(last_dim=256, point_pos_emb=True, encoder_q_input_channels=512, encoder_kv_input_channels=256,
                                       encoder_num_heads=8, encoder_widening_factor=1, encoder_dropout=0.1, encoder_residual_dropout=0.0,
                                       encoder_self_attn_num_layers=2, decoder_q_input_channels=256, decoder_kv_input_channels=512,
                                       decoder_num_heads=8, decoder_widening_factor=1, decoder_dropout=0.1, decoder_residual_dropout=0.0)),
        diffusion=dict(predict_xstart=True, steps=steps, noise_schedule="cosine", timestep_respacing=respacing,
                       rescale_timesteps=False, loss_type="MSE", learn_sigma=False, sigma_small=True)))


@pytest.fixture(scope="module")
def cdm():
    m = create_model(cdm_cfg(), device=dev())
    load_named_weights(m)
    return m.to(dev()).eval()


def test_forward_vs_reference_golden(cdm):
    g = golden("cdm_forward_N256")
    out = cdm(g["x"].to(dev()), g["t"].to(dev()), c_text_feat=g["text_feat"].to(dev()), c_pc_xyz=g["xyz"].to(dev()))
    report("CDM forward N=256 vs reference", out, g["out"], 2e-4)


def test_loop_vs_reference_golden(cdm):
    g = golden("cdm_forward_N256")
    diff = create_gaussian_diffusion(cdm_cfg(steps=500, respacing="4"))
    nz = torch.stack([synth.gaussian(f"cdm_loop_{j}", (2, 256, 6)) for j in range(diff.num_timesteps)]).to(dev())
    xT = synth.gaussian("cdm_loop_xT", (2, 256, 6)).to(dev())
    kw = dict(c_text_feat=g["text_feat"].to(dev()), c_pc_xyz=g["xyz"].to(dev()))
    out = diff.p_sample_loop(cdm, (2, 256, 6), noise=xT, clip_denoised=False, model_kwargs=kw, step_noise=nz)
    report("CDM 4-step loop vs reference", out, golden("cdm_loop_r4")["sample"], 1e-3)


def test_forward_with_point_features_vs_reference_golden():
    g, g2 = golden("cdm_forward_N256"), golden("cdm_forward_feat32")
    m = create_model(cdm_cfg(point_feats=True), device=dev())
    load_named_weights(m)
    m = m.to(dev()).eval()
    out = m(g["x"].to(dev()), g["t"].to(dev()), c_text_feat=g["text_feat"].to(dev()), c_pc_xyz=g["xyz"].to(dev()),
            c_pc_feat=g2["pc_feat"].to(dev()))
    report("CDM forward with 32-d point features vs reference", out, g2["out"], 2e-4)


def test_full_size_vs_oracle():
    """BASELINE configs[2] shape (B = 2 here): N = 8192 points."""
    from oracle import denoiser_ref as dr, shapes as sh
    m = create_model(cdm_cfg(num_points=8192), device=dev())
    load_named_weights(m)
    m = m.to(dev()).eval()
    B, N = 2, 8192
    x = synth.gaussian("cdm_full_x", (B, N, 6)); xyz = synth.scene_cloud(B, N, seed=51); text = synth.text_feature(B)
    t = torch.tensor([499, 3])
    want = dr.cdm_forward(sh.weights(sh.cdm()), x, t, text, xyz)
    got = m(x.to(dev()), t.to(dev()), c_text_feat=text.to(dev()), c_pc_xyz=xyz.to(dev()))
    report("CDM forward N=8192 vs oracle", got, want, 2e-4)
