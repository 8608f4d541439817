"""Lean target for rocprofv3 --pmc passes: 12 native CMDM steps at the bench shape, single stream, no profiler events."""
import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/afford-motion_amd')
from afm import synth
from afm.base import create_model_and_diffusion
from afm.config import load_config
dev = torch.device('cuda:0')
cfg = load_config("text_to_motion_contact_motion_gen", "cmdm", ["model.data_repr=h3d", "model.input_feats=263", "diffusion.timestep_respacing='12'"])
model, diff = create_model_and_diffusion(cfg, device=dev)
synth.fill_module_(model); model = model.to(dev).eval(); model.loop_streams = 1
B, L = 32, 196
kw = dict(c_text_feat=synth.text_feature(B).to(dev), c_cont_emb=synth.gaussian("c", (B, 128, 256)).to(dev), x_mask=synth.frame_mask(B, L, all_valid=True).to(dev))
diff.p_sample_loop(model, (B, L, 263), clip_denoised=False, model_kwargs=kw, seed=1)
torch.cuda.synchronize()
