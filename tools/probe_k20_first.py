#!/usr/bin/env python
"""The bench flow in a fresh process: first-ever 5-step loop, then the FIRST-ever 20-step loop, timed; then the second and third 20-step
loops.  `prealloc` as argv[1]: touch the 20-step schedule scratch size in the caching allocator before the timed call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "afford-motion_amd")):
    sys.path.insert(0, p)
import torch
from afm import ffi, synth
from afm.base import create_gaussian_diffusion, create_model
from afm.config import load_config
dev = torch.device("cuda:0")
B, L, D, N = 32, 196, 263, 8192
cfg_for = lambda k: load_config("text_to_motion_contact_motion_gen", "cmdm", ["model.data_repr=h3d", "model.input_feats=263", f"diffusion.timestep_respacing='{k}'"])
model = create_model(cfg_for(20), device=dev); synth.fill_module_(model); model = model.to(dev).eval()
kw = dict(c_text_feat=synth.text_feature(B).to(dev), c_pc_xyz=synth.scene_cloud(B, N).to(dev), c_pc_contact=synth.contact_map(B, N).to(dev),
          x_mask=synth.frame_mask(B, L, all_valid=True).to(dev))
model.condition_tokens(**kw)
d5, d20 = create_gaussian_diffusion(cfg_for(5)), create_gaussian_diffusion(cfg_for(20))
d5.tables(dev); d20.tables(dev)
run = lambda d: d.p_sample_loop(model, (B, L, D), clip_denoised=False, model_kwargs=kw, seed=2)
if len(sys.argv) > 1:
    t = torch.empty(ffi.load().afm_cmdm_sched_scratch_bytes(20, B), dtype=torch.uint8, device=dev); del t
run(d5); torch.cuda.synchronize()
out = []
for i in range(3):
    t0 = time.perf_counter(); run(d20); te = time.perf_counter() - t0; torch.cuda.synchronize(); out.append((1e3 * (time.perf_counter() - t0) / 20, 1e3 * te))
print(("prealloc " if len(sys.argv) > 1 else "plain    ") + "first / second / third 20-step loop: " + " | ".join(f"{a:.4f} ms/step (host enqueue {b:.2f} ms)" for a, b in out))
