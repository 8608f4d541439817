#!/usr/bin/env python
"""Why is the driver's command (--steps 20 --warmup 5) ~7 % below the steady-state rate?  The bench flow (5 warm-up steps, synchronise, 20 timed
steps) after an idle period of various lengths, and back to back.  python tools/probe_k20_gap.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "afford-motion_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from afm import synth  # noqa: E402
from afm.base import create_gaussian_diffusion, create_model  # noqa: E402
from afm.config import load_config  # noqa: E402

dev = torch.device("cuda:0")
B, L, D, N = 32, 196, 263, 8192
cfg_for = lambda k: load_config("text_to_motion_contact_motion_gen", "cmdm", ["model.data_repr=h3d", "model.input_feats=263", f"diffusion.timestep_respacing='{k}'"])
model = create_model(cfg_for(20), device=dev)
synth.fill_module_(model)
model = model.to(dev).eval()
kw = dict(c_text_feat=synth.text_feature(B).to(dev), c_pc_xyz=synth.scene_cloud(B, N).to(dev), c_pc_contact=synth.contact_map(B, N).to(dev),
          x_mask=synth.frame_mask(B, L, all_valid=True).to(dev))
model.condition_tokens(**kw)
d5, d20, d200 = (create_gaussian_diffusion(cfg_for(k)) for k in (5, 20, 200))
for d in (d5, d20, d200):
    d.tables(dev)
run = lambda d: d.p_sample_loop(model, (B, L, D), clip_denoised=False, model_kwargs=kw, seed=2)


def timed(d, k):
    torch.cuda.synchronize(); t0 = time.perf_counter(); run(d); torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / k


run(d5); run(d20); torch.cuda.synchronize()
print(f"steady state: 200 steps at {timed(d200, 200):.4f} ms/step")
for idle in (0.0, 0.01, 0.1, 1.0, 3.0):
    res = []
    for _ in range(4):
        time.sleep(idle)
        run(d5); a = timed(d20, 20); b = timed(d20, 20)          # the bench flow, then the same 20 steps again at once
        res.append((a, b))
    print(f"idle {idle:5.2f} s -> 5 warm-up steps -> 20 timed steps: " + ", ".join(f"{a:.4f}" for a, _ in res) + " ms/step;  the next 20 steps: " + ", ".join(f"{b:.4f}" for _, b in res))
# per-step times inside one 20-step run after an idle second: 20 single-step calls chained through their output
time.sleep(1.0)
