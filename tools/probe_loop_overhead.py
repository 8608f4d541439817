#!/usr/bin/env python
"""Fixed cost of one CMDM `p_sample_loop` call at the bench shape (B = 32, L = 196): wall time (synchronise before and after) of K-step calls
for several K -> intercept / slope, and a cProfile of the host side of 50 two-step calls.  python tools/probe_loop_overhead.py"""
import cProfile
import os
import pstats
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


def cfg_for(k):
    return load_config("text_to_motion_contact_motion_gen", "cmdm", ["model.data_repr=h3d", "model.input_feats=263", f"diffusion.timestep_respacing='{k}'"])


model = create_model(cfg_for(20), device=dev)
synth.fill_module_(model)
model = model.to(dev).eval()
kw = dict(c_text_feat=synth.text_feature(B).to(dev), c_pc_xyz=synth.scene_cloud(B, N).to(dev), c_pc_contact=synth.contact_map(B, N).to(dev),
          x_mask=synth.frame_mask(B, L, all_valid=True).to(dev))
model.condition_tokens(**kw)
res = {}
for K in (2, 5, 10, 20, 40):
    d = create_gaussian_diffusion(cfg_for(K))
    run = lambda: d.p_sample_loop(model, (B, L, D), clip_denoised=False, model_kwargs=kw, seed=2)
    run(); run()
    ts = []
    for _ in range(7):
        torch.cuda.synchronize(); t0 = time.perf_counter(); run(); te = time.perf_counter() - t0; torch.cuda.synchronize(); ts.append((time.perf_counter() - t0, te))
    ts.sort()
    res[K] = ts[3]
    print(f"K = {K:3d}: call {1e3 * ts[3][0]:8.3f} ms (host enqueue {1e3 * ts[3][1]:7.3f} ms)  -> {1e3 * ts[3][0] / K:7.3f} ms per step", flush=True)
slope = (res[40][0] - res[10][0]) / 30
print(f"slope (K = 10 .. 40) {1e3 * slope:.4f} ms per step; fixed cost of a call = {1e3 * (res[20][0] - 20 * slope):.3f} ms (K = 20), {1e3 * (res[2][0] - 2 * slope):.3f} ms (K = 2)")
d1 = create_gaussian_diffusion(cfg_for(2))
pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
for _ in range(50):
    d1.p_sample_loop(model, (B, L, D), clip_denoised=False, model_kwargs=kw, seed=2)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("cumulative"); st.print_stats(28)
