#!/usr/bin/env python
"""Launches per step of the CMDM native loop by kernel family, per GPU batch size: the library's own profiler (afm_profile_*) counts the launches of
two loop calls of K1 and K2 steps; (count(K2) - count(K1)) / (K2 - K1) is the per-step figure with every per-call launch cancelled, the time per
launch comes from the same HIP-event brackets (serialising: for counting and attribution, not a throughput measurement).
    python tools/count_launches.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "afford-motion_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

from afm import ffi, synth  # noqa: E402
from afm.base import create_gaussian_diffusion, create_model  # noqa: E402
from afm.config import load_config  # noqa: E402

dev = torch.device("cuda:0")
L, D, NPTS = 196, 263, 8192
K1, K2 = 20, 60


def loop(B, steps, kw, m):
    cfg = load_config("text_to_motion_contact_motion_gen", "cmdm", ["model.data_repr=h3d", "model.input_feats=263", "diffusion.steps=1000",
                                                                   f"diffusion.timestep_respacing='{steps}'"])
    d = create_gaussian_diffusion(cfg)
    d.tables(dev)
    d.p_sample_loop(m, (B, L, D), clip_denoised=False, model_kwargs=kw, seed=2)      # warm
    torch.cuda.synchronize()
    ffi.profile_enable(True); ffi.profile_read()
    d.p_sample_loop(m, (B, L, D), clip_denoised=False, model_kwargs=kw, seed=2)
    torch.cuda.synchronize()
    prof = ffi.profile_read(); ffi.profile_enable(False)
    return prof


cfg0 = load_config("text_to_motion_contact_motion_gen", "cmdm", ["model.data_repr=h3d", "model.input_feats=263", "diffusion.steps=1000"])
m = create_model(cfg0, device=dev)
synth.fill_module_(m)
m = m.to(dev).eval()
full = dict(c_text_feat=synth.text_feature(32).to(dev), c_pc_xyz=synth.scene_cloud(32, NPTS).to(dev), c_pc_contact=synth.contact_map(32, NPTS).to(dev),
            x_mask=synth.frame_mask(32, L, all_valid=True).to(dev))
for B in (32, 4, 1):
    kw = {k: v[:B].contiguous() for k, v in full.items()}
    m.condition_tokens(**kw)
    p1, p2 = loop(B, K1, kw, m), loop(B, K2, kw, m)
    per = {}
    for name in p2:
        n = (p2[name]["launches"] - p1.get(name, {"launches": 0})["launches"]) / (K2 - K1)
        ms = (p2[name]["total_ms"] - p1.get(name, {"total_ms": 0.0})["total_ms"]) / (K2 - K1)
        if n > 0:
            per[name] = {"launches_per_step": round(n, 2), "us_per_launch": round(1e3 * ms / n, 2), "us_per_step": round(1e3 * ms, 1)}
    print(json.dumps({"B": B, "launches_per_step": round(sum(v["launches_per_step"] for v in per.values()), 2),
                      "kernel_us_per_step": round(sum(v["us_per_step"] for v in per.values()), 1), "kernels": per}), flush=True)
