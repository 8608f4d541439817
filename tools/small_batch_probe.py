#!/usr/bin/env python
"""Round 4: the CMDM native loop at the per-GPU batch sizes of the strong-scaling job (B = 8 / 4 / 2 / 1), sub-batch streams 1 / 2 / 4,
one process, one JSON line per setting (steps/s, us per step, host enqueue).
    python tools/small_batch_probe.py [steps]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "afford-motion_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

from afm import synth  # noqa: E402
from afm.base import create_gaussian_diffusion, create_model  # noqa: E402
from afm.config import load_config  # noqa: E402

dev = torch.device("cuda:0")
L, D, NPTS = 196, 263, 8192
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
cfg = load_config("text_to_motion_contact_motion_gen", "cmdm", ["model.data_repr=h3d", "model.input_feats=263", "diffusion.steps=1000",
                                                               f"diffusion.timestep_respacing='{steps}'"])
m = create_model(cfg, device=dev)
synth.fill_module_(m)
m = m.to(dev).eval()
d = create_gaussian_diffusion(cfg)
d.tables(dev)
full = dict(c_text_feat=synth.text_feature(32).to(dev), c_pc_xyz=synth.scene_cloud(32, NPTS).to(dev), c_pc_contact=synth.contact_map(32, NPTS).to(dev),
            x_mask=synth.frame_mask(32, L, all_valid=True).to(dev))
ref = {}
for B in (32, 8, 4, 2, 1):
    kw = {k: v[:B].contiguous() for k, v in full.items()}
    m.condition_tokens(**kw)
    for ns in ((1, 2, 3) if B == 32 else (1, 2, 4)):
        if ns > B:
            continue
        m.loop_streams, m.loop_streams_auto = ns, False
        run = lambda: d.p_sample_loop(m, (B, L, D), clip_denoised=False, model_kwargs=kw, seed=2)
        out = run(); torch.cuda.synchronize()
        ts, te = [], []
        for _ in range(3):
            t0 = time.perf_counter(); run(); t1 = time.perf_counter(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / steps); te.append((t1 - t0) / steps)
        dt = sorted(ts)[1]
        ref.setdefault(B, out.clone())
        print(json.dumps({"B": B, "streams": ns, "steps_per_s": round(1 / dt, 1), "us_per_step": round(1e6 * dt, 1), "host_enqueue_us_per_step": round(1e6 * sorted(te)[1], 1),
                          "bit_identical_to_one_stream": bool(torch.equal(out, ref[B]))}), flush=True)
