#!/usr/bin/env python
"""CDM sampling loop (BASELINE configs[2]: B = 32, N = 8192, H3D variant with 9 input channels) under the measurement attributes of
afm.cdm.CDM, plus the HUMANISE variant (41 input channels = contact 6 + 32 scene features of the hoisted frozen backbone + xyz 3,
reference models/cdm.py:444-446,508), one JSON line per variant, all in ONE process (boxes differ by +-8 %).
    python tools/cdm_ab.py [steps] > profiles/rNN_cdm_ab.jsonl"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "afford-motion_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

from afm import ffi, synth  # noqa: E402
from afm.base import create_gaussian_diffusion, create_model  # noqa: E402
from afm.config import load_config  # noqa: E402

dev = torch.device("cuda:0")
B, N = 32, 8192
steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 100


def build(feats: bool):
    over = ["model.arch=Perceiver", "model.input_feats=6", "model.text_model.max_length=20", "diffusion.steps=500", f"diffusion.timestep_respacing='{steps}'"]
    over += (["model.scene_model.use_scene_model=True", "model.scene_model.use_openscene=True", "model.scene_model.point_feat_dim=32",
              "model.scene_model.pretrained_weight=''", "task.dataset.use_openscene=True"] if feats else ["model.scene_model.use_scene_model=False"])
    cfg = load_config("text_to_motion_contact_gen", "cdm", over)
    m = create_model(cfg, device=dev)
    synth.fill_module_(m)
    return m.to(dev).eval(), create_gaussian_diffusion(cfg)


def measure(m, d, kw, tag, **attrs):
    for k, v in attrs.items():
        setattr(m, k, v)
    run = lambda: d.p_sample_loop(m, (B, N, 6), clip_denoised=False, model_kwargs=kw, seed=1)
    run(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); run(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / steps)
    dt = sorted(ts)[1]
    ffi.profile_enable(True); ffi.profile_read(); run(); prof = ffi.profile_read(); ffi.profile_enable(False)
    print(json.dumps({"variant": tag, "steps_per_s": round(1 / dt, 1), "ms_per_step": round(1e3 * dt, 4), "attrs": attrs,
                      "kernels_ms_per_step": {k: round(v["total_ms"] / steps, 4) for k, v in prof.items()}}), flush=True)


m, d = build(False)
kw = dict(c_text_feat=synth.text_feature(B).to(dev), c_pc_xyz=synth.scene_cloud(B, N).to(dev))
base = dict(no_gen=False, loop_sub_batches=1, no_fold=False, gemm_tile=0, pipeline=False, dec_chunks=0)
measure(m, d, kw, "H3D default: no per-point rows (enc_point / lat_head / lat_dectables / dec_point), one stream", **base)
measure(m, d, kw, "H3D default form, two sub-batch streams", **dict(base, loop_sub_batches=2)) if "--all" in sys.argv else None
for nsub, chunks in ((2, 0), (2, 24), (2, 32), (3, 0), (3, 23)):
    measure(m, d, kw, f"H3D default form, PIPELINE of {nsub} sub-batches (point kernels round-robin on one stream, chains on side streams), dec chunks {chunks or 16}",
            **dict(base, loop_sub_batches=nsub, pipeline=True, dec_chunks=chunks))
if "--all" in sys.argv:
    measure(m, d, kw, "H3D folded rows (round 2 form; linear1 weight-stationary), one stream", **dict(base, no_gen=True))
    measure(m, d, kw, "H3D folded rows, linear1 on staged 64x64 tiles", **dict(base, no_gen=True, gemm_tile=3))
del m
mh, dh = build(True)
kwh = dict(kw, c_pc_feat=synth.gaussian("cdm_ab_feat", (B, N, 32)).to(dev))
measure(mh, dh, kwh, "HUMANISE variant (41 input channels, backbone features hoisted): no per-point rows (K = 44 inputs), one stream", **base)
if "--all" in sys.argv:
    measure(mh, dh, kwh, "HUMANISE variant, two sub-batch streams", **dict(base, loop_sub_batches=2))
for nsub, chunks in ((2, 0), (3, 0), (3, 23)):
    measure(mh, dh, kwh, f"HUMANISE variant, PIPELINE of {nsub} sub-batches, dec chunks {chunks or 16}", **dict(base, loop_sub_batches=nsub, pipeline=True, dec_chunks=chunks))
if "--all" in sys.argv:
    measure(mh, dh, kwh, "HUMANISE variant, folded rows (round 2 form), one stream", **dict(base, no_gen=True))
