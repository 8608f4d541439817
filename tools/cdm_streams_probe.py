#!/usr/bin/env python
"""Round 4 experiment: the CDM sampling loop (configs[2], B = 32, N = 8192, H3D variant) with sub-batch streams, the 2-latent chain on a
side stream, CU-masked streams and other dec_point chunkings - all variants bit-identical, one JSON line each, ONE process.
    python tools/cdm_streams_probe.py [steps] > gpurun_out/.../cdm_streams.jsonl"""
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
B, N = 32, 8192
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
cfg = load_config("text_to_motion_contact_gen", "cdm", ["model.arch=Perceiver", "model.input_feats=6", "model.text_model.max_length=20", "diffusion.steps=500",
                                                       f"diffusion.timestep_respacing='{steps}'", "model.scene_model.use_scene_model=False"])
m = create_model(cfg, device=dev)
synth.fill_module_(m)
m = m.to(dev).eval()
d = create_gaussian_diffusion(cfg)
kw = dict(c_text_feat=synth.text_feature(B).to(dev), c_pc_xyz=synth.scene_cloud(B, N).to(dev))
ref = None


def words(bits):
    """CU mask words with the CUs `bits` (iterable of indices) enabled; 256 CUs = 8 words."""
    w = [0] * 8
    for b in bits:
        w[b >> 5] |= 1 << (b & 31)
    return w


ALL = range(256)
MASKS = {
    "low32": words(range(32)), "not_low32": words(range(32, 256)),                                # 32 consecutive bits
    "stride8": words(range(0, 256, 8)), "not_stride8": words(b for b in ALL if b % 8),            # every 8th bit (one per XCD if bits interleave XCDs)
    "low16": words(range(16)), "not_low16": words(range(16, 256)),
    "low64": words(range(64)), "not_low64": words(range(64, 256)),
}


def measure(tag, **attrs):
    global ref
    base = dict(loop_sub_batches=1, chain_side=False, dec_chunks=0, chain_cu_mask=None, point_cu_mask=None)
    base.update(attrs)
    for k, v in base.items():
        setattr(m, k, MASKS[v] if isinstance(v, str) else v)
    try:
        run = lambda: d.p_sample_loop(m, (B, N, 6), clip_denoised=False, model_kwargs=kw, seed=1)
        out = run(); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); run(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / steps)
        dt = sorted(ts)[1]
        if ref is None:
            ref = out.clone()
        print(json.dumps({"variant": tag, "steps_per_s": round(1 / dt, 1), "us_per_step": round(1e6 * dt, 1), "bit_identical": bool(torch.equal(out, ref)),
                          "attrs": {k: v for k, v in attrs.items()}}), flush=True)
    except Exception as e:      # noqa: BLE001
        print(json.dumps({"variant": tag, "error": f"{type(e).__name__}: {e}"}), flush=True)


measure("one stream (default)")
measure("one stream, 32 dec_point chunks", dec_chunks=32)
measure("one stream, 24 dec_point chunks", dec_chunks=24)
for ns in (2, 3, 4):
    measure(f"{ns} sub-batches, chain in line", loop_sub_batches=ns)
    measure(f"{ns} sub-batches, chain on the side stream", loop_sub_batches=ns, chain_side=True)
    measure(f"{ns} sub-batches, chain on the side stream, {16 * ns} chunks", loop_sub_batches=ns, chain_side=True, dec_chunks=min(63, 16 * ns))
for cm, pm in (("low32", None), ("low32", "not_low32"), ("stride8", None), ("stride8", "not_stride8"), ("low16", "not_low16"), ("low64", "not_low64")):
    for ns in (2, 3):
        for ch in (0, 16 * ns, 14 * ns):
            measure(f"{ns} sub-batches, chain side stream on CUs {cm}, point streams {pm or 'unmasked'}, chunks {ch or 16}", loop_sub_batches=ns, chain_side=True,
                    chain_cu_mask=cm, point_cu_mask=pm, dec_chunks=ch)
measure("one stream (default) again")
