#!/usr/bin/env python
"""Where does the ~45 ms stall of a fresh process land?  Consecutive synchronised 5-step loops from the first GPU work on; wall time of each."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "afford-motion_amd")):
    sys.path.insert(0, p)
import torch
from afm import ffi, synth
from afm.base import create_gaussian_diffusion, create_model
from afm.config import load_config
dev = torch.device("cuda:0")
B, L, D, N = int(os.environ.get("PROBE_B", 32)), 196, 263, 8192
cfg_for = lambda k: load_config("text_to_motion_contact_motion_gen", "cmdm", ["model.data_repr=h3d", "model.input_feats=263", f"diffusion.timestep_respacing='{k}'"])
model = create_model(cfg_for(20), device=dev); synth.fill_module_(model); model = model.to(dev).eval()
kw = dict(c_text_feat=synth.text_feature(B).to(dev), c_pc_xyz=synth.scene_cloud(B, N).to(dev), c_pc_contact=synth.contact_map(B, N).to(dev),
          x_mask=synth.frame_mask(B, L, all_valid=True).to(dev))
mode = sys.argv[1] if len(sys.argv) > 1 else "loops"
t_start = time.perf_counter()
model.condition_tokens(**kw); torch.cuda.synchronize()
print(f"[{mode}] condition tokens (first GPU work): {1e3 * (time.perf_counter() - t_start):.1f} ms")
d5 = create_gaussian_diffusion(cfg_for(5)); d5.tables(dev)
if mode == "sleep":
    time.sleep(0.5)
if mode == "noscratch":          # 8-wave attention groups: no kernel of the loop uses scratch memory (the 12-wave form parks 84 B per lane)
    model.attn_group_waves = 8
if mode.startswith("heavy"):     # N ms of an unrelated heavy kernel first (a large GEMM through afm_linear, back to back)
    from afm import ops
    a_ = torch.randn(10432, 512, device=dev); w_ = torch.randn(1536, 512, device=dev); o_ = torch.empty(10432, 1536, device=dev)
    n_ = int(mode[5:] or 150)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < n_ * 1e-3:
        for _ in range(20):
            ops.linear(a_, w_, out=o_)
        torch.cuda.synchronize()
    print(f"[{mode}] heavy GEMMs for {1e3 * (time.perf_counter() - t0):.0f} ms")
if mode.startswith("prestreams"):   # the loop's side streams created and used (one tiny kernel each) up front, then idle / busy for a while
    z = torch.zeros(1024, device=dev)
    model._side_streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    for st_ in model._side_streams:
        ffi.load().afm_clamp(z.data_ptr(), 1024, -1.0, 1.0, st_.cuda_stream)
    torch.cuda.synchronize()
    time.sleep(float(mode[10:] or 0.3))
    print(f"[{mode}] side streams created and touched, waited")
if mode == "one_stream":
    model.loop_streams, model.loop_streams_auto = 1, False
if mode == "trivial":            # 3000 launches of an unrelated tiny kernel on the main stream first
    z = torch.zeros(1024, device=dev)
    for _ in range(3000):
        ffi.load().afm_clamp(z.data_ptr(), 1024, -1.0, 1.0, ffi.stream_of(z))
    torch.cuda.synchronize()
    print(f"[{mode}] 3000 tiny launches done at {1e3 * (time.perf_counter() - t_start):.0f} ms")
if mode == "long2":              # the second call is a 40-step loop: does the stall move into it?
    d40 = create_gaussian_diffusion(cfg_for(40)); d40.tables(dev)
ts = []
for i in range(int(os.environ.get("PROBE_LOOPS", 14))):
    if mode == "gap" and i == 1:
        time.sleep(0.3)                      # idle after the first loop: does the stall pass unseen?
    if mode == "long2" and i == 1:
        t0 = time.perf_counter(); d40.p_sample_loop(model, (B, L, D), clip_denoised=False, model_kwargs=kw, seed=2); torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0)); continue
    t0 = time.perf_counter()
    d5.p_sample_loop(model, (B, L, D), clip_denoised=False, model_kwargs=kw, seed=2)
    torch.cuda.synchronize()
    ts.append(1e3 * (time.perf_counter() - t0))
print(f"[{mode}, B={B}] 14 consecutive 5-step loops, ms each: " + " ".join(f"{t:.1f}" for t in ts) + f"   (since first GPU work: {1e3 * (time.perf_counter() - t_start):.0f} ms)")
