#!/usr/bin/env python
"""Training-step measurement for the CMDM trunk: one optimisation step =
training_losses forward + loss.mean().backward() + AdamW over the trainable parameters (utils/training.py:140-155),
B = 32, L = 196, train mode (all dropouts on), SceneMapEncoder frozen (its output is a step input here).
One JSON object; per-kernel times from the library's HIP-event profiler.

    python tools/bench_train.py [--steps 20] [--batch 32] [--cpu-steps 1]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "afford-motion_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

from afm import autograd as AG  # noqa: E402
from afm import ffi, synth  # noqa: E402
from afm.base import create_model_and_diffusion  # noqa: E402
from afm.config import load_config  # noqa: E402


def cpu_baseline(B, L, steps):
    """The same optimisation step on the host cores: torch autograd over the oracle's restatement of the reference."""
    from oracle import diffusion_ref as df
    from oracle import shapes as sh
    from oracle import train_ref as tr
    sd = sh.weights(sh.cmdm())
    x0, tn = synth.gaussian("bt_x0", (B, L, 263)), synth.gaussian("bt_noise", (B, L, 263))
    text, cont = synth.text_feature(B), synth.gaussian("bt_cont", (B, 128, 256)) * 0.5
    t = torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(0))
    sched = df.Schedule(1000)
    best = None
    for nt in (64, 32, 16):
        torch.set_num_threads(nt)
        t0 = time.perf_counter()
        for _ in range(steps):
            tr.cmdm_loss_and_grads(sd, sched, x0, t, tn, text, cont, synth.frame_mask(B, L))
        dt = (time.perf_counter() - t0) / steps
        if best is None or dt < best[0]:
            best = (dt, nt)
    return best


def main_cdm(a, dev):
    """ADM: CDM Perceiver (text_to_motion_contact_gen config: 500 steps, use_scene_model=False) over N = 8192 points."""
    B, N = a.batch, 8192
    cfg = load_config("text_to_motion_contact_gen", "cdm", ["model.arch=Perceiver", "model.scene_model.use_scene_model=False",
                                                           "model.input_feats=6", "model.text_model.max_length=20", "diffusion.steps=500"])
    model, diff = create_model_and_diffusion(cfg, device=dev)
    synth.fill_module_(model)
    model = model.to(dev).train()
    params = [p for n, p in model.named_parameters() if p.requires_grad and not n.startswith("text_model")]
    x0 = synth.gaussian("bt_cdm_x0", (B, N, 6)).to(dev)
    kw = dict(c_text_feat=synth.text_feature(B).to(dev), c_pc_xyz=synth.scene_cloud(B, N).to(dev))
    state = {}
    gen = torch.Generator(device="cpu").manual_seed(0)

    def step():
        t = torch.randint(0, diff.num_timesteps, (B,), generator=gen).to(dev)
        for p in params:
            p.grad = None
        loss = diff.training_losses(model, x0, t, model_kwargs=kw)["loss"].mean()
        loss.backward()
        AG.adamw_step(params, state, lr=1e-4)
        return loss
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    ffi.profile_enable(True); ffi.profile_read()
    for _ in range(3):
        step()
    prof = ffi.profile_read(); ffi.profile_enable(False)
    out = {"config": f"CDM (Perceiver) training step, B={B}, N={N} points, f32, train mode (attention dropout on), 1 MI355X",
           "metric": "optimisation steps/sec", "value": round(1 / dt, 3), "ms_per_step": round(1e3 * dt, 3), "samples_per_sec": round(B / dt, 1),
           "final_loss": round(loss.item(), 4), "as_written_tflops": round(3 * 313.4e9 * B / 32 / dt / 1e12, 1),
           "trainable_params": sum(p.numel() for p in params),
           "kernels_ms_per_step": {k: round(v["total_ms"] / 3, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])}}
    print(json.dumps(out))


def measure_cmdm(B, L, scene, steps, warmup, cpu_steps, dev):
    """One training-step measurement of the AMDM (trunk, or trunk + SceneMapEncoder with `scene`) -> the JSON object of this tool; also the
    `training` entry of bench.py's `secondary` block (tools/bench_configs.py::training_block)."""
    cfg = load_config("text_to_motion_contact_motion_gen", "cmdm", ["model.data_repr=h3d", "model.input_feats=263", "diffusion.steps=1000"])
    model, diff = create_model_and_diffusion(cfg, device=dev)
    synth.fill_module_(model)
    model = model.to(dev).train()
    if not scene:
        model.contact_encoder.requires_grad_(False)
    params = [p for p in model.parameters() if p.requires_grad]
    x0 = synth.gaussian("bt_x0", (B, L, 263)).to(dev)
    kw = dict(c_text_feat=synth.text_feature(B).to(dev), x_mask=synth.frame_mask(B, L).to(dev))
    if scene:
        kw.update(c_pc_xyz=synth.scene_cloud(B, 8192).to(dev), c_pc_contact=synth.contact_map(B, 8192).to(dev))
    else:
        kw.update(c_cont_emb=(synth.gaussian("bt_cont", (B, 128, 256)) * 0.5).to(dev))
    state = {}
    gen = torch.Generator(device="cpu").manual_seed(0)

    def step():
        t = torch.randint(0, diff.num_timesteps, (B,), generator=gen).to(dev)
        for p in params:
            p.grad = None
        terms = diff.training_losses(model, x0, t, model_kwargs=kw)
        loss = terms["loss"].mean()
        loss.backward()
        AG.adamw_step(params, state, lr=1e-4)
        return loss

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ffi.profile_enable(True); ffi.profile_read()
    for _ in range(3):
        step()
    prof = ffi.profile_read(); ffi.profile_enable(False)
    T = 2 + 128 + L
    M = B * T
    d, ff = 512, 1024
    gemm = 2.0 * M * d * (3 * d + d + 2 * ff) * 5 + 2.0 * B * L * 263 * d * 2
    attn = 4.0 * B * 8 * T * T * 64 * 5
    flops = 3 * gemm + attn * (1 + 7 / 2)          # fwd + dX + dW GEMMs; attention fwd (2 products) + bwd (7 products)
    what = "CMDM full-model (trunk + SceneMapEncoder over N=8192 points)" if scene else "CMDM trunk"
    out = {"config": f"{what} training step, B={B}, L={L}, T={T} tokens, f32, train mode (dropout on), 1 MI355X",
           "metric": "optimisation steps/sec", "value": round(1 / dt, 3), "ms_per_step": round(1e3 * dt, 3), "samples_per_sec": round(B / dt, 1),
           "final_loss": round(loss.item(), 4), "executed_tflops": round(flops / dt / 1e12, 1),
           "trainable_params": sum(p.numel() for p in params),
           "kernels_ms_per_step": {k: round(v["total_ms"] / 3, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])},
           "kernels_tflops": {k: round(v["total_work"] / (v["total_ms"] * 1e-3) / 1e12, 1) for k, v in prof.items() if v["total_work"] > 0 and v["total_ms"] > 0}}
    if cpu_steps > 0 and not scene:
        cdt, nt = cpu_baseline(B, L, cpu_steps)
        out["cpu_baseline"] = {"value": round(1 / cdt, 4), "unit": "steps/s", "cores": nt, "kind": "port",
                               "sample": f"{cpu_steps} forward+backward step(s) of the oracle restatement (torch autograd, f32), same B/L"}
    # roofline of the step's dominant kernel (by time, among the kernels whose work is counted): f32-equivalent FLOPs against the f32 MFMA peak
    counted = {k: v for k, v in prof.items() if v["total_work"] > 0 and v["total_ms"] > 0}
    if counted:
        k = max(counted, key=lambda n: counted[n]["total_ms"])
        ach = counted[k]["total_work"] / (counted[k]["total_ms"] * 1e-3) / 1e12
        out["roofline"] = {"bound": "mfma", "kernel": k, "achieved": round(ach, 2), "peak": 157.3, "unit": "TFLOP/s", "frac": round(ach / 157.3, 4),
                           "avg_launch_us": round(1e3 * counted[k]["total_ms"] / counted[k]["launches"], 2), "launches_per_step": counted[k]["launches"] // 3, "traffic": None}
    out["step_frac_of_f32_mfma_peak"] = round(flops / dt / 1e12 / 157.3, 4)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=196)
    ap.add_argument("--cpu-steps", type=int, default=1)
    ap.add_argument("--cdm", action="store_true", help="train the ADM (CDM Perceiver over N=8192 points) instead of the AMDM")
    ap.add_argument("--scene", action="store_true", help="train the SceneMapEncoder too (N=8192 points per sample, batch-statistics BatchNorm)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    if a.cdm:
        return main_cdm(a, dev)
    print(json.dumps(measure_cmdm(a.batch, a.frames, a.scene, a.steps, a.warmup, a.cpu_steps, dev)))


if __name__ == "__main__":
    main()
