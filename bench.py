#!/usr/bin/env python
"""Benchmark of the north-star hot path: denoising steps/sec of the CMDM (AMDM) `p_sample_loop`.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One "step" = one `p_sample` (CMDM denoiser forward over the rank's batch + DDPM posterior update) of
BASELINE.json configs[1]: B = 32 samples per GPU, L = 196 frames, D = 263, N = 8192 scene points
(128 contact-group tokens), hoisted step-invariant conditions, synthetic inputs, name-keyed random weights.
Weak scaling: every rank runs its own 32 samples (global sample indices rank*32 ...), one all_gather at
the end; value = (ranks x K steps) / max-over-ranks wall time.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "afford-motion_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

B_PER_GPU, L, D, NPTS = 32, 196, 263, 8192
F32_MFMA_PEAK_TFLOPS = 157.3            # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0          # same guide: v_mfma_f32_32x32x16_bf16, dense


def step_flops(batch: int, frames: int = L, groups: int = NPTS // 64) -> float:
    """Algorithmic FLOPs of one CMDM step: 5 encoder layers + motion adapters + time MLP."""
    T = 2 + groups + frames
    return batch * (5 * T * (4194304 + 2048 * T) + 2 * (2 * 263 * 512 * frames) + 2 * 2 * 512 * 512)


def pmc_traffic(kernel: str):
    """HBM bytes per launch of `kernel` from the rocprofv3 --pmc passes (FETCH_SIZE x 2 + WRITE_SIZE, corrected as
    MI355X_MICROARCH.md prescribes and calibrated on layernorm_kernel): written by tools/summarize_profiles.py into
    profiles/traffic.json on the GPU box; counters cannot be read from inside the process, so null when absent."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            tr = json.load(f)
    except (OSError, ValueError):
        return None
    want = kernel.replace(" ", "").rstrip(">")
    for name, v in tr.get("kernels", {}).items():        # rocprof names carry every template argument: match on the prefix
        if name.replace(" ", "").startswith(want):
            return dict(v, kernel=name, source=tr.get("source"))
    return None


def build(dev, steps_cfg: str):
    from afm import synth
    from afm.base import create_model_and_diffusion
    from afm.config import load_config
    cfg = load_config("text_to_motion_contact_motion_gen", "cmdm",
                      ["model.data_repr=h3d", "model.input_feats=263", "model.text_model.max_length=20", "diffusion.steps=1000",
                       f"diffusion.timestep_respacing='{steps_cfg}'"])
    model, diff = create_model_and_diffusion(cfg, device=dev)
    synth.fill_module_(model)
    return model.to(dev).eval(), diff, cfg


def cpu_baseline(n_steps: int = 3):
    """The reference's CPU path = its PyTorch-CPU math (oracle restatement, checked against the reference
    by tests/test_oracle_golden.py) on this host's cores, same B/L/T, conditions hoisted."""
    from afm import synth
    from oracle import denoiser_ref as dr, diffusion_ref as df, shapes as sh
    sd = sh.weights(sh.cmdm())
    x = synth.gaussian("bench_x", (B_PER_GPU, L, D))
    text, cont = synth.text_feature(B_PER_GPU), synth.gaussian("bench_cont", (B_PER_GPU, NPTS // 64, 256))
    mask = synth.frame_mask(B_PER_GPU, L, all_valid=True)
    s = df.Schedule(1000)
    model = lambda xx, t, **k: dr.cmdm_forward(sd, xx, t, text, x_mask=mask, cont_emb=cont)
    nz = synth.gaussian("bench_nz", (B_PER_GPU, L, D))
    t = torch.full((B_PER_GPU,), 500)
    # torch-CPU scales badly past the physical cores of one socket on these ops: probe a few thread
    # counts with one step each and report the FASTEST (the fairest CPU number we can produce here)
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cands = sorted({c for c in (avail, avail // 2, avail // 4, 64, 32, 16, 8) if 1 <= c <= avail}, reverse=True)
    best = (None, float("inf"))
    with torch.no_grad():
        for c in cands:
            torch.set_num_threads(c)
            df.p_sample(s, model, x, t, nz)                               # warm-up at this thread count
            t0 = time.perf_counter()
            df.p_sample(s, model, x, t, nz)
            d1 = time.perf_counter() - t0
            if d1 < best[1]:
                best = (c, d1)
            if d1 > 4 * best[1]:
                break
        torch.set_num_threads(best[0])
        t0 = time.perf_counter()
        for _ in range(n_steps):
            x = df.p_sample(s, model, x, t, nz)["sample"]
        dt = (time.perf_counter() - t0) / n_steps
    return {"value": round(1.0 / dt, 4), "unit": "steps/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n_steps} p_sample steps at B={B_PER_GPU}, L={L}, T=326 tokens, f32, conditions hoisted "
                      f"(torch-CPU restatement of the reference, {1e3 * dt:.0f} ms/step; best of thread counts {cands}, "
                      f"{avail} logical CPUs available)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--latency-runs", type=int, default=3, help="full 1000-step loops for the p50 sample latency (N=1 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt-gemm", action="store_true", help="skip the informational passes with the split-bf16 GEMM modes")
    ap.add_argument("--streams", type=int, default=None, help="sub-batch HIP streams of the native loop (default: model default, 2)")
    args = ap.parse_args()

    from afm import dist as adist, ffi, synth
    rank, world, local = adist.init_process_group()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py measures the HIP path: no GPU visible"
    # AFM_BENCH_SHARE_GPU=1 (testing only, with AFM_DIST_BACKEND=gloo): every rank uses cuda:0, to exercise the N > 1 control flow
    dev = torch.device("cuda:0" if os.environ.get("AFM_BENCH_SHARE_GPU") else f"cuda:{local}")
    torch.cuda.set_device(dev)
    ffi.load()

    K, W = args.steps, args.warmup
    model, diff_k, cfg = build(dev, str(K))
    if args.streams is not None:
        model.loop_streams = args.streams
    streams_default = model.loop_streams
    from afm.base import create_gaussian_diffusion
    cfg.diffusion.timestep_respacing = str(max(W, 1))
    diff_w = create_gaussian_diffusion(cfg)

    # synthetic batch of this rank (global sample indices rank*B ...), resident in HBM before any timing
    B = B_PER_GPU
    kw = dict(c_text_feat=synth.text_feature(B).to(dev), c_pc_xyz=synth.scene_cloud(B, NPTS).to(dev),
              c_pc_contact=synth.contact_map(B, NPTS).to(dev), x_mask=synth.frame_mask(B, L, all_valid=True).to(dev))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.condition_tokens(**kw)                      # one-off: SceneMapEncoder (FPS, kNN, set abstraction, attention)
    torch.cuda.synchronize()
    setup_ms = 1e3 * (time.perf_counter() - t0)          # cold: includes module load / first-launch costs
    kw2 = dict(kw, c_pc_xyz=kw["c_pc_xyz"].flip(0).contiguous(), c_pc_contact=kw["c_pc_contact"].flip(0).contiguous())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.condition_tokens(**kw2)                     # a different scene batch -> cache miss, warm kernels
    torch.cuda.synchronize()
    setup_ms_steady = 1e3 * (time.perf_counter() - t0)
    model.condition_tokens(**kw)

    def run(diffusion, seed, gather=True):
        x = diffusion.p_sample_loop(model, (B, L, D), clip_denoised=False, model_kwargs=kw, seed=seed, sample_index0=rank * B)
        if world > 1 and gather:                       # the path's only collective: gather the shards at the end
            out = [torch.empty_like(x) for _ in range(world)]
            dist.all_gather(out, x)
        return x

    if W > 0:
        run(diff_w, 1)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    run(diff_k, 2)                                     # exactly K steps
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()

    # roofline of the dominant kernel: the same K steps again with every launch bracketed by HIP events on its
    # stream.  Sub-batch streams are switched OFF for this pass so a launch's elapsed time is the kernel's own
    # (with 2 streams the other sub-batch's kernels share the GPU inside every bracket); same kernels, same
    # total work, launches of twice the rows.  `rocprofv3 ... bench.py --streams 1` reproduces these averages.
    roof = None
    if rank == 0:
        streams_timed = model.loop_streams
        model.loop_streams = 1
        run(diff_w, 1, gather=False)                  # rank 0 only: no collective in this pass
        ffi.profile_enable(True)
        ffi.profile_read()
        run(diff_k, 2, gather=False)
        prof = ffi.profile_read()
        ffi.profile_enable(False)
        model.loop_streams = streams_timed
        name = max(prof, key=lambda k: prof[k]["total_ms"]) if prof else None      # dominant kernel of the step
        g = prof.get(name)
        if g:
            ach = g["total_work"] / (g["total_ms"] * 1e-3) / 1e12
            roof = {"bound": "mfma", "kernel": name, "achieved": round(ach, 2), "peak": F32_MFMA_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(ach / F32_MFMA_PEAK_TFLOPS, 4), "traffic": pmc_traffic(name),
                    "avg_launch_us": round(1e3 * g["total_ms"] / g["launches"], 2), "launches": g["launches"],
                    "flops_per_launch": g["total_work"] / g["launches"],
                    "all_kernels_ms_per_step": {k: round(v["total_ms"] / K, 4) for k, v in prof.items()}}
            if "split_bf16" in name:
                # f32 results computed as 9 exact bf16 x bf16 products per f32 product on the bf16 matrix pipe: `achieved` / `peak` above
                # are ALGORITHMIC f32 FLOPs against the f32 MFMA peak (the dtype's peak); the pipe actually used is priced here
                from afm import ops as afm_ops
                nprod = afm_ops.get_gemm_split()[0]
                roof["matrix_pipe"] = {"instruction": "v_mfma_f32_32x32x16_bf16", "bf16_products_per_f32_product": nprod,
                                       "issued_tflops": round(ach * nprod, 1), "peak": BF16_MFMA_PEAK_TFLOPS,
                                       "frac": round(ach * nprod / BF16_MFMA_PEAK_TFLOPS, 4)}

    lat = None
    if rank == 0 and world == 1 and args.latency_runs > 0:
        cfg.diffusion.timestep_respacing = ""
        diff_full = create_gaussian_diffusion(cfg)
        ts = []
        for i in range(args.latency_runs):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(diff_full, 10 + i)
            torch.cuda.synchronize()
            ts.append(1e3 * (time.perf_counter() - t0))
        lat = {"p50_ms": round(statistics.median(ts), 1), "runs": args.latency_runs, "steps": 1000, "batch": B}

    # informational only (never `value`): the same K steps with afm_linear's other arithmetic settings.  Default (the timed run above):
    # the exact nine-product bf16x3 split on the wide GEMMs (N >= 1024), native f32 MFMA kernels elsewhere.
    alt = None
    if rank == 0 and world == 1 and not args.no_alt_gemm:
        from afm import ops as afm_ops
        alt = {}
        saved = afm_ops.get_gemm_split()
        try:
            for tag, (products, min_n) in (("native_f32_mfma_everywhere", (0, 0)), ("split9_all_gemms", (9, 0)),
                                           ("split6_wide_gemms", (6, 1024)), ("split6_all_gemms", (6, 0))):
                afm_ops.set_gemm_split(products, min_n)
                run(diff_w, 1)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                run(diff_k, 2)
                torch.cuda.synchronize()
                alt[tag] = round(K / (time.perf_counter() - t0), 2)
        finally:
            afm_ops.set_gemm_split(*saved)
        alt["unit"] = "steps/s"

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

    if rank == 0:
        ms = 1e3 * dt / K
        line = {
            "metric": "denoising steps/sec (B=32, L=196, N=8192)", "value": round(world * K / dt, 2), "unit": "steps/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(ms, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "CMDM trans_enc p_sample_loop, HumanML3D t2m_contact_motion config (BASELINE configs[1])",
                       "gemm_arithmetic": "f32 in / f32 accumulate; N >= 1024 GEMMs: exact 3-way bf16 operand split, all 9 cross products on the bf16 MFMA pipe; others: f32 MFMA",
                       "batch_per_gpu": B, "frames": L, "motion_dim": D, "scene_points": NPTS, "tokens": 2 + NPTS // 64 + L,
                       "conditions": "hoisted (step-invariant, computed once: setup_ms)", "parallelism": f"batch-shard x{world}", "sub_batch_streams": model.loop_streams},
            "algorithmic_tflops": round(step_flops(B) * world * K / dt / 1e12, 2),
            "setup_ms": round(setup_ms, 2), "setup_ms_steady": round(setup_ms_steady, 2),
            "roofline": roof, "cpu_baseline": cpu, "sample_latency": lat, "alt_gemm_modes": alt,
        }
        print(json.dumps(line))
    if world > 1:
        dist.barrier()                                 # rank 0 is still in its roofline pass: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
