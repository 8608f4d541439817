#!/usr/bin/env python
"""Benchmark of the north-star hot path: denoising steps/sec of the CMDM (AMDM) `p_sample_loop`.

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1 without a launcher environment: re-executes itself under the line below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One "step" = one `p_sample` (CMDM denoiser forward over the rank's batch + DDPM posterior update) of
BASELINE.json configs[1]: B = 32 samples per GPU, L = 196 frames, D = 263, N = 8192 scene points
(128 contact-group tokens), hoisted step-invariant conditions, synthetic inputs, name-keyed random weights.
`--scaling strong` (the default for N > 1: BASELINE configs[4] is "k_sample = 32 sharded 8x"): ONE 32-sample job (k_sample = 32,
reference test.py:88-101) is sharded over the ranks (32 / N samples per GPU), one all_gather at the end, value = K / max-over-ranks
wall time.  `--scaling weak`: every rank runs its own 32 samples (global sample indices rank*32 ...), value = (ranks x K steps) /
max-over-ranks wall time.  At N = 1 the two are the same job; with N > 1 the line carries the OTHER mode as well
(`other_scaling_mode`).  Rank 0 prints ONE JSON line.  At N = 1 the line also carries `secondary`: the other BASELINE configs
([0], [2], [3], [4]), the headline shape with the conditions recomputed every step ("faithful") and the per-GPU batch sizes of the
strong-scaling job (B = 16 / 8 / 4 / 1), all measured after the headline and outside its timed region.
"""
import argparse
import json
import os
import statistics
import sys
import time

# multi-process GPU work on this host driver needs dmabuf IPC (RCCL / device-tensor sharing fail with `hipIpcGetMemHandle: invalid argument`
# otherwise); the launcher normally exports it - keep it if it does not.  Read by the HSA runtime at its initialisation: before `import torch`.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

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


def elided_flops(batch: int, frames: int = L, groups: int = NPTS // 64) -> float:
    """FLOPs of the algorithmic step that the loop does NOT execute (bit-neutral): after the LAST layer's attention only the motion
    rows are used (out_proj + FFN on the other 2 + groups rows per sample are skipped), and layer 0's in_proj rows of the 1 + groups
    step-invariant condition tokens are computed on the first step of a loop only."""
    other_rows = 2 + groups
    last_layer = other_rows * (2 * 512 * 512 + 2 * 2 * 512 * 1024)
    layer0_qkv = (1 + groups) * 2 * 512 * 1536
    return batch * float(last_layer + layer0_qkv)


def pmc_traffic(kernel: str):
    """HBM bytes per launch of `kernel` from the rocprofv3 --pmc passes (FETCH_SIZE x 2 + WRITE_SIZE, corrected as
    MI355X_MICROARCH.md prescribes and calibrated on layernorm_kernel): written by tools/summarize_profiles.py into
    profiles/traffic.json on the GPU box; counters cannot be read from inside the process, so null when absent."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            tr = json.load(f)
    except (OSError, ValueError):
        return None
    want = kernel.replace(" ", "").replace(",split-K", "").rstrip(">")
    split_k = "split-K" in kernel
    # rocprof names carry every template argument: match on the prefix; the bf16-split kernel's template arguments are <BM, BN, BK, products,
    # K groups, LDS stages, segments per group> (round 5) - K groups == 1 is the large-launch form, > 1 the split-K small-launch forms: keep the
    # two apart, then take the most launched
    cands = []
    for name, v in tr.get("kernels", {}).items():
        n = name.replace(" ", "")
        if not n.startswith(want):
            continue
        if "split_bf16<" in n:
            targs = n[n.index("<") + 1:].rstrip(">").split(",")
            kg = int(targs[4]) if len(targs) > 4 and targs[4].isdigit() else 1
            if (kg > 1) != split_k:
                continue
        cands.append((v.get("launches", 0), name, v))
    if not cands:
        return None
    _, name, v = max(cands, key=lambda c: c[0])
    return dict(v, kernel=name, source=tr.get("source"))


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


def cpu_baseline(n_steps: int = 20, reps: int = 3):
    """The reference's CPU path = its PyTorch-CPU math (oracle restatement, checked against the reference
    by tests/test_oracle_golden.py) on this host's cores, same B/L/T, conditions hoisted: `reps` repetitions of
    `n_steps` chained p_sample steps at the fastest thread count (the faithful variant - contact encoder
    recomputed every step - is timed at configs[0]'s size by tools/bench_configs.py)."""
    from afm import synth
    from oracle import denoiser_ref as dr, diffusion_ref as df, shapes as sh
    sd = sh.weights(sh.cmdm())
    x0 = synth.gaussian("bench_x", (B_PER_GPU, L, D))
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
    probe = {}                                         # thread count -> ms per step (best of two single steps after a warm-up step): the whole probe is reported
    with torch.no_grad():
        for c in cands:
            torch.set_num_threads(c)
            df.p_sample(s, model, x0, t, nz)                              # warm-up at this thread count
            d1 = float("inf")
            for _ in range(2):
                t0 = time.perf_counter()
                df.p_sample(s, model, x0, t, nz)
                d1 = min(d1, time.perf_counter() - t0)
            probe[c] = round(1e3 * d1, 1)
            if d1 < best[1]:
                best = (c, d1)
            if d1 > 4 * best[1]:
                break
        torch.set_num_threads(best[0])
        per_rep = []
        for _ in range(reps):
            x = x0
            t0 = time.perf_counter()
            for _ in range(n_steps):
                x = df.p_sample(s, model, x, t, nz)["sample"]
            per_rep.append((time.perf_counter() - t0) / n_steps)
    dt = statistics.median(per_rep)
    try:
        cpu_model = next(ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name"))
    except (OSError, StopIteration):
        cpu_model = "unknown"
    return {"value": round(1.0 / dt, 4), "unit": "steps/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{reps} x {n_steps} chained p_sample steps at B={B_PER_GPU}, L={L}, T=326 tokens, f32, conditions hoisted "
                      f"(torch-CPU restatement of the reference; median {1e3 * dt:.0f} ms/step, per repetition "
                      f"{[round(1e3 * v) for v in per_rep]} ms/step; thread-count probe, ms per single step: {probe}; {avail} logical CPUs available)",
            "thread_probe_ms_per_step": {str(k): v for k, v in probe.items()},
            "host": {"cpu": cpu_model, "logical_cpus": avail, "torch": torch.__version__, "threads_used": torch.get_num_threads()}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=B_PER_GPU, help="samples per GPU (weak) / in the whole job (strong); the headline is 32")
    ap.add_argument("--scaling", choices=("auto", "weak", "strong"), default=os.environ.get("AFM_BENCH_SCALING", "auto"),
                    help="strong: ONE --batch-sample job (k_sample = 32 of test.py:88-101, BASELINE configs[4]) sharded over the GPUs, value = "
                         "steps/s of that job; weak: --batch samples on EVERY GPU; auto (default): strong when N > 1 (at N = 1 both are the same job)")
    ap.add_argument("--settle-s", type=float, default=0.0, help="idle time after the setup's priming loop call (process-start transient, see the comment at its use); 0 = none")
    ap.add_argument("--no-secondary", action="store_true", help="skip the `secondary` block (other BASELINE configs, faithful headline, small-batch table; N = 1 only)")
    ap.add_argument("--latency-runs", type=int, default=20, help="full 1000-step loops at B=32 for the p50 sample latency (N=1 only; SURVEY 8d: >= 20 runs after 3 warm-ups)")
    ap.add_argument("--latency-warmups", type=int, default=3, help="untimed full 1000-step loops in front of the latency runs of each batch size")
    ap.add_argument("--latency-runs-b1", type=int, default=20, help="full 1000-step loops at B=1 for the p50 sample latency (N=1 only)")
    ap.add_argument("--cpu-steps", type=int, default=20, help="p_sample steps per repetition of the CPU baseline")
    ap.add_argument("--cpu-reps", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt-gemm", action="store_true", help="skip the informational passes with the other GEMM arithmetic settings")
    ap.add_argument("--streams", type=int, default=None, help="sub-batch HIP streams of the native loop (default: model default)")
    ap.add_argument("--no-ln-fold", action="store_true", help="measurement: separate LayerNorm launches instead of the statistics-carrying GEMM epilogues")
    ap.add_argument("--fused-ln", action="store_true", help="measurement: norm1 / norm2 inside the out_proj / linear2 GEMMs (bit-identical; slower, profiles/r03_ln_fusion.md)")
    ap.add_argument("--gemm-tile", type=int, default=0, help="measurement: AFM_TUNE_TILE code forced on the wide encoder GEMMs (5 = 128x128; bit-neutral)")
    ap.add_argument("--pair-launch", type=int, default=None, help="1 / 0: sub-batch A's out_proj + sub-batch B's linear1 as one 128x128-tile launch per layer (bit-identical; default: the model's setting)")
    ap.add_argument("--attn-group", type=int, default=None, help="waves per attention workgroup (bit-neutral tuning; default: library choice)")
    args = ap.parse_args()

    from afm import dist as adist
    if adist.needs_self_launch(args.gpus) and not os.environ.get("AFM_SELF_LAUNCHED"):
        # plain `python bench.py --gpus N` (no torch.distributed.run environment): launch the N ranks ourselves, exactly as the documented
        # torchrun form does (one process per GPU, loopback rendezvous on a free port - the reference's convention,
        # scripts/t2m_contact_motion/train_ddp.sh:9); rank 0's JSON line is this process's output, its exit code ours
        sys.exit(adist.self_launch(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    from afm import ffi, synth
    rank, world, local = adist.init_process_group()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if args.scaling == "auto":
        args.scaling = "strong" if world > 1 else "weak"
    assert torch.cuda.is_available(), "bench.py measures the HIP path: no GPU visible"
    # AFM_BENCH_SHARE_GPU=1 (testing only, with AFM_DIST_BACKEND=gloo): every rank uses cuda:0, to exercise the N > 1 control flow
    dev = torch.device("cuda:0" if os.environ.get("AFM_BENCH_SHARE_GPU") else f"cuda:{local}")
    torch.cuda.set_device(dev)
    ffi.load()
    # who is in the job: (rank, device, name, bus id) of every rank, gathered over the job's own backend.  One rank per DISTINCT device is a
    # precondition of the N > 1 line - fail loudly instead of reporting a "2-GPU" number from two ranks on one device (the shared-GPU
    # switch of the functional tests is the only exemption, and the line says so)
    seen = adist.ranks_seen(dev, world)
    shared_gpu = bool(os.environ.get("AFM_BENCH_SHARE_GPU"))
    assert shared_gpu or adist.distinct_devices(seen) == world, f"{world} ranks on {adist.distinct_devices(seen)} distinct devices: {seen}"

    K, W = args.steps, args.warmup
    model, diff_k, cfg = build(dev, str(K))
    if args.streams is not None:
        model.loop_streams, model.loop_streams_auto = args.streams, False
    model.gemm_tile = args.gemm_tile
    if args.pair_launch is not None:
        model.pair_launch = bool(args.pair_launch)
    if args.attn_group is not None:
        model.attn_group_waves = args.attn_group
    model.fused_layernorm = bool(args.fused_ln)
    model.no_ln_fold = bool(args.no_ln_fold)
    from afm.base import create_gaussian_diffusion
    cfg.diffusion.timestep_respacing = str(max(W, 1))
    diff_w = create_gaussian_diffusion(cfg)
    # the schedule tables are built at construction in the reference (gaussian_diffusion.py:119-170, "init only"); ours are uploaded
    # lazily per device, so touch them here instead of inside the timed region
    diff_k.tables(dev); diff_w.tables(dev)

    # this rank's shard [i0, i0 + B) of the job, resident in HBM before any timing.  weak: every rank owns --batch samples (global
    # sample indices rank * batch ...); strong: the --batch samples of ONE job are sharded contiguously (afm.dist.shard_range)
    def shard(scaling):
        return adist.job_shard(scaling, args.batch, rank, world)

    def batch_kwargs(total, i0, cnt):
        full = dict(c_text_feat=synth.text_feature(total), c_pc_xyz=synth.scene_cloud(total, NPTS),
                    c_pc_contact=synth.contact_map(total, NPTS), x_mask=synth.frame_mask(total, L, all_valid=True))
        return {k: v[i0:i0 + cnt].contiguous().to(dev) for k, v in full.items()}

    i0, B, total = shard(args.scaling)
    assert B >= 1, f"--batch {args.batch} leaves rank {rank} of {world} without a sample"
    kw = batch_kwargs(total, i0, B)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.condition_tokens(**kw)                      # one-off: SceneMapEncoder (FPS, kNN, set abstraction, attention)
    torch.cuda.synchronize()
    setup_ms = 1e3 * (time.perf_counter() - t0)          # cold: includes module load / first-launch costs
    kw2 = dict(kw, c_pc_xyz=kw["c_pc_xyz"].flip(0).contiguous(), c_pc_contact=kw["c_pc_contact"].flip(0).contiguous())

    def steady_setup(reps):
        """Warm cost of the step-invariant conditions: alternating scene batches (every call a cache miss), median of `reps`."""
        ts = []
        for i in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model.condition_tokens(**(kw2 if i % 2 == 0 else kw))
            torch.cuda.synchronize()
            ts.append(1e3 * (time.perf_counter() - t0))
        model.condition_tokens(**kw)
        return statistics.median(ts)

    def run(diffusion, seed, kwargs=None, nb=None, index0=None, gather=True):
        kwargs, nb, index0 = kwargs or kw, B if nb is None else nb, i0 if index0 is None else index0
        x = diffusion.p_sample_loop(model, (nb, L, D), clip_denoised=False, model_kwargs=kwargs, seed=seed, sample_index0=index0)
        if world > 1 and gather:                       # the path's only collective: gather the shards at the end (RCCL over xGMI)
            adist.all_gather_rows(x, world)
        return x

    def timed(kwargs=None, nb=None, index0=None):
        """W untimed warm-up steps, then EXACTLY K steps between barrier + synchronize on both sides; max over ranks."""
        if W > 0:
            run(diff_w, 1, kwargs, nb, index0)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        run(diff_k, 2, kwargs, nb, index0)             # exactly K steps
        t_enq = time.perf_counter() - t0               # host time to enqueue the K steps (the loop never synchronises)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = tt.item()
        return dt, t_enq

    # ---- process-start transient (profiles/r04_first_loop_transient.md, tools/probe_first_ms.py): the FIRST native-loop call of a process is
    # followed, 50 - 150 ms after it began, by one ~70 ms stall of the device (at every batch size; it passes unseen when the GPU idles;
    # it never comes back).  With the driver's command the W = 5 warm-up steps are that first call and the 20 timed steps land exactly in
    # the window: 232 - 433 steps/s in three such runs against 465 for the same 20 steps a moment later.  So the one-time part of a
    # sampling service is done HERE, as part of the setup and outside the timed region like the condition tokens: one 2-step loop call
    # (workspace / stream allocation, first launches), then the GPU idles until the window has passed.  The W warm-up steps and the K timed
    # steps follow unchanged.  The window is spent on the OTHER setup measurement (the warm cost of the condition tokens, median of 16 scene
    # batches: ~0.2 s of GPU work) instead of idling: the warm-up steps then start on a busy, clocked-up device (an idle gap costs the first
    # 20 steps another 2-3 %, tools/probe_k20_gap.py).
    cfg.diffusion.timestep_respacing = "2"                 # (the shortest schedule gaussian_diffusion.py can build: posterior_variance[1])
    diff_prime = create_gaussian_diffusion(cfg)
    diff_prime.tables(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(diff_prime, 0, gather=False)
    torch.cuda.synchronize()
    prime_ms = 1e3 * (time.perf_counter() - t0)
    t0 = time.perf_counter()
    setup_ms_steady = steady_setup(16)
    if args.settle_s > 0:
        time.sleep(args.settle_s)
    preflight = {"priming_loop_steps": 2, "priming_loop_ms": round(prime_ms, 2), "setup_repeats_after_priming": 16,
                 "ms_between_priming_and_warmup": round(1e3 * (time.perf_counter() - t0), 1), "settle_sleep_s": args.settle_s,
                 "why": "one-time ~70 ms device stall 50-150 ms after a process's first native-loop call (profiles/r04_first_loop_transient.md)",
                 "unprimed_first_request": {"first_loop_call_ms": round(prime_ms, 2), "steps": 2, "one_time_stall_ms": 70,
                                            "note": "a process's FIRST loop call pays workspace / stream allocation and first launches (first_loop_call_ms for 2 steps) and is followed "
                                                    "by the one-time stall: a first 1000-step request of a fresh process takes about first_loop_call_ms + 70 ms longer than sample_latency's p50"}}
    if world > 1:
        dist.barrier()

    dt, t_enq = timed()

    # the path's only collective on its own (N > 1): all_gather of this rank's [B / G, L, D] f32 shard - already inside `dt`, priced here
    gather = None
    if world > 1:
        gather = adist.time_all_gather(torch.zeros(B, L, D, device=dev), world)
        gt = torch.tensor([gather["median_us"]], device=dev, dtype=torch.float64)
        if dist.get_backend() == "gloo":
            gt = gt.cpu()
        dist.all_reduce(gt, op=dist.ReduceOp.MAX)
        gather["max_over_ranks_median_us"] = round(float(gt.item()), 1)
        gather["fraction_of_timed_region"] = round(gather["max_over_ranks_median_us"] * 1e-6 / dt, 6)

    # the other scaling mode in the same invocation (N > 1 only; informational): the driver calls bench.py without --scaling, so a
    # weak run also reports the strong-scaling rate of ONE 32-sample job sharded over the ranks (B/N samples per GPU), and vice versa
    other = None
    if world > 1:
        o_mode = "strong" if args.scaling == "weak" else "weak"
        o_i0, o_B, o_total = shard(o_mode)
        o_kw = batch_kwargs(o_total, o_i0, o_B)
        o_dt, _ = timed(o_kw, o_B, o_i0)
        o_steps = (world if o_mode == "weak" else 1) * K
        other = {"scaling": o_mode, "value": round(o_steps / o_dt, 2), "unit": "steps/s", "ms_per_step": round(1e3 * o_dt / K, 4),
                 "batch_per_gpu": o_B, "job_samples": o_total,
                 "note": "steps/s of the whole job; strong = one 32-sample job (k_sample = 32) sharded over the GPUs, one all_gather at the end"}

    # roofline of the dominant kernel: the same K steps again with every launch bracketed by HIP events on its
    # stream.  Sub-batch streams are switched OFF for this pass so a launch's elapsed time is the kernel's own
    # (with 2 streams the other sub-batch's kernels share the GPU inside every bracket); same kernels, same
    # total work, launches of twice the rows.  `rocprofv3 ... bench.py --streams 1` reproduces these averages.
    roof = None
    if rank == 0:
        streams_timed = model.loop_streams
        model.loop_streams = 1
        # the multi-stream loop runs its wide GEMMs on 128 x 128 tiles when every sub-batch has >= 4096 rows (csrc/cmdm.hip, round 6): the
        # single-stream pass must time the SAME tile program, so the rule's outcome is forced here (tile shapes are bit-identical)
        n_sub = max(1, min(streams_timed, B // 8 if model.loop_streams_auto else B))
        tile_timed = model.gemm_tile
        if n_sub >= 2 and not tile_timed and (B // n_sub) * (2 + NPTS // 64 + L) >= 4096:
            model.gemm_tile = 5
        run(diff_w, 1, gather=False)                  # rank 0 only: no collective in this pass
        ffi.profile_enable(True)
        ffi.profile_read()
        run(diff_k, 2, gather=False)
        prof = ffi.profile_read()
        ffi.profile_enable(False)
        model.loop_streams, model.gemm_tile = streams_timed, tile_timed
        name = max(prof, key=lambda k: prof[k]["total_ms"]) if prof else None      # dominant kernel of the step
        g = prof.get(name)
        if g:
            ach = g["total_work"] / (g["total_ms"] * 1e-3) / 1e12
            roof = {"bound": "mfma", "kernel": name, "achieved": round(ach, 2), "peak": F32_MFMA_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(ach / F32_MFMA_PEAK_TFLOPS, 4), "traffic": pmc_traffic(name),
                    "avg_launch_us": round(1e3 * g["total_ms"] / g["launches"], 2), "launches": g["launches"],
                    "flops_per_launch": g["total_work"] / g["launches"],
                    "all_kernels_ms_per_step": {k: round(v["total_ms"] / K, 4) for k, v in prof.items()},
                    "all_kernels_tflops": {k: round(v["total_work"] / (v["total_ms"] * 1e-3) / 1e12, 1) for k, v in prof.items() if v["total_work"] > 0}}
            if "split_bf16" in name:
                # f32 results computed as 9 exact bf16 x bf16 products per f32 product on the bf16 matrix pipe: `achieved` / `peak` above
                # are ALGORITHMIC f32 FLOPs against the f32 MFMA peak (the dtype's peak); the pipe actually used is priced here
                from afm import ops as afm_ops
                nprod = afm_ops.get_gemm_split()[0]
                roof["matrix_pipe"] = {"instruction": "v_mfma_f32_32x32x16_bf16", "bf16_products_per_f32_product": nprod,
                                       "issued_tflops": round(ach * nprod, 1), "peak": BF16_MFMA_PEAK_TFLOPS,
                                       "frac": round(ach * nprod / BF16_MFMA_PEAK_TFLOPS, 4)}

    # p50 sequence-sample latency: full 1000-step p_sample_loop, wall time of the whole call incl. the final synchronise
    lat = None
    if rank == 0 and world == 1 and (args.latency_runs > 0 or args.latency_runs_b1 > 0):
        cfg.diffusion.timestep_respacing = ""
        diff_full = create_gaussian_diffusion(cfg)
        diff_full.tables(dev)
        lat = {"steps": 1000, "warmups": args.latency_warmups, "protocol": "SURVEY 8d: p50 of >= 20 full 1000-step p_sample_loop calls after 3 warm-up calls, wall time incl. the final synchronise"}
        for nb, runs in ((args.batch, args.latency_runs), (1, args.latency_runs_b1)):
            if runs <= 0:
                continue
            kwb = kw if nb == B else {k: v[:nb].contiguous() for k, v in kw.items()}
            run(diff_w, 3, kwb, nb, 0)                 # first call at this batch size (condition tokens, workspaces)
            for i in range(args.latency_warmups):
                run(diff_full, 3 + i, kwb, nb, 0)
            ts = []
            for i in range(runs):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                run(diff_full, 10 + i, kwb, nb, 0)
                torch.cuda.synchronize()
                ts.append(1e3 * (time.perf_counter() - t0))
            lat[f"B{nb}"] = {"p50_ms": round(statistics.median(ts), 1), "min_ms": round(min(ts), 1), "max_ms": round(max(ts), 1), "runs": runs}
        model.condition_tokens(**kw)

    # informational only (never `value`): the same K steps with afm_linear's other arithmetic settings.  Default (the timed run above, since
    # round 6): the SIX largest cross products of the bf16x3 split on every eligible GEMM (K >= 128, K % 16 == 0) and in the attention - the
    # worst-case table of tests/test_gpu_arith.py (profiles/r06_arith_worstcase.json) is the evidence; all nine products are an alternative here.
    alt = None
    if rank == 0 and world == 1 and not args.no_alt_gemm:
        from afm import ops as afm_ops
        alt = {}
        saved = afm_ops.get_gemm_split()
        try:
            for tag, (products, min_n) in (("native_f32_mfma_everywhere", (0, 0)), ("split9_all_gemms_and_attention", (9, 0)),
                                           ("split9_wide_gemms_only", (9, 1024)), ("split6_wide_gemms", (6, 1024)), ("split6_all_gemms", (6, 0))):
                afm_ops.set_gemm_split(products, min_n)
                run(diff_w, 1)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                run(diff_k, 2)
                torch.cuda.synchronize()
                alt[tag] = round(K / (time.perf_counter() - t0), 2)
            # plain bf16 x bf16 (ONE product, the two leading terms): NOT f32 arithmetic, fails the parity bar, never used by the library.
            # Reported so the price of the f32 requirement is a measured number: its rate (this kernel still pays the operand split,
            # so a real bf16 kernel would be faster still) and the drift of a full 1000-step chain against the default arithmetic
            # on the same Philox noise (B = 2; the default itself sits 4e-6 from the CPU oracle after 1000 steps).
            afm_ops.set_gemm_split(1, 0)
            run(diff_w, 1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(diff_k, 2)
            torch.cuda.synchronize()
            bf16_rate = round(K / (time.perf_counter() - t0), 2)
            cfg.diffusion.timestep_respacing = ""
            diff_1k = create_gaussian_diffusion(cfg)
            kw_d = {k: v[:2].contiguous() for k, v in kw.items()}
            chains = {}
            for products in (9, 6, 1):
                afm_ops.set_gemm_split(products, 0)
                snaps = {10: None, 100: None}
                snaps[1000] = diff_1k.p_sample_loop(model, (2, L, D), clip_denoised=False, model_kwargs=kw_d, seed=77, sample_index0=0,
                                                    snapshots=snaps)
                chains[products] = snaps
            # six products (the three smallest dropped: each <= 2^-24 |a||w|, i.e. of the size of f32's own rounding of a product) against all
            # nine: the drift of a full chain between the two, in the same terms as the parity tolerance
            alt["split6_vs_split9_max_abs_drift"] = {str(k): float(f"{(chains[6][k] - chains[9][k]).abs().max().item():.3e}")
                                                     for k in (10, 100, 1000)}
            alt["bf16_one_product_NOT_f32"] = {
                "steps_per_s": bf16_rate,
                "max_abs_drift_vs_split9": {str(k): float(f"{(chains[1][k] - chains[9][k]).abs().max().item():.3e}") for k in (10, 100, 1000)},
                "max_abs_x": round(chains[9][1000].abs().max().item(), 2), "parity_tolerance": 1e-3}
            model.condition_tokens(**kw)
        finally:
            afm_ops.set_gemm_split(*saved)
        alt["unit"] = "steps/s"

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.cpu_steps, args.cpu_reps)

    sub_streams = max(1, min(model.loop_streams, B // 8 if model.loop_streams_auto else B))       # of the timed headline run
    # ---- `secondary` (N = 1 only; after the headline, outside its timed region): everything else BASELINE.json names, driver-visible
    secondary = None
    if rank == 0 and world == 1 and not args.no_secondary:
        secondary = {}
        K2, W2 = 100, 10
        cfg.diffusion.timestep_respacing = str(K2)
        diff_k2 = create_gaussian_diffusion(cfg); diff_k2.tables(dev)

        def rate(kwargs, nb, diffusion=diff_k2, steps=K2):
            """steps/s of a `steps`-step loop call at this batch size: the MEDIAN of three calls after a warm-up call (the first call at a new
            batch size allocates its workspace; single calls of the same process differed by up to 18 % at B = 8)."""
            run(diff_w, 1, kwargs, nb, 0)
            rs = []
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                run(diffusion, 2, kwargs, nb, 0)
                torch.cuda.synchronize()
                rs.append(steps / (time.perf_counter() - t0))
            return statistics.median(rs)
        try:
            # what each GPU runs when the 32-sample job is sharded over G GPUs (strong scaling): B = 32 / G samples; per-sample efficiency
            # against B = 32 bounds the speed-up at G GPUs (no collective inside the loop; the final all_gather is 0.8 MB per rank)
            base = rate(kw, B)
            tab = {}
            for nb in (16, 8, 4, 1):
                kwb = {k: v[:nb].contiguous() for k, v in kw.items()}
                r = rate(kwb, nb)
                tab[f"B{nb}"] = {"steps_per_s": round(r, 1), "ms_per_step": round(1e3 / r, 4), "gpus_at_32_samples": 32 // nb,
                                 "per_sample_efficiency_vs_B32": round((r * nb) / (base * B), 4), "predicted_speedup_at_that_gpu_count": round(r / base, 2)}
            secondary["strong_scaling_batch_per_gpu"] = {"steps": K2, "statistic": "median of 3 loop calls", "B32_steps_per_s": round(base, 1), **tab}
            model.condition_tokens(**kw)
        except Exception as e:                          # noqa: BLE001
            secondary["strong_scaling_batch_per_gpu"] = {"error": f"{type(e).__name__}: {e}"}
        try:
            # the headline shape as the reference runs it: conditions (text adapter, SceneMapEncoder over 32 x 8192 points, contact adapter)
            # recomputed inside EVERY step (models/cmdm.py:134-156), p_sample by p_sample
            Kf = 10
            model.hoist_conditions = False
            tvec = diff_k.tables(dev).timesteps(B)

            def faithful():
                x = torch.zeros(B, L, D, device=dev)
                for j in range(Kf):
                    x = diff_k.p_sample(model, x, tvec[K - 1 - j], clip_denoised=False, model_kwargs=kw, seed=1, step=j)["sample"]
                return x
            faithful()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            faithful()
            torch.cuda.synchronize()
            dtf = (time.perf_counter() - t0) / Kf
            secondary["configs[1] faithful"] = {"steps_per_s": round(1 / dtf, 2), "ms_per_step": round(1e3 * dtf, 3), "steps": Kf,
                                                "note": "B=32, L=196, N=8192, conditions recomputed every step (hoist_conditions=False), one p_sample call per step"}
        except Exception as e:                          # noqa: BLE001
            secondary["configs[1] faithful"] = {"error": f"{type(e).__name__}: {e}"}
        finally:
            model.hoist_conditions = True
        try:
            import importlib.util
            spec = importlib.util.spec_from_file_location("afm_bench_configs", os.path.join(ROOT, "tools", "bench_configs.py"))
            bc = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(bc)
            del model                                    # the headline model's workspaces are not needed any more
            torch.cuda.empty_cache()
            secondary.update(bc.secondary_block(quick=True))
        except Exception as e:                          # noqa: BLE001
            secondary["error"] = f"{type(e).__name__}: {e}"

    if rank == 0:
        ms = 1e3 * dt / K
        job_steps = (world if args.scaling == "weak" else 1) * K
        flops = step_flops(args.batch * (world if args.scaling == "weak" else 1))
        elided = elided_flops(args.batch * (world if args.scaling == "weak" else 1))
        line = {
            "metric": f"denoising steps/sec (B={args.batch}, L=196, N=8192)", "value": round(job_steps / dt, 2), "unit": "steps/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(ms, 4), "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "CMDM trans_enc p_sample_loop, HumanML3D t2m_contact_motion config (BASELINE configs[1])",
                       "gemm_arithmetic": "f32 in / f32 out / f32 accumulate; every GEMM with K % 16 == 0 and the attention's two products: exact 3-way bf16 split of both "
                                          "operands, the SIX largest of the nine cross products (each exact in f32; dropped: a2 w3 + a3 w2 + a3 w3 <= 2^-23 |a w|) on the bf16 MFMA "
                                          "pipe, f32 accumulation - worst-case error never above the nine-product form's and in the class of an unfused f32 chain "
                                          "(tests/test_gpu_arith.py, profiles/r06_arith_worstcase.json); all nine products: alt_gemm_modes.split9_all_gemms_and_attention",
                       "gemm_products": __import__("afm.ops", fromlist=["ops"]).get_gemm_split()[0],
                       "batch_per_gpu": B, "job_samples": total, "frames": L, "motion_dim": D, "scene_points": NPTS, "tokens": 2 + NPTS // 64 + L,
                       "conditions": "hoisted (step-invariant, computed once: setup_ms)",
                       "parallelism": f"batch-shard x{world} ({args.scaling})", "sub_batch_streams": sub_streams},
            "algorithmic_tflops": round(flops * K / dt / 1e12, 2),
            "executed_tflops": round((flops - elided) * K / dt / 1e12, 2),
            "elided_gflop_per_step": round(elided / 1e9, 2),
            "elided_note": "bit-neutral eliminations inside the timed step: the last layer's out_proj / FFN run on the motion rows only, layer 0's "
                           "q|k|v rows of the step-invariant condition tokens are computed once per loop; algorithmic_tflops prices the full "
                           "SURVEY 8d step (257 GFLOP at B=32), executed_tflops what the kernels really did",
            "host_enqueue_ms_per_step": round(1e3 * t_enq / K, 4),
            "setup_ms": round(setup_ms, 2), "setup_ms_steady": round(setup_ms_steady, 2), "preflight": preflight,
            "ranks_seen": seen, "distinct_devices": adist.distinct_devices(seen), "shared_gpu_test_mode": shared_gpu,
            "batch_per_gpu_by_rank": [adist.job_shard(args.scaling, args.batch, r, world)[1] for r in range(world)],
            "final_all_gather": gather,
            "scaling_modes": ({args.scaling: round(job_steps / dt, 2), other["scaling"]: other["value"]} if other else {args.scaling: round(job_steps / dt, 2)}),
            "roofline": roof, "cpu_baseline": cpu, "sample_latency": lat, "alt_gemm_modes": alt, "other_scaling_mode": other,
            "secondary": secondary,
        }
        print(json.dumps(line))
    if world > 1:
        dist.barrier()                                 # rank 0 is still in its roofline pass: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
