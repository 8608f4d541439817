#!/usr/bin/env python
"""Secondary measurements for the BASELINE.json configs that bench.py's headline line does not cover
(configs[0] the reference's CPU case B=4 / L=60 / 100 steps, faithful and hoisted, CPU oracle next to the HIP path; configs[2] CDM Perceiver,
configs[3] set abstraction, configs[4] two-stage ADM -> AMDM at k_sample = 32),
one JSON object per line.  Single GPU; the per-stage timings come from the library's HIP-event profiler.

    python tools/bench_configs.py [--quick] > profiles/rNN_configs.jsonl
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

from afm import ffi, pointops, synth  # noqa: E402
from afm import scene as S  # noqa: E402
from afm.base import create_gaussian_diffusion, create_model  # noqa: E402
from afm.config import load_config  # noqa: E402
from afm.pipeline import two_stage_sample  # noqa: E402

dev = torch.device("cuda:0")
B, N, L = 32, 8192, 196


def pmc_traffic(which, prefix):
    """HBM bytes per launch of the kernel whose rocprof name starts with `prefix`, from the committed PMC passes of this round's
    `tools/collect_profiles.sh <round> cdm | points` (profiles/traffic_<which>.json: FETCH_SIZE x 2 + WRITE_SIZE, separate passes, corrected
    as MI355X_MICROARCH.md prescribes); None when the file or the kernel is absent (counters cannot be read from inside the process)."""
    try:
        with open(os.path.join(ROOT, "profiles", f"traffic_{which}.json")) as f:
            tr = json.load(f)
    except (OSError, ValueError):
        return None
    cands = [(v.get("launches", 0), k, v) for k, v in tr.get("kernels", {}).items() if k.replace(" ", "").startswith(prefix)]
    if not cands:
        return None
    _, name, v = max(cands, key=lambda c: c[0])
    return dict(v, kernel=name, source=tr.get("source"))


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def cdm_models(steps_adm="", steps_amdm=""):
    ca = load_config("text_to_motion_contact_gen", "cdm", ["model.arch=Perceiver", "model.scene_model.use_scene_model=False",
                                                           "model.input_feats=6", "model.text_model.max_length=20", "diffusion.steps=500",
                                                           f"diffusion.timestep_respacing='{steps_adm}'"])
    cm = load_config("text_to_motion_contact_motion_gen", "cmdm", ["model.data_repr=h3d", "model.input_feats=263", "diffusion.steps=1000",
                                                                   f"diffusion.timestep_respacing='{steps_amdm}'"])
    adm, amdm = create_model(ca, device=dev), create_model(cm, device=dev)
    synth.fill_module_(adm); synth.fill_module_(amdm)
    return adm.to(dev).eval(), create_gaussian_diffusion(ca), amdm.to(dev).eval(), create_gaussian_diffusion(cm)


def config0(quick, cpu=True):
    """BASELINE configs[0]: CMDM trans_enc, t2m_contact_motion settings, synthetic B=4, L=60, D=263, N=8192, 100 DDPM steps - the reference's CPU
    path next to the HIP path, both variants: *faithful* (the contact encoder re-run every step, as models/cmdm.py:149-156
    does) and *hoisted* (step-invariant conditions once).  CPU = the oracle restatement (torch-CPU, pinned to the reference by tests/golden) on this
    box's host cores; its faithful variant is timed on a bounded sample of steps (the SceneMapEncoder over 4 x 8192 points takes seconds per step)."""
    import statistics
    from oracle import denoiser_ref as dr, diffusion_ref as df, shapes as sh
    Bc, Lc, steps = 4, 60, 100
    cm = load_config("text_to_motion_contact_motion_gen", "cmdm", ["model.data_repr=h3d", "model.input_feats=263", "diffusion.steps=100"])
    amdm = create_model(cm, device=dev)
    synth.fill_module_(amdm)
    amdm = amdm.to(dev).eval()
    diff = create_gaussian_diffusion(cm)
    text, xyz, contact = synth.text_feature(Bc), synth.scene_cloud(Bc, N), synth.contact_map(Bc, N)
    mask = synth.frame_mask(Bc, Lc, seed=9)
    kw = dict(c_text_feat=text.to(dev), c_pc_xyz=xyz.to(dev), c_pc_contact=contact.to(dev), x_mask=mask.to(dev))
    run = lambda: diff.p_sample_loop(amdm, (Bc, Lc, 263), clip_denoised=False, model_kwargs=kw, seed=1)
    hip_hoisted = timed(run, 3) / steps                                   # native loop: conditions computed once per run (cache warm: 0 times)
    amdm.hoist_conditions = False

    def run_faithful():                                                    # per-step path: CMDM.forward re-encodes the scene every step
        x = torch.randn(Bc, Lc, 263, device=dev)
        tvec = diff.tables(dev).timesteps(Bc)
        for i in range(steps - 1, -1, -1):
            x = diff.p_sample(amdm, x, tvec[i], clip_denoised=False, model_kwargs=kw, seed=1, step=steps - 1 - i)["sample"]
        return x
    hip_faithful = timed(run_faithful, 1) / steps
    amdm.hoist_conditions = True
    flops = Bc * (5 * 190 * (4194304 + 2048 * 190) + 2 * (2 * 263 * 512 * Lc) + 2 * 2 * 512 * 512)
    hip = {"hoisted_steps_per_s": round(1 / hip_hoisted, 1), "hoisted_ms_per_step": round(1e3 * hip_hoisted, 4),
           "faithful_steps_per_s": round(1 / hip_faithful, 1), "faithful_ms_per_step": round(1e3 * hip_faithful, 4),
           "note": "hoisted = native sync-free loop; faithful = per-step CMDM.forward with hoist_conditions=False (SceneMapEncoder re-run every step)"}
    if not cpu:
        return {"config": "configs[0] CMDM trans_enc, t2m_contact_motion, B=4, L=60, D=263, N=8192, 100 DDPM steps", "hip": hip,
                "algorithmic_gflop_per_step_hoisted": round(flops / 1e9, 2)}
    # ---- CPU (oracle)
    sd = sh.weights(sh.cmdm())
    s100 = df.Schedule(100)
    avail = len(os.sched_getaffinity(0))
    torch.set_num_threads(min(avail, 16))                                  # 16 threads were the fastest on the headline shape (bench.py probes)
    x = synth.gaussian("c0b_x", (Bc, Lc, 263)); nz = synth.gaussian("c0b_nz", (Bc, Lc, 263)); t = torch.full((Bc,), 50)
    with torch.no_grad():
        t0 = time.perf_counter()
        cont = None
        from oracle import scene_ref as sr
        cont = sr.scene_map_encoder({k: v for k, v in sd.items() if k.startswith("contact_encoder.")}, "contact_encoder", xyz, contact, blocks=(2, 2, 2, 2))
        enc_s = time.perf_counter() - t0
        hoisted = lambda xx, tt, **k: dr.cmdm_forward(sd, xx, tt, text, x_mask=mask, cont_emb=cont)
        df.p_sample(s100, hoisted, x, t, nz)
        reps = []
        n_h = 20 if quick else steps
        for _ in range(1 if quick else 3):
            xx = x
            t0 = time.perf_counter()
            for i in range(n_h):
                xx = df.p_sample(s100, hoisted, xx, torch.full((Bc,), steps - 1 - i % steps), nz)["sample"]
            reps.append((time.perf_counter() - t0) / n_h)
        cpu_hoisted = statistics.median(reps)
        faithful = lambda xx, tt, **k: dr.cmdm_forward(sd, xx, tt, text, xyz, contact, mask)
        n_f = 1 if quick else 2
        t0 = time.perf_counter()
        xx = x
        for _ in range(n_f):
            xx = df.p_sample(s100, faithful, xx, t, nz)["sample"]
        cpu_faithful = (time.perf_counter() - t0) / n_f
    return {"config": "configs[0] CMDM trans_enc, t2m_contact_motion, B=4, L=60, D=263, N=8192, 100 DDPM steps",
            "cpu_oracle": {"hoisted_steps_per_s": round(1 / cpu_hoisted, 2), "hoisted_ms_per_step": round(1e3 * cpu_hoisted, 1),
                           "hoisted_sample": f"{len(reps)} x {n_h} chained p_sample steps", "hoisted_gflops": round(flops / cpu_hoisted / 1e9, 1),
                           "faithful_steps_per_s": round(1 / cpu_faithful, 3), "faithful_ms_per_step": round(1e3 * cpu_faithful, 1),
                           "faithful_sample": f"{n_f} steps (the contact encoder alone: {enc_s:.2f} s per call for 4 scenes)",
                           "threads": torch.get_num_threads(), "logical_cpus": avail},
            "hip": hip, "algorithmic_gflop_per_step_hoisted": round(flops / 1e9, 2)}


def config2(quick):
    """CDM Perceiver over N = 8192 points + text token, B = 32 (H3D variant: 9 input channels, 500-step schedule), and the HUMANISE variant
    (41 input channels: 32 scene features per point of the frozen backbone, hoisted out of the loop - SURVEY 8d[2] secondary)."""
    steps = 100            # (also in the quick pass of bench.py's `secondary` block: a 20-step call is 3.5 ms of stepping behind ~0.4 ms of per-call setup, 5100 vs 5800 steps/s)
    adm, d_adm, _, _ = cdm_models(str(steps), "2")
    lines = []
    for tag, model, diff, extra in (("H3D variant (9 input channels)", adm, d_adm, {}), ("HUMANISE variant (41 input channels, scene features hoisted)", None, None, None)):
        if model is None:
            cfg = load_config("text_to_motion_contact_gen", "cdm", ["model.arch=Perceiver", "model.input_feats=6", "model.text_model.max_length=20", "diffusion.steps=500",
                                                                   f"diffusion.timestep_respacing='{steps}'", "model.scene_model.use_scene_model=True",
                                                                   "model.scene_model.use_openscene=True", "model.scene_model.point_feat_dim=32",
                                                                   "model.scene_model.pretrained_weight=''", "task.dataset.use_openscene=True"])
            model = create_model(cfg, device=dev); synth.fill_module_(model); model = model.to(dev).eval()
            diff = create_gaussian_diffusion(cfg)
            extra = dict(c_pc_feat=synth.gaussian("cfg2_feat", (B, N, 32)).to(dev))
        kw = dict(c_text_feat=synth.text_feature(B).to(dev), c_pc_xyz=synth.scene_cloud(B, N).to(dev), **extra)
        run = lambda: diff.p_sample_loop(model, (B, N, 6), clip_denoised=False, model_kwargs=kw, seed=1)
        dt = timed(run, 2 if quick else 3) / steps
        ffi.profile_enable(True); ffi.profile_read(); run(); prof = ffi.profile_read(); ffi.profile_enable(False)
        dom = max(prof, key=lambda k: prof[k]["total_ms"]) if prof else None
        roof = None
        if dom:
            # dec_point_kernel is bound by the vector ALU's ISSUE slots (per 16-point tile and wave ~9.5 k issue cycles at K = 12 -
            # 64 erf-GELUs per lane with two quarter-rate transcendentals each + 91 f32 MFMAs that share the VALU's issue port; ~15 k at
            # K = 44), so the roofline is issue cycles: achieved = tiles x cycles per tile / launch time against 256 CUs x 4 SIMDs x 2.4 GHz.
            us = 1e3 * prof[dom]["total_ms"] / max(prof[dom]["launches"], 1)
            # FLOP/s like every other line: the matrix products dec_point_kernel executes per 16-point tile, counted from its
            # tile loop (csrc/perceiver_points.hip) - v_mfma_f32_16x16x4_f32 (2048 FLOP each): scores NKS, query-variance form NT NKS, variance
            # form (1 + NT)(4 + NKS), row-dots 64, tail 4 + NKS = 91 at NKS = 3 (H3D) / 347 at NKS = 11 (HUMANISE); linear1 as 16 NSTEP f32-equivalent
            # 16x16x32 products (16384 FLOP each; issued as 6 bf16 MFMAs per product) - against the f32 MFMA peak, the dtype's peak
            nks, nt, nstep = (3, 1, 1) if "H3D" in tag else (11, 3, 2)
            f32_mfma = nks + nt * nks + (1 + nt) * (4 + nks) + 64 + 4 + nks
            flop_tile = f32_mfma * 2048.0 + 16 * nstep * 16384.0
            tiles = B * N / 16
            ach = tiles * flop_tile / (us * 1e-6) / 1e12
            cyc = 9500.0 if "H3D" in tag else 15000.0
            roof = {"bound": "mfma", "kernel": dom, "avg_launch_us": round(us, 2), "achieved": round(ach, 2), "peak": 157.3,
                    "unit": "TFLOP/s", "frac": round(ach / 157.3, 4), "flops_per_launch": tiles * flop_tile,
                    "traffic": pmc_traffic("cdm" if "H3D" in tag else "cdm_h", dom.split("<")[0]),
                    # per point: x_t (6 floats) + xyz (3) + the DDPM update's noise (6) read, x_{t-1} (6) written = 84 B (+ the per-sample tables);
                    # the 3 xyz floats sit in 36-byte feature rows, so a row-granular fetch moves 108 B per point
                    "algorithmic_bytes_per_launch": B * N * (6 + 3 + 6 + 6) * 4, "row_granular_bytes_per_launch": B * N * (6 + 9 + 6 + 6) * 4,
                    "counting": f"{f32_mfma} v_mfma_f32_16x16x4_f32 + {16 * nstep} f32-equivalent 16x16x32 products (x 6 bf16 MFMAs) per 16-point tile; the erf-GELU of 256 hidden "
                                "channels per point (VALU) is not counted as FLOPs",
                    "issue_model": {"note": "what actually bounds the kernel: VALU + MFMA issue slots", "cycles_per_tile": cyc,
                                    "frac_of_issue_peak": round(tiles * cyc / (us * 1e-6) / 1e9 / (256 * 4 * 2.4), 4)}}
        lines.append({"config": f"configs[2] CDM Perceiver, B=32, N=8192, text token, 1 MI355X: {tag}", "metric": "denoising steps/sec", "value": round(1 / dt, 2), "roofline": roof,
                      "ms_per_step": round(1e3 * dt, 4), "dtype": "f32", "as_written_tflops": round(313.4e9 / dt / 1e12, 1),
                      "formulation": ("row-less sampling form (csrc/perceiver_points.hip): a point is its K = 12 / 44 inputs [x_t | features | 1] and 16 decoder attention "
                                      "weights; the 256-wide rows of the reference (adapters, LayerNorms, attention output, the MLP's input and hidden row) are never "
                                      "generated - enc_point / lat_head / 11 toklin launches / lat_dectables / dec_point per step; executed per step ~3 GF on the f32 "
                                      "matrix pipe + 2.4 GF x 6 bf16 products (linear1 as a K = 28 / 60 product, six-product arithmetic); CDM.no_gen = round 2's folded rows, CDM.no_fold = "
                                      "layer by layer (115 GF per step)"),
                      "kernels_ms_per_step": {k: round(v["total_ms"] / steps, 4) for k, v in prof.items()}})
        del model
    return lines


def config3(quick):
    """Set abstraction TransitionDown(32 -> 64, k = 16) at N = 8192 -> 2048 (reference stride 4) and -> 1024 (BASELINE-literal)."""
    out = []
    p = synth.scene_cloud(B, N).reshape(B * N, 3).to(dev)
    x = synth.gaussian("sa_feat", (B * N, 32)).to(dev)
    for stride in (4, 8):
        td = S.TransitionDown(32, 64, stride=stride, nsample=16)
        synth.fill_module_(td)
        td = td.to(dev).eval()
        m = N // stride
        t_all = timed(lambda: td.run(p, x, B), 3 if quick else 10)
        ffi.profile_enable(True); ffi.profile_read(); td.run(p, x, B); prof = ffi.profile_read(); ffi.profile_enable(False)
        fused = prof.get("transition_down_kernel", {"total_ms": float("nan")})["total_ms"]
        alg_bytes = B * (N * 3 * 4 + N * 32 * 4 + m * 16 * 4 + m * 64 * 4)          # xyz + feat + idx + out
        out.append({"config": f"configs[3] set abstraction N=8192->{m}, B=32, in 32 -> out 64, k=16", "total_ms": round(1e3 * t_all, 3),
                    "fps_ms": round(prof["fps_kernel"]["total_ms"], 3), "fps_us_per_round": round(1e3 * prof["fps_kernel"]["total_ms"] / (m - 1), 3),
                    "knn_ms": round(prof["knn_kernel"]["total_ms"], 3), "fused_gather_mlp_max_ms": round(fused, 4),
                    "fused_algorithmic_GBps": round(alg_bytes / (fused * 1e-3) / 1e9, 1), "fused_tflops": round(2 * 35 * 64 * B * m * 16 / (fused * 1e-3) / 1e12, 2),
                    "bounds": "FPS latency-bound (dependent rounds); kNN VALU; fused stage L2-gather / f32 MFMA",
                    "fused_algorithmic_bytes": alg_bytes,
                    "traffic": {k: pmc_traffic("points", k) for k in ("fps_pruned_kernel", "knn_kernel", "transition_down_kernel")}
                               if stride == 4 else "see the 8192 -> 2048 entry (one PMC target runs both strides; the json's means are over both)"})
    return out


def config4(quick):
    """Two-stage ADM (500 steps) -> glue -> AMDM (1000 steps), one text + scene, k_sample = 32 flattened into the batch."""
    sa, sm = ("10", "20") if quick else ("", "")
    adm, d_adm, amdm, d_amdm = cdm_models(sa, sm)
    text = synth.text_feature(1).repeat(B, 1).contiguous().to(dev)
    xyz = synth.scene_cloud(1, N).repeat(B, 1, 1).contiguous().to(dev)
    run = lambda: two_stage_sample(adm, d_adm, amdm, d_amdm, text_feat=text, xyz=xyz, frames=L, sigma=0.8, seed=3)
    dt = timed(run, 1)
    return {"config": "configs[4] ADM(500) -> AMDM(1000), k_sample=32 in one batch, 1 MI355X", "adm_steps": d_adm.num_timesteps,
            "amdm_steps": d_amdm.num_timesteps, "seconds_per_32_samples": round(dt, 3), "samples_per_sec": round(B / dt, 2),
            "as_written_pflop": round((d_adm.num_timesteps * 313.4e9 + d_amdm.num_timesteps * 257e9) / 1e15, 3)}


def training_block(quick=True):
    """SURVEY 8 row f-3 made driver-visible: one optimisation step of the FULL AMDM (trunk + SceneMapEncoder over 8192 points,
    B = 32, L = 196, train mode: dropout on, batch-statistics BatchNorm) = training_losses forward + backward + AdamW (utils/training.py:140-155),
    measured by tools/bench_train.py's function: steps/s, the step's dominant kernel with its `roofline`, the step against the f32 MFMA peak."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("afm_bench_train", os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_train.py"))
    bt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bt)
    r = bt.measure_cmdm(32, 196, True, 6 if quick else 20, 2 if quick else 3, 0, dev)
    keep = ("config", "metric", "value", "ms_per_step", "samples_per_sec", "executed_tflops", "step_frac_of_f32_mfma_peak", "roofline", "trainable_params")
    out = {k: r[k] for k in keep if k in r}
    top = sorted(r["kernels_ms_per_step"].items(), key=lambda kv: -kv[1])[:8]
    out["top_kernels_ms_per_step"] = dict(top)
    out["gradients"] = "deterministic: every reduction in a fixed order, the scatter-adds of the point branch as segmented sums (tests/test_gpu_train.py::test_two_identical_training_steps_of_the_scene_branch_are_bit_identical)"
    return out


def secondary_block(quick=True):
    """The `secondary` object of bench.py's JSON line (N = 1 only, after the headline, outside its timed region): every BASELINE config the
    headline does not cover, measured by the same driver-run process.  `quick`: bounded repetitions (the whole block stays within ~1-2 min;
    configs[0]'s CPU oracle: 20 chained hoisted steps and ONE faithful step - the contact encoder over 4 x 8192 points is seconds per call).  A failing part records its
    error instead of taking the headline line down with it."""
    ffi.load()
    out = {"note": "N = 1 only; measured after the headline by the same process, outside its timed region; tools/bench_configs.py functions"}
    for key, fn in (("configs[4]", lambda: config4(False)), ("configs[2]", lambda: config2(quick)), ("configs[3]", lambda: config3(quick)),
                    ("training", lambda: training_block(quick)),
                    ("configs[0]", lambda: config0(quick, cpu=True))):      # BASELINE configs[0] IS the reference's CPU path: the oracle on the host cores, bounded (20 hoisted steps, one faithful step, ~10 s)
        try:
            out[key] = fn()
        except Exception as e:                       # noqa: BLE001 - reported, never silenced
            out[key] = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--only", default="", help="config2 | config3 | config0 | config4: run one configuration")
    a = ap.parse_args()
    ffi.load()
    if a.only:
        r = globals()[a.only](a.quick)
        for line in (r if isinstance(r, list) else [r]):
            print(json.dumps(line), flush=True)
        return
    results = []
    for fn in (config4, config2, config3, config0):        # the long two-stage run first, on a fresh allocator state
        r = fn(a.quick)
        results.append(r if isinstance(r, list) else [r])
        torch.cuda.empty_cache()
    for r in (results[3], results[1], results[2], results[0]):
        for line in r:
            print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
