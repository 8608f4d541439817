"""Interleaved A/B of afm_linear's arithmetic settings inside one process (same box, same thermal state): R rounds, each timing K native-loop
steps per setting in rotating order.  Prints every sample and the medians.

    python tools/ab_gemm_modes.py [--rounds 7] [--steps 200] [--streams 2]
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

SETTINGS = {"native": (0, 0), "split9_wide": (9, 1024), "split9_all": (9, 0), "split6_wide": (6, 1024), "split6_all": (6, 0)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--streams", type=int, default=2)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    from afm import ops, synth
    model, diff, _ = bench.build(dev, str(a.steps))
    model.loop_streams = a.streams
    B, L, D = bench.B_PER_GPU, bench.L, bench.D
    kw = dict(c_text_feat=synth.text_feature(B).to(dev), c_pc_xyz=synth.scene_cloud(B, bench.NPTS).to(dev),
              c_pc_contact=synth.contact_map(B, bench.NPTS).to(dev), x_mask=synth.frame_mask(B, L, all_valid=True).to(dev))
    model.condition_tokens(**kw)
    names = list(SETTINGS)
    samples = {n: [] for n in names}
    saved = ops.get_gemm_split()
    try:
        for r in range(a.rounds + 1):                      # round 0 = warm-up, discarded
            order = names[r % len(names):] + names[:r % len(names)]
            for n in order:
                ops.set_gemm_split(*SETTINGS[n])
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                diff.p_sample_loop(model, (B, L, D), clip_denoised=False, model_kwargs=kw, seed=r)
                torch.cuda.synchronize()
                if r:
                    samples[n].append(round(a.steps / (time.perf_counter() - t0), 1))
    finally:
        ops.set_gemm_split(*saved)
    print(json.dumps({"streams": a.streams, "steps": a.steps, "samples_steps_per_s": samples,
                      "median": {n: statistics.median(v) for n, v in samples.items()}}))


if __name__ == "__main__":
    main()
