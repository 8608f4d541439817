"""test.py's call pattern (`progress=True`, reference test.py:94-101) against the unsliced native loop and the per-step path.

    python tools/bench_progress.py [--steps 1000] [--batch 32]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--batch", type=int, default=32)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    from afm import synth
    model, diff, _ = bench.build(dev, "" if a.steps == 1000 else str(a.steps))
    B, L, D = a.batch, bench.L, 263
    kw = dict(c_text_feat=synth.text_feature(B).to(dev), c_pc_xyz=synth.scene_cloud(B, bench.NPTS).to(dev),
              c_pc_contact=synth.contact_map(B, bench.NPTS).to(dev), x_mask=synth.frame_mask(B, L, all_valid=True).to(dev),
              info_index=list(range(B)), c_text=["walk to the sofa"] * B)
    model.condition_tokens(**kw)
    res = {}
    for name, call in (
            ("native", lambda: diff.p_sample_loop(model, (B, L, D), clip_denoised=False, noise=None, model_kwargs=kw, seed=1)),
            ("native_progress", lambda: diff.p_sample_loop(model, (B, L, D), clip_denoised=False, noise=None, model_kwargs=kw,
                                                           seed=1, progress=True)),
            ("per_step", lambda: [o for o in diff.p_sample_loop_progressive(model, (B, L, D), clip_denoised=False, noise=None,
                                                                           model_kwargs=kw, seed=1)][-1]["sample"])):
        call() if name != "per_step" else None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = call()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res[name] = {"seconds": round(dt, 4), "steps_per_s": round(a.steps / dt, 1), "abs_sum": float(out.double().abs().sum())}
    print(json.dumps({"batch": B, "steps": a.steps, **res}))


if __name__ == "__main__":
    main()
