"""Run-to-run determinism probe of the native sampling loops: R runs with two sub-batch streams against the single-stream result, bit for
bit; reports the runs that differ, which samples and by how much.  This is the harness of profiles/r02_decfold_nondeterminism.md (a
kernel that only misbehaved with the second stream active: ~1/4 of the runs differed); since round 3 the `-m gpu` suite runs 50 loops of
each (tests/test_gpu_cdm.py::test_two_stream_loop_soak, tests/test_gpu_cmdm.py::test_two_stream_loop_soak).
    python tools/loop_determinism_probe.py [R] [cdm|cmdm]
`cdm`:  BASELINE configs[4]'s ADM size (32 samples x 8192 points, 50 respaced steps);
`cmdm`: the headline loop (B = 32, L = 196, N = 8192, 100 respaced steps)."""
import os
import sys

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(_HERE, "..", "afford-motion_amd"), os.path.join(_HERE, "..", "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

K, N = 32, 8192


def _diffs(out, ref, r):
    d = (out != ref).flatten(1).sum(1)
    if int(d.sum()):
        return [(r, [i for i, c in enumerate(d.tolist()) if c], f"{(out - ref).abs().max().item():.2e}")]
    return []


def probe_cmdm(R: int, dev=None):
    from afm import synth
    from afm.base import create_gaussian_diffusion, create_model
    from gpu_util import load_named_weights
    from test_gpu_cmdm import cmdm_cfg
    dev = dev or torch.device("cuda:0")
    L = 196
    model = create_model(cmdm_cfg(num_points=N), device=dev); load_named_weights(model); model = model.to(dev).eval()
    diff = create_gaussian_diffusion(cmdm_cfg(steps=1000, respacing="100"))
    kwm = dict(c_text_feat=synth.text_feature(K).to(dev), c_pc_xyz=synth.scene_cloud(K, N, seed=3).to(dev),
               c_pc_contact=synth.contact_map(K, N).to(dev), x_mask=synth.frame_mask(K, L, seed=2).to(dev))

    def run(streams):
        model.loop_streams, model.loop_streams_auto = streams, False
        return diff.p_sample_loop(model, (K, L, 263), clip_denoised=False, model_kwargs=kwm, seed=9).clone()

    ref, bad = run(1), []
    assert torch.isfinite(ref).all()
    for r in range(R):
        junk = torch.randn(32 << 20, device=dev) if r % 2 else None      # perturb timing / allocator state between runs
        out = run(2)
        del junk
        bad += _diffs(out, ref, r)
    return bad


def probe_cdm(R: int, dev=None):
    from afm import synth
    from afm.base import create_gaussian_diffusion, create_model
    from gpu_util import load_named_weights
    from test_gpu_cdm import cdm_cfg
    dev = dev or torch.device("cuda:0")
    adm = create_model(cdm_cfg(num_points=N), device=dev); load_named_weights(adm); adm = adm.to(dev).eval()
    d_adm = create_gaussian_diffusion(cdm_cfg(steps=500, respacing="50"))
    kw = dict(c_text_feat=synth.text_feature(1).repeat(K, 1).contiguous().to(dev), c_pc_xyz=synth.scene_cloud(1, N, seed=71).repeat(K, 1, 1).contiguous().to(dev))

    def run(nsub):
        adm.loop_sub_batches = nsub
        return d_adm.p_sample_loop(adm, (K, N, 6), clip_denoised=False, model_kwargs=kw, seed=5).clone()

    ref, bad = run(1), []
    assert torch.isfinite(ref).all()
    for r in range(R):
        junk = torch.randn(64 << 20, device=dev) if r % 2 else None
        out = run(2)
        del junk
        bad += _diffs(out, ref, r)
    return bad


if __name__ == "__main__":
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    which = sys.argv[2] if len(sys.argv) > 2 else "cdm"
    bad = probe_cmdm(R) if which == "cmdm" else probe_cdm(R)
    print(f"{which}: {len(bad)} bad of {R}:", bad[:6], flush=True)
