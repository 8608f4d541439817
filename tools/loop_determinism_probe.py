"""Run-to-run determinism probe of the native CDM sampling loop at BASELINE configs[4]'s ADM size (32 samples x 8192 points, 50 steps):
R runs with the default two sub-batch streams against the single-stream result, bit for bit; prints the runs that differ, which
samples and by how much.  This is the harness of profiles/r02_decfold_nondeterminism.md (a kernel that only misbehaved with the second
stream active: ~1/4 of the runs differed).    python tools/loop_determinism_probe.py [R] [cdm|cmdm]
`cmdm`: the same for the headline loop (B = 32, L = 196, N = 8192, 100 respaced steps, two sub-batch streams against one)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "afford-motion_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from afm import synth                                         # noqa: E402
from afm.base import create_gaussian_diffusion, create_model  # noqa: E402
from gpu_util import load_named_weights                       # noqa: E402
from test_gpu_cdm import cdm_cfg                              # noqa: E402

K, N = 32, 8192
R = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda:0")
if len(sys.argv) > 2 and sys.argv[2] == "cmdm":
    from test_gpu_cmdm import cmdm_cfg                        # noqa: E402
    L = 196
    model = create_model(cmdm_cfg(num_points=N), device=dev); load_named_weights(model); model = model.to(dev).eval()
    diff = create_gaussian_diffusion(cmdm_cfg(steps=1000, respacing="100"))
    kwm = dict(c_text_feat=synth.text_feature(K).to(dev), c_pc_xyz=synth.scene_cloud(K, N, seed=3).to(dev),
               c_pc_contact=synth.contact_map(K, N).to(dev), x_mask=synth.frame_mask(K, L, seed=2).to(dev))

    def runm(streams):
        model.loop_streams, model.loop_streams_auto = streams, False
        return diff.p_sample_loop(model, (K, L, 263), clip_denoised=False, model_kwargs=kwm, seed=9).clone()

    refm, badm = runm(1), []
    for r in range(R):
        junk = torch.randn(32 << 20, device=dev) if r % 2 else None
        out = runm(2)
        del junk
        d = (out != refm).flatten(1).sum(1)
        if int(d.sum()):
            badm.append((r, [i for i, c in enumerate(d.tolist()) if c], f"{(out - refm).abs().max().item():.2e}"))
    print(f"cmdm: {len(badm)} bad of {R}:", badm[:6], flush=True)
    sys.exit(0)
adm = create_model(cdm_cfg(num_points=N), device=dev); load_named_weights(adm); adm = adm.to(dev).eval()
d_adm = create_gaussian_diffusion(cdm_cfg(steps=500, respacing="50"))
kw = dict(c_text_feat=synth.text_feature(1).repeat(K, 1).contiguous().to(dev), c_pc_xyz=synth.scene_cloud(1, N, seed=71).repeat(K, 1, 1).contiguous().to(dev))

def run(nsub):
    adm.loop_sub_batches = nsub
    return d_adm.p_sample_loop(adm, (K, N, 6), clip_denoised=False, model_kwargs=kw, seed=5).clone()

ref = run(1)
bad = []
for r in range(R):
    junk = torch.randn(64 << 20, device=dev) if r % 2 else None      # perturb timing / allocator state between runs
    out = run(0)
    del junk
    diff = (out != ref).flatten(1).sum(1)
    if int(diff.sum()):
        bad.append((r, [i for i, c in enumerate(diff.tolist()) if c], f"{(out - ref).abs().max().item():.2e}"))
print(f"cdm: {len(bad)} bad of {R}:", bad[:6], flush=True)
