"""Phase timeline of workgroup (0, 0) of the last 13 toklin launches of a CDM step (debug build, see chain note in profiles/r03_cdm_chain.md)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "afford-motion_amd")):
    sys.path.insert(0, p)
import torch
from afm import ffi, synth
from afm.base import create_gaussian_diffusion, create_model
from afm.config import load_config
dev = torch.device("cuda:0")
B, N = 32, 8192
cfg = load_config("text_to_motion_contact_gen", "cdm", ["model.arch=Perceiver", "model.scene_model.use_scene_model=False", "model.input_feats=6",
                                                        "model.text_model.max_length=20", "diffusion.steps=500", "diffusion.timestep_respacing='20'"])
m = create_model(cfg, device=dev); synth.fill_module_(m); m = m.to(dev).eval()
m.loop_sub_batches = 1
d = create_gaussian_diffusion(cfg)
kw = dict(c_text_feat=synth.text_feature(B).to(dev), c_pc_xyz=synth.scene_cloud(B, N).to(dev))
for _ in range(2):
    d.p_sample_loop(m, (B, N, 6), clip_denoised=False, model_kwargs=kw, seed=1)
torch.cuda.synchronize()
lib = ffi.load()
buf = (C.c_ulonglong * 128)()
assert lib.afm_debug_toklin_timeline(buf) == 0
rows = [[buf[s * 8 + i] for i in range(5)] for s in range(16)]
rows = [r for r in rows if r[4] > r[0] > 0]
rows.sort(key=lambda r: r[0])
print("toklin launch (workgroup 0): staging us | LayerNorm us | product us | epilogue us | gap to next launch us")
for i, r in enumerate(rows):
    ln = (r[2] - r[1]) * 0.01 if r[2] > r[1] else 0.0
    t2 = r[2] if r[2] > r[1] else r[1]
    gap = (rows[i + 1][0] - r[4]) * 0.01 if i + 1 < len(rows) else float("nan")
    print(f"  {(r[1] - r[0]) * 0.01:7.2f} {ln:7.2f} {(r[3] - t2) * 0.01:7.2f} {(r[4] - r[3]) * 0.01:7.2f} {gap:9.2f}")
