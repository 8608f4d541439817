"""Lean target for rocprofv3 --pmc / --kernel-trace passes: 12 native steps at the bench shape, single stream, no profiler events.
    python tools/pmc_target.py          -> CMDM trans_enc loop (BASELINE configs[1]: B = 32, L = 196, T = 326)
    python tools/pmc_target.py cdm      -> CDM Perceiver loop  (BASELINE configs[2]: B = 32, N = 8192 points + text token)
    python tools/pmc_target.py cdm_h    -> the same loop, HUMANISE variant (41 input channels: 32 hoisted scene features per point)
    python tools/pmc_target.py points   -> BASELINE configs[3]: set abstraction TransitionDown(32 -> 64, k = 16) at N = 8192 -> 2048 and -> 1024, B = 32
                                           (fps_pruned_kernel, knn_kernel<16>, transition_down_kernel), three repetitions each"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "afford-motion_amd"))
from afm import synth  # noqa: E402
from afm.base import create_model_and_diffusion  # noqa: E402
from afm.config import load_config  # noqa: E402

dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "cmdm"
B = 32
if which == "points":
    from afm import scene as S
    N = 8192
    p = synth.scene_cloud(B, N).reshape(B * N, 3).to(dev)
    x = synth.gaussian("sa_feat", (B * N, 32)).to(dev)
    for stride in (4, 8):
        td = S.TransitionDown(32, 64, stride=stride, nsample=16)
        synth.fill_module_(td)
        td = td.to(dev).eval()
        for _ in range(3):
            td.run(p, x, B)
elif which in ("cdm", "cdm_h"):
    N = 8192
    scene = (["model.scene_model.use_scene_model=True", "model.scene_model.use_openscene=True", "model.scene_model.point_feat_dim=32",
              "model.scene_model.pretrained_weight=''", "task.dataset.use_openscene=True"] if which == "cdm_h" else ["model.scene_model.use_scene_model=False"])
    cfg = load_config("text_to_motion_contact_gen", "cdm", ["model.arch=Perceiver", "model.input_feats=6", "model.text_model.max_length=20",
                                                            "diffusion.steps=500", "diffusion.timestep_respacing='12'"] + scene)
    model, diff = create_model_and_diffusion(cfg, device=dev)
    synth.fill_module_(model)
    model = model.to(dev).eval()
    # one sub-batch stream: per-kernel durations and counters of whole-batch launches, not of two half-batch launches overlapping each
    # other (the product default from B = 16 on is two streams; steps/s is quoted on that)
    model.loop_sub_batches = int(os.environ.get("AFM_PROFILE_SUBBATCH", "1"))
    kw = dict(c_text_feat=synth.text_feature(B).to(dev), c_pc_xyz=synth.scene_cloud(B, N).to(dev))
    if which == "cdm_h":
        kw["c_pc_feat"] = synth.gaussian("pmc_feat", (B, N, 32)).to(dev)
    for _ in range(2):
        diff.p_sample_loop(model, (B, N, 6), clip_denoised=False, model_kwargs=kw, seed=1)
else:
    L = 196
    cfg = load_config("text_to_motion_contact_motion_gen", "cmdm", ["model.data_repr=h3d", "model.input_feats=263", "diffusion.timestep_respacing='12'"])
    model, diff = create_model_and_diffusion(cfg, device=dev)
    synth.fill_module_(model)
    model = model.to(dev).eval()
    # AFM_PROFILE_STREAMS / AFM_PROFILE_PAIR / AFM_PROFILE_TILE (round 6, tools/pair_timeline.py): the two-stream loop, with or without the paired launch
    model.loop_streams = int(os.environ.get("AFM_PROFILE_STREAMS", "1"))
    model.pair_launch = os.environ.get("AFM_PROFILE_PAIR", "0") == "1"
    model.gemm_tile = int(os.environ.get("AFM_PROFILE_TILE", "0"))
    kw = dict(c_text_feat=synth.text_feature(B).to(dev), c_cont_emb=synth.gaussian("c", (B, 128, 256)).to(dev), x_mask=synth.frame_mask(B, L, all_valid=True).to(dev))
    diff.p_sample_loop(model, (B, L, 263), clip_denoised=False, model_kwargs=kw, seed=1)
torch.cuda.synchronize()
