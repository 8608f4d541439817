import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "afford-motion_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from afm import synth
from afm.base import create_model_and_diffusion
from gpu_util import load_named_weights
from test_gpu_cmdm import cmdm_cfg
dev = torch.device("cuda:0")
L, B = 196, 32
sync = len(sys.argv) > 1 and sys.argv[1] == "sync"
for trial in range(3):
    cfg = cmdm_cfg(num_points=8192, steps=1000, respacing="6")
    model, diff = create_model_and_diffusion(cfg, device=dev)
    load_named_weights(model); model = model.to(dev).eval()
    kw = dict(c_text_feat=synth.text_feature(B).to(dev), c_cont_emb=synth.gaussian("lf_cont", (B, 128, 256)).to(dev), x_mask=synth.frame_mask(B, L, seed=6).to(dev))
    out = []
    class _S:                       # both "side streams" are the main stream: sub-batches run one after the other
        cuda_stream = torch.cuda.current_stream(dev).cuda_stream
    for tag, fold, streams in (("F1", False, 1), ("T1", True, 1), ("T2", True, 2), ("T2", True, 2), ("T2seq", True, 2), ("T2seq", True, 2), ("T2", True, 2)):
        model.no_ln_fold = not fold
        model.loop_streams, model.loop_streams_auto = streams, False
        model._side_streams = [_S(), _S()] if tag == "T2seq" else [x for x in model._side_streams if not isinstance(x, _S)]
        if sync: torch.cuda.synchronize()
        r = diff.p_sample_loop(model, (B, L, 263), clip_denoised=False, model_kwargs=kw, seed=44).clone()
        if sync: torch.cuda.synchronize()
        out.append((tag, r))
    ref = out[0][1]
    print(f"trial {trial}:", [(t, f"{(r - ref).abs().max().item():.1e}", [i for i, v in enumerate(((r - ref).flatten(1).abs().max(1).values > 1e-3).tolist()) if v][:3] + ["..."] + [i for i, v in enumerate(((r - ref).flatten(1).abs().max(1).values > 1e-3).tolist()) if v][-2:]) for t, r in out], flush=True)
