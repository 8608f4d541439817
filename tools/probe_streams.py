"""Does splitting the batch over HIP streams fill the GEMM tails? (measurement tooling)"""
import sys, time, threading, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/afford-motion_amd')
from afm import synth
from afm.base import create_model_and_diffusion
from afm.config import load_config
dev = torch.device('cuda:0')
steps = 100
cfg = load_config("text_to_motion_contact_motion_gen", "cmdm", ["model.data_repr=h3d", "model.input_feats=263", f"diffusion.timestep_respacing='{steps}'"])
B, L = 32, 196
def make_model():
    m, d = create_model_and_diffusion(cfg, device=dev)
    synth.fill_module_(m); return m.to(dev).eval(), d
for nsub in (1, 2, 4):
    models = [make_model() for _ in range(nsub)]     # one engine (workspace + caches) per sub-batch
    b = B // nsub
    kws = [dict(c_text_feat=synth.text_feature(b).to(dev), c_cont_emb=synth.gaussian("c", (b, 128, 256)).to(dev),
                x_mask=synth.frame_mask(b, L, all_valid=True).to(dev)) for _ in range(nsub)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(nsub)]
    def work(i):
        with torch.cuda.stream(streams[i]):
            m, d = models[i]
            d.p_sample_loop(m, (b, L, 263), clip_denoised=False, model_kwargs=kws[i], seed=1, sample_index0=i * b)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(i,)) for i in range(nsub)]
        [t.start() for t in th]; [t.join() for t in th]
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{nsub} stream(s) x {b} samples: {1e3*dt/steps:.3f} ms/step -> {steps/dt:.1f} steps/s")
