import sys, time, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/afford-motion_amd')
from afm import synth, ffi
from afm.base import create_model_and_diffusion
from afm.config import load_config
dev = torch.device('cuda:0')
cfg = load_config("text_to_motion_contact_motion_gen", "cmdm", ["model.data_repr=h3d", "model.input_feats=263"])
model, diff = create_model_and_diffusion(cfg, device=dev)
synth.fill_module_(model); model = model.to(dev).eval()
B, L = 32, 196
kw = dict(c_text_feat=synth.text_feature(B).to(dev), c_cont_emb=synth.gaussian("c", (B,128,256)).to(dev), x_mask=synth.frame_mask(B, L, all_valid=True).to(dev))
x = synth.gaussian("x", (B, L, 263)).to(dev); t = torch.full((B,), 500, device=dev)
for _ in range(3): model(x, t, **kw)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(20): model(x, t, **kw)
torch.cuda.synchronize(); dt = (time.time() - t0) / 20
print(f"forward B=32 L=196: {dt*1e3:.3f} ms  -> {257.0/dt/1e3:.1f} TFLOP/s algorithmic")
from afm.base import create_gaussian_diffusion
cfg.diffusion.timestep_respacing = '50'
d50 = create_gaussian_diffusion(cfg)
d50.p_sample_loop(model, (B, L, 263), clip_denoised=False, model_kwargs=kw, seed=1)
torch.cuda.synchronize(); t0 = time.time()
d50.p_sample_loop(model, (B, L, 263), clip_denoised=False, model_kwargs=kw, seed=1)
torch.cuda.synchronize(); dt = (time.time() - t0) / 50
print(f"native loop: {dt*1e3:.3f} ms/step -> {1/dt:.1f} steps/s")
