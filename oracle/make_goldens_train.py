"""Generate tests/golden/cmdm_training_grads.npz by running the REAL reference's
training_losses + backward (utils/training.py:140-152) on the reduced CMDM of make_goldens.py.

Run in the build container only:   python -m oracle.make_goldens_train  [--scene | --cdm | --masks | --mlp | --pointtrans | --pointtrans_train | --trans_dec | --trans_dec_train]
eval() mode (dropout off, BatchNorm on running statistics) so the result is a deterministic function of the inputs;
gradients of the denoiser trunk, the adapters and the TimestepEmbedder are stored (small tensors in full, large ones as a
strided sample + sum / abs-sum).
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.append(os.path.join(ROOT, "afford-motion_amd"))

from oracle._refimport import import_reference, to_attr  # noqa: E402
from oracle.make_goldens import GOLD, TEXTS, cmdm_cfg, diffusion_cfg, save  # noqa: E402

SAMPLE = 1024


def grad_digest(g: torch.Tensor):
    flat = g.detach().reshape(-1)
    if flat.numel() <= 2048:
        sample = flat
    else:
        sample = flat[:: flat.numel() // SAMPLE][:SAMPLE]
    return sample, torch.stack([flat.double().sum(), flat.double().abs().sum()])


def main():
    from afm import synth
    base, _ = import_reference()
    torch.manual_seed(0)
    g = np.load(os.path.join(GOLD, "cmdm_forward_N1024_L16.npz"))
    xyz, con, x_mask = torch.from_numpy(g["xyz"]), torch.from_numpy(g["contact"]), torch.from_numpy(g["x_mask"])
    B, L, N = 2, 16, xyz.shape[1]
    cfg = to_attr(dict(model=cmdm_cfg(num_points=N), diffusion=diffusion_cfg(1000, "")))
    model, diff = base.create_model_and_diffusion(cfg, device="cpu")
    synth.fill_module_(model)
    model.eval()
    kw = dict(c_text=TEXTS, c_pc_xyz=xyz, c_pc_contact=con, x_mask=x_mask)
    x0 = synth.gaussian("train_x0", (B, L, 263))
    tn = synth.gaussian("train_noise", (B, L, 263))
    tt = torch.tensor([17, 803])
    model.zero_grad()
    terms = diff.training_losses(model, x0, tt, model_kwargs=kw, noise=tn)
    terms["loss"].mean().backward()
    out = {"t": tt, "loss": terms["loss"].detach()}
    n = 0
    for name, p in model.named_parameters():
        if name.startswith(("contact_encoder.", "text_model.")) or p.grad is None:
            continue
        sample, sums = grad_digest(p.grad)
        out["g/" + name], out["s/" + name] = sample, sums
        n += 1
    save("cmdm_training_grads", **out)
    print(f"{n} parameter gradients")


def scene_main():
    """tests/golden/scene_encoder_train_N1024.npz: the reference SceneMapEncoder in train() mode (BatchNorm on batch
    statistics) - forward output, gradients of every parameter for the loss sum(out * dy), and the updated running
    statistics of two BatchNorm layers."""
    from afm import synth
    import_reference()
    import models.modules as rmod
    g = np.load(os.path.join(GOLD, "scene_map_encoder_N1024.npz"))
    xyz, con = torch.from_numpy(g["xyz"]), torch.from_numpy(g["contact"])
    sme = rmod.SceneMapEncoder(point_feat_dim=6, planes=[32, 64, 128, 256], blocks=[2, 2, 2, 2], num_points=xyz.shape[1])
    synth.fill_module_(sme)
    sme.train()
    out = sme(xyz, con)
    dy = synth.gaussian("scene_train_dy", tuple(out.shape))
    (out * dy).sum().backward()
    res = {"out": out.detach()}
    n = 0
    for name, p in sme.named_parameters():
        sample, sums = grad_digest(p.grad)
        res["g/" + name], res["s/" + name] = sample, sums
        n += 1
    for bn in ("enc1.0.bn", "enc2.1.transformer2.linear_w.0", "enc4.1.bn3"):
        mod = dict(sme.named_modules())[bn]
        res["rm/" + bn], res["rv/" + bn] = mod.running_mean.detach(), mod.running_var.detach()
    save("scene_encoder_train_N1024", **res)
    print(f"{n} scene-encoder parameter gradients")


def cdm_main():
    """tests/golden/cdm_training_grads.npz: the reference CDM (Perceiver, N=256) training_losses + backward, eval mode."""
    from afm import synth
    from oracle.make_goldens import cdm_cfg
    base, _ = import_reference()
    torch.manual_seed(0)
    B, Nc = 2, 256
    ccfg = to_attr(dict(model=cdm_cfg(num_points=Nc), diffusion=diffusion_cfg(500, "")))
    cdm, cdiff = base.create_model_and_diffusion(ccfg, device="cpu")
    synth.fill_module_(cdm)
    cdm.eval()
    cxyz = synth.scene_cloud(B, Nc, seed=14)
    x0 = synth.gaussian("cdm_train_x0", (B, Nc, 6))
    tn = synth.gaussian("cdm_train_noise", (B, Nc, 6))
    tt = torch.tensor([33, 470])
    cdm.zero_grad()
    terms = cdiff.training_losses(cdm, x0, tt, model_kwargs=dict(c_text=TEXTS, c_pc_xyz=cxyz), noise=tn)
    terms["loss"].mean().backward()
    out = {"t": tt, "loss": terms["loss"].detach()}
    n = 0
    for name, p in cdm.named_parameters():
        if name.startswith("text_model.") or p.grad is None:
            continue
        sample, sums = grad_digest(p.grad)
        out["g/" + name], out["s/" + name] = sample, sums
        n += 1
    save("cdm_training_grads", **out)
    print(f"{n} CDM parameter gradients")


def masks_main():
    """tests/golden/cmdm_forward_cond_masks.npz: CMDM.forward with the training-time condition switches of
    datasets/transforms.py (c_text_mask / c_text_erase / c_pc_mask / c_pc_erase), eval mode."""
    from afm import synth
    base, _ = import_reference()
    g = np.load(os.path.join(GOLD, "cmdm_forward_N1024_L16.npz"))
    xyz, con, x_mask = torch.from_numpy(g["xyz"]), torch.from_numpy(g["contact"]), torch.from_numpy(g["x_mask"])
    cfg = to_attr(dict(model=cmdm_cfg(num_points=xyz.shape[1]), diffusion=diffusion_cfg(1000, "")))
    model = base.create_model(cfg, device="cpu")
    synth.fill_module_(model)
    model.eval()
    x, t = torch.from_numpy(g["x"]), torch.from_numpy(g["t"])
    kw = dict(c_text=TEXTS, c_pc_xyz=xyz, c_pc_contact=con, x_mask=x_mask)
    sw = dict(c_text_mask=torch.tensor([[True], [False]]), c_text_erase=torch.tensor([[False], [True]]),
              c_pc_mask=torch.tensor([[False], [True]]), c_pc_erase=torch.tensor([[True], [False]]))
    with torch.no_grad():
        out_all = model(x, t, **kw, **sw)
        out_tm = model(x, t, **kw, c_text_mask=sw["c_text_mask"])
        out_pe = model(x, t, **kw, c_pc_erase=sw["c_pc_erase"])
    save("cmdm_forward_cond_masks", out_all=out_all, out_text_mask=out_tm, out_pc_erase=out_pe, **{k: v for k, v in sw.items()})


def mlp_main():
    """tests/golden/cdm_mlp_N256.npz: the reference CDM with `arch: 'MLP'` (the config default) - forward output and the
    gradients of training_losses + backward (eval mode), N = 256 points, 32-d per-point features supplied."""
    from afm import synth
    from oracle.make_goldens import cdm_cfg
    base, _ = import_reference()
    B, Nc = 2, 256
    mc = cdm_cfg(num_points=Nc)
    mc.update(arch="MLP", arch_mlp=dict(last_dim=512, point_mlp_dims=[512, 512], point_mlp_widening_factor=1, point_mlp_bias=True))
    # the reference's ContactMLP only runs WITH per-point features (cdm.py:80 uses `num_points` that is unbound otherwise)
    mc["scene_model"].update(use_scene_model=True, use_openscene=True, point_feat_dim=32)
    cdm, cdiff = base.create_model_and_diffusion(to_attr(dict(model=mc, diffusion=diffusion_cfg(500, ""))), device="cpu")
    synth.fill_module_(cdm)
    cdm.eval()
    cxyz = synth.scene_cloud(B, Nc, seed=14)
    cx = synth.gaussian("cdm_x", (B, Nc, 6))
    tc = torch.tensor([499, 7])
    kw = dict(c_text=TEXTS, c_pc_xyz=cxyz, c_pc_feat=synth.gaussian("cdm_pc_feat", (B, Nc, 32)))
    with torch.no_grad():
        out = cdm(cx, tc, **kw)
    x0, tn = synth.gaussian("cdm_train_x0", (B, Nc, 6)), synth.gaussian("cdm_train_noise", (B, Nc, 6))
    tt = torch.tensor([33, 470])
    cdm.zero_grad()
    terms = cdiff.training_losses(cdm, x0, tt, model_kwargs=kw, noise=tn)
    terms["loss"].mean().backward()
    res = {"t": tc, "out": out, "t_train": tt, "loss": terms["loss"].detach()}
    for name, p in cdm.named_parameters():
        if name.startswith("text_model.") or p.grad is None:
            continue
        res["g/" + name], res["s/" + name] = grad_digest(p.grad)
    save("cdm_mlp_N256", **res)
    keys = sorted(k for k in cdm.state_dict().keys() if "text_model" not in k)
    with open(os.path.join(GOLD, "cdm_mlp_state_dict_keys.txt"), "w") as f:
        f.write("\n".join(f"{k} {tuple(cdm.state_dict()[k].shape)}" for k in keys) + "\n")


def pointtrans_main():
    """tests/golden/cdm_pointtrans{,v2}_N1024.npz: forward of the reference CDM with `arch: 'PointTrans'` / `'PointTransV2'`."""
    from afm import synth
    from oracle.make_goldens import cdm_cfg
    base, _ = import_reference()
    B, Nc = 2, 1024
    cxyz = synth.scene_cloud(B, Nc, seed=16)
    cx = synth.gaussian("cdm_pt_x", (B, Nc, 6))
    tc = torch.tensor([499, 7])
    for arch, tag in (("PointTrans", "cdm_pointtrans_N1024"), ("PointTransV2", "cdm_pointtransv2_N1024")):
        mc = cdm_cfg(num_points=Nc)
        mc.update(arch=arch, arch_pointtrans=dict(last_dim=64, num_points=Nc, blocks=[2, 2, 2, 2]))
        cdm = base.create_model(to_attr(dict(model=mc)), device="cpu")
        synth.fill_module_(cdm)
        cdm.eval()
        with torch.no_grad():
            out = cdm(cx, tc, c_text=TEXTS, c_pc_xyz=cxyz)
        sel = torch.arange(0, Nc, 4)
        save(tag, t=tc, rows=sel, out_rows=out[:, sel], out_sum=out.double().sum(), out_abs_sum=out.double().abs().sum())
        keys = sorted(k for k in cdm.state_dict().keys() if "text_model" not in k)
        with open(os.path.join(GOLD, tag.replace("_N1024", "") + "_state_dict_keys.txt"), "w") as f:
            f.write("\n".join(f"{k} {tuple(cdm.state_dict()[k].shape)}" for k in keys) + "\n")


def pointtrans_train_main():
    """tests/golden/cdm_pointtrans{,v2}_training_grads.npz: the reference CDM with `arch: 'PointTrans'` (train() mode: BatchNorm on batch
    statistics; the arch has no dropout) and `'PointTransV2'` (eval() mode: its bottleneck encoder layer has dropout 0.1, which torch
    draws from its own generator) - training_losses + backward (utils/training.py:140-152), N = 1024."""
    from afm import synth
    from oracle.make_goldens import cdm_cfg
    base, _ = import_reference()
    B, Nc = 2, 1024
    cxyz = synth.scene_cloud(B, Nc, seed=16)
    x0 = synth.gaussian("cdm_pt_train_x0", (B, Nc, 6))
    tn = synth.gaussian("cdm_pt_train_noise", (B, Nc, 6))
    tt = torch.tensor([41, 388])
    for arch, tag, train in (("PointTrans", "cdm_pointtrans_training_grads", True), ("PointTransV2", "cdm_pointtransv2_training_grads", False)):
        torch.manual_seed(0)
        mc = cdm_cfg(num_points=Nc)
        mc.update(arch=arch, arch_pointtrans=dict(last_dim=64, num_points=Nc, blocks=[2, 2, 2, 2]))
        cdm, cdiff = base.create_model_and_diffusion(to_attr(dict(model=mc, diffusion=diffusion_cfg(500, ""))), device="cpu")
        synth.fill_module_(cdm)
        cdm.train() if train else cdm.eval()
        cdm.zero_grad()
        terms = cdiff.training_losses(cdm, x0, tt, model_kwargs=dict(c_text=TEXTS, c_pc_xyz=cxyz), noise=tn)
        terms["loss"].mean().backward()
        out = {"t": tt, "loss": terms["loss"].detach()}
        n = 0
        for name, p in cdm.named_parameters():
            if name.startswith("text_model.") or p.grad is None:
                continue
            sample, sums = grad_digest(p.grad)
            out["g/" + name], out["s/" + name] = sample, sums
            n += 1
        if train:
            mods = dict(cdm.named_modules())
            for bn in ("contact_model.enc1.0.bn", "contact_model.dec2.0.linear2.1", "contact_model.ctx.1"):
                out["rm/" + bn], out["rv/" + bn] = mods[bn].running_mean.detach(), mods[bn].running_var.detach()
            # the same step in float64: ~40 batch-statistics BatchNorms in series (down to 32 rows per batch at the bottleneck) make some of
            # these gradients ill-conditioned in f32 - the reference's OWN f32 result differs from its f64 result by what is stored as d64/
            cdm64, cdiff64 = base.create_model_and_diffusion(to_attr(dict(model=mc, diffusion=diffusion_cfg(500, ""))), device="cpu")
            synth.fill_module_(cdm64)
            cdm64 = cdm64.double().train()
            cdm64.zero_grad()
            t64 = cdiff64.training_losses(cdm64, x0.double(), tt, model_kwargs=dict(c_text=TEXTS, c_pc_xyz=cxyz.double()), noise=tn.double())
            t64["loss"].mean().backward()
            p64 = dict(cdm64.named_parameters())
            for k in [k for k in out if k.startswith("g/")]:
                s64, _ = grad_digest(p64[k[2:]].grad)
                out["g64/" + k[2:]] = s64.float()
            out["loss64"] = t64["loss"].detach().float()
        save(tag, **out)
        print(f"{arch}: {n} parameter gradients, loss {terms['loss'].tolist()}")


def trans_dec_main():
    """tests/golden/cmdm_trans_dec_N1024_L16.npz: CMDM.forward of the reference with `arch: 'trans_dec'`."""
    from afm import synth
    base, _ = import_reference()
    g = np.load(os.path.join(GOLD, "cmdm_forward_N1024_L16.npz"))
    xyz, con, x_mask = torch.from_numpy(g["xyz"]), torch.from_numpy(g["contact"]), torch.from_numpy(g["x_mask"])
    mc = cmdm_cfg(num_points=xyz.shape[1])
    mc["arch"] = "trans_dec"
    model = base.create_model(to_attr(dict(model=mc)), device="cpu")
    synth.fill_module_(model)
    model.eval()
    with torch.no_grad():
        out = model(torch.from_numpy(g["x"]), torch.from_numpy(g["t"]), c_text=TEXTS, c_pc_xyz=xyz, c_pc_contact=con, x_mask=x_mask)
    save("cmdm_trans_dec_N1024_L16", out=out)
    keys = sorted(k for k in model.state_dict().keys() if "text_model" not in k)
    with open(os.path.join(GOLD, "cmdm_trans_dec_state_dict_keys.txt"), "w") as f:
        f.write("\n".join(f"{k} {tuple(model.state_dict()[k].shape)}" for k in keys) + "\n")


def trans_dec_train_main():
    """tests/golden/cmdm_trans_dec_training_grads.npz: the reference CMDM with `arch: 'trans_dec'` (multi-scale SceneMapEncoderDecoder memories,
    decoder layers with cross-attention) - training_losses + backward, eval() mode (dropout off, BatchNorm on running statistics), every
    parameter gradient incl. the scene encoder-decoder's."""
    from afm import synth
    base, _ = import_reference()
    torch.manual_seed(0)
    g = np.load(os.path.join(GOLD, "cmdm_forward_N1024_L16.npz"))
    xyz, con, x_mask = torch.from_numpy(g["xyz"]), torch.from_numpy(g["contact"]), torch.from_numpy(g["x_mask"])
    B, L, N = 2, 16, xyz.shape[1]
    mc = cmdm_cfg(num_points=N)
    mc["arch"] = "trans_dec"
    model, diff = base.create_model_and_diffusion(to_attr(dict(model=mc, diffusion=diffusion_cfg(1000, ""))), device="cpu")
    synth.fill_module_(model)
    model.eval()
    x0 = synth.gaussian("train_x0", (B, L, 263))
    tn = synth.gaussian("train_noise", (B, L, 263))
    tt = torch.tensor([17, 803])
    model.zero_grad()
    terms = diff.training_losses(model, x0, tt, model_kwargs=dict(c_text=TEXTS, c_pc_xyz=xyz, c_pc_contact=con, x_mask=x_mask), noise=tn)
    terms["loss"].mean().backward()
    out = {"t": tt, "loss": terms["loss"].detach()}
    n = 0
    for name, p in model.named_parameters():
        if name.startswith("text_model.") or p.grad is None:
            continue
        sample, sums = grad_digest(p.grad)
        out["g/" + name], out["s/" + name] = sample, sums
        n += 1
    save("cmdm_trans_dec_training_grads", **out)
    print(f"trans_dec: {n} parameter gradients, loss {terms['loss'].tolist()}")


if __name__ == "__main__":
    if "--trans_dec_train" in sys.argv:
        trans_dec_train_main()
    elif "--pointtrans_train" in sys.argv:
        pointtrans_train_main()
    elif "--trans_dec" in sys.argv:
        trans_dec_main()
    elif "--pointtrans" in sys.argv:
        pointtrans_main()
    elif "--mlp" in sys.argv:
        mlp_main()
    elif "--masks" in sys.argv:
        masks_main()
    elif "--cdm" in sys.argv:
        cdm_main()
    elif "--scene" in sys.argv:
        scene_main()
    else:
        main()
