"""`-m gpu`: a sampling job launches NO eager ATen arithmetic kernel and NO vendor BLAS kernel.

Every device kernel of (weight-pack build -> condition tokens -> native p_sample_loop) for the CMDM, the CDM and the two-stage
pipeline is captured with torch.profiler and its name checked: the product's own kernels (libafm_hip.so) plus copies / fills /
the allocator's memsets are allowed; `at::native::*` element-wise / reduction kernels, rocBLAS / Tensile GEMMs and MIOpen are not.
(The reference's `th.randn` for x_T is `afm_randn` here, so no ATen RNG kernel either.)"""
import re

import pytest
import torch

from afm import synth
from afm.base import create_model_and_diffusion
from afm.config import load_config
from afm.pipeline import two_stage_sample
from gpu_util import dev

pytestmark = pytest.mark.gpu

# kernels that move bytes but do no arithmetic
_MOVERS = re.compile(r"(FillFunctor|fill_kernel|copy|Copy|memcpy|Memcpy|memset|Memset|CatArray|index_elementwise|index_kernel|gather|scatter)")
_FORBIDDEN = re.compile(r"(at::native|rocblas|Cijk_|miopen|MIOpen|hipblas|at_cuda_detail|cub::|rocprim::)")


def _device_kernel_names(fn):
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        fn()
        torch.cuda.synchronize()
    names = {}
    for ev in prof.events():
        if getattr(ev, "device_type", None) == torch.autograd.DeviceType.CUDA or str(getattr(ev, "device_type", "")).endswith("CUDA"):
            names[ev.name] = names.get(ev.name, 0) + 1
    return names


def _check(names, what):
    if not names:
        pytest.skip("torch.profiler recorded no device kernels on this box (`tools/gpu_call.sh validate` keeps the rocprofv3 kernel CSVs instead)")
    ours = [n for n in names if ("afm" in n or "_kernel" in n or "gemm_f32" in n) and not _FORBIDDEN.search(n)]
    assert ours, f"{what}: no product kernel among {sorted(names)[:20]}"
    bad = {n: c for n, c in names.items() if _FORBIDDEN.search(n) and not _MOVERS.search(n)}
    assert not bad, f"{what}: eager ATen / vendor-library arithmetic kernels in a sampling job: {bad}"


def _cmdm(respacing="3"):
    cfg = load_config("text_to_motion_contact_motion_gen", "cmdm", ["model.data_repr=h3d", "model.input_feats=263", "diffusion.steps=1000",
                                                                   f"diffusion.timestep_respacing='{respacing}'",
                                                                   "model.contact_model.num_points=1024"])
    model, diff = create_model_and_diffusion(cfg, device=dev())
    synth.fill_module_(model)
    return model.to(dev()).eval(), diff


def _cdm(respacing="3"):
    cfg = load_config("text_to_motion_contact_gen", "cdm", ["model.arch=Perceiver", "model.scene_model.use_scene_model=False", "model.input_feats=6",
                                                           "diffusion.steps=500", f"diffusion.timestep_respacing='{respacing}'"])
    model, diff = create_model_and_diffusion(cfg, device=dev())
    synth.fill_module_(model)
    return model.to(dev()).eval(), diff


def test_cmdm_sampling_job_launches_no_eager_arithmetic():
    model, diff = _cmdm()
    B, L, N = 2, 24, 1024
    kw = dict(c_text_feat=synth.text_feature(B).to(dev()), c_pc_xyz=synth.scene_cloud(B, N).to(dev()),
              c_pc_contact=synth.contact_map(B, N).to(dev()), x_mask=synth.frame_mask(B, L, min_len=8).to(dev()))
    torch.cuda.synchronize()

    def job():          # NOT under torch.no_grad(): the caller of test.py does not disable autograd either
        model.condition_tokens(**kw)                                     # SceneMapEncoder + adapters (weight pack, BN / LN folds built here)
        diff.p_sample_loop(model, (B, L, 263), clip_denoised=False, model_kwargs=kw, seed=5)
        # the per-step entry point as well (p_sample wraps the denoiser call in no_grad, as gaussian_diffusion.py:524-533 does)
        diff.p_sample(model, torch.zeros(B, L, 263, device=dev()), torch.tensor([1, 2], device=dev()), clip_denoised=False, model_kwargs=kw, seed=5)
    _check(_device_kernel_names(job), "CMDM")


def test_cdm_and_two_stage_sampling_jobs_launch_no_eager_arithmetic():
    adm, d_adm = _cdm()
    amdm, d_amdm = _cmdm("2")
    B, N, L = 2, 1024, 16
    text, xyz = synth.text_feature(B).to(dev()), synth.scene_cloud(B, N).to(dev())
    torch.cuda.synchronize()

    def job():
        d_adm.p_sample_loop(adm, (B, N, 6), clip_denoised=False, model_kwargs=dict(c_text_feat=text, c_pc_xyz=xyz), seed=3)
        two_stage_sample(adm, d_adm, amdm, d_amdm, text_feat=text, xyz=xyz, frames=L, sigma=0.8, seed=3)
    _check(_device_kernel_names(job), "CDM + two-stage")
