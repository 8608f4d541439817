"""Worker of tests/test_gpu_dist.py: one rank of a `python -m torch.distributed.run` launch that runs the REAL sharded sampling path
(afm.dist.sharded_sample over the native CMDM and CDM loops, keyed by the global sample index) and lets rank 0 save the gathered
results.  AFM_TEST_SHARE_GPU=1: every rank uses cuda:0 (a 1-GPU box; backend gloo); otherwise rank r uses cuda:r (backend nccl = RCCL).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P tests/helpers/dist_gpu_worker.py OUT.pt TOTAL
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "afford-motion_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def build_models(dev):
    from afm import synth
    from afm.base import create_model_and_diffusion
    from afm.config import load_config
    cm = load_config("text_to_motion_contact_motion_gen", "cmdm", ["model.data_repr=h3d", "model.input_feats=263", "diffusion.steps=1000",
                                                                  "diffusion.timestep_respacing='6'", "model.contact_model.num_points=1024"])
    ca = load_config("text_to_motion_contact_gen", "cdm", ["model.arch=Perceiver", "model.scene_model.use_scene_model=False", "model.input_feats=6",
                                                          "diffusion.steps=500", "diffusion.timestep_respacing='5'"])
    amdm, d_amdm = create_model_and_diffusion(cm, device=dev)
    adm, d_adm = create_model_and_diffusion(ca, device=dev)
    synth.fill_module_(amdm); synth.fill_module_(adm)
    return amdm.to(dev).eval(), d_amdm, adm.to(dev).eval(), d_adm


def job(dev, total, rank, world, L=24, N=1024):
    """-> (motion [total, L, 263], contact [total, N, 6]) gathered on every rank; identical for every world size."""
    from afm import dist as adist, synth
    amdm, d_amdm, adm, d_adm = build_models(dev)
    kw = dict(c_text_feat=synth.text_feature(total).to(dev), c_pc_xyz=synth.scene_cloud(total, N).to(dev),
              c_pc_contact=synth.contact_map(total, N).to(dev), x_mask=synth.frame_mask(total, L, min_len=8).to(dev))
    motion = adist.sharded_sample(
        lambda skw, cnt, i0: d_amdm.p_sample_loop(amdm, (cnt, L, 263), clip_denoised=False, model_kwargs=skw, seed=11, sample_index0=i0),
        total, kw, rank, world)
    ckw = dict(c_text_feat=kw["c_text_feat"], c_pc_xyz=kw["c_pc_xyz"])
    contact = adist.sharded_sample(
        lambda skw, cnt, i0: d_adm.p_sample_loop(adm, (cnt, N, 6), clip_denoised=False, model_kwargs=skw, seed=12, sample_index0=i0),
        total, ckw, rank, world)
    return motion, contact


def main():
    from afm import dist as adist, ffi
    out_path, total = sys.argv[1], int(sys.argv[2])
    rank, world, local = adist.init_process_group()
    dev = torch.device("cuda:0" if os.environ.get("AFM_TEST_SHARE_GPU") else f"cuda:{local}")
    torch.cuda.set_device(dev)
    ffi.load()
    motion, contact = job(dev, total, rank, world)
    torch.cuda.synchronize()
    if rank == 0:
        torch.save({"motion": motion.cpu(), "contact": contact.cpu(), "world": world, "backend": dist.get_backend() if world > 1 else "none"}, out_path)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
