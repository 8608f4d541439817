"""CDM / ADM denoiser (`Perceiver`): drop-in for the reference's `models.cdm.CDM`
(reference models/cdm.py:411-513 with `ContactPerceiver` :88-188) - same registry name, constructor,
config keys, call signature and state-dict keys; forward on the HIP path (csrc/perceiver*.hip).

Only `arch='Perceiver'` is built (every shipped script selects it); the scene
backbone of the HUMANISE variant (`use_scene_model=True` without openscene features) is a "next" row, so
per-point scene features must be supplied as `c_pc_feat` (the `use_openscene` path of cdm.py:495-505) or be absent.
"""
from __future__ import annotations

import ctypes as C
from typing import List

import torch
import torch.nn as nn

from . import autograd as AG
from . import autograd_points as AP
from . import ffi, ops
from .base import Model
from ._cache import HeldKey
from .cmdm import TimestepEmbedder, _FlatParamsMixin, _param_version
from .text import TextEncoderMixin, lang_feat_dim_type


class _Wrap(nn.Module):
    """Key-compatible stand-in for the reference's `Residual(module)` wrapper (modules.py:222-231)."""

    def __init__(self, module: nn.Module):
        super().__init__()
        self.module = module


class _MHA(nn.Module):
    """Parameter container of MultiHeadAttention (modules.py:234-299)."""

    def __init__(self, q_in: int, kv_in: int):
        super().__init__()
        self.q_proj = nn.Linear(q_in, q_in)
        self.k_proj = nn.Linear(kv_in, q_in)
        self.v_proj = nn.Linear(kv_in, q_in)
        self.o_proj = nn.Linear(q_in, q_in)


class _CrossAttention(nn.Module):
    def __init__(self, q_in: int, kv_in: int):
        super().__init__()
        self.q_norm = nn.LayerNorm(q_in)
        self.kv_norm = nn.LayerNorm(kv_in)
        self.attention = _MHA(q_in, kv_in)


class _SelfAttention(nn.Module):
    def __init__(self, ch: int):
        super().__init__()
        self.norm = nn.LayerNorm(ch)
        self.attention = _MHA(ch, ch)


def _mlp(ch: int, widening: int) -> nn.Sequential:
    return nn.Sequential(nn.LayerNorm(ch), nn.Linear(ch, widening * ch), nn.GELU(), nn.Linear(widening * ch, ch))


class PointSceneMLP(nn.Module):
    """Parameter container with the reference's names (cdm.py:13-39): per-point MLP, then a second MLP over
    [point feature | mean feature of the sample]."""

    def __init__(self, in_dim: int, out_dim: int, widening_factor: int = 1, bias: bool = True) -> None:
        super().__init__()
        self.mlp_pre = nn.Sequential(nn.LayerNorm(in_dim), nn.Linear(in_dim, widening_factor * in_dim, bias=bias), nn.GELU(),
                                     nn.Linear(widening_factor * in_dim, out_dim, bias=bias))
        self.mlp_post = nn.Sequential(nn.LayerNorm(2 * out_dim), nn.Linear(2 * out_dim, 2 * out_dim, bias=bias), nn.GELU(),
                                      nn.Linear(2 * out_dim, out_dim, bias=bias))

    @staticmethod
    def _mlp(seq: nn.Sequential, x):
        h = AG.linear(AG.layer_norm(x, seq[0]), seq[1].weight, seq[1].bias, act=ffi.ACT_GELU)
        return AG.linear(h, seq[3].weight, seq[3].bias)

    def run(self, x):
        """x [B, N, in] -> [B, N, out] on the (differentiable) HIP operators."""
        pf = self._mlp(self.mlp_pre, x)
        B, N, Cn = pf.shape
        scene = AG.segment_mean(pf).view(B, 1, Cn).expand(B, N, Cn)
        return self._mlp(self.mlp_post, torch.cat([pf, scene], dim=-1))


class ContactMLP(nn.Module):
    """`arch: 'MLP'` - the default of configs/model/cdm.yaml (cdm.py:41-85)."""

    def __init__(self, arch_cfg, contact_dim: int, point_feat_dim: int, text_feat_dim: int, time_emb_dim: int) -> None:
        super().__init__()
        layers, idim = [], contact_dim + point_feat_dim + text_feat_dim + time_emb_dim
        for odim in arch_cfg.point_mlp_dims:
            layers.append(PointSceneMLP(idim, odim, widening_factor=arch_cfg.point_mlp_widening_factor, bias=arch_cfg.point_mlp_bias))
            idim = odim
        self.point_mlp = nn.Sequential(*layers)

    def run(self, x, point_feat, language_feat, time_embedding):
        """x [B,N,J], point_feat [B,N,F] or None, language_feat [B,1,Ft], time_embedding [B,1,Te] -> [B,N,last_dim]."""
        B, N, _ = x.shape
        parts = [x] + ([point_feat] if point_feat is not None else []) + [language_feat.expand(B, N, -1), time_embedding.expand(B, N, -1)]
        h = torch.cat(parts, dim=-1)
        for layer in self.point_mlp:
            h = layer.run(h)
        return h


class ContactPointTrans(nn.Module):
    """`arch: 'PointTrans'` / `'PointTransV2'` (cdm.py:190-410): a 4-level Point Transformer U-Net over the noisy contact map
    itself (planes 64..512), the text / time context injected through small MLPs at the bottleneck (V1) or at levels 4, 3, 2
    plus one ReLU transformer-encoder layer over the N/64 bottleneck tokens (V2).  `run`: the fused point kernels of afm.scene
    (eval-mode BatchNorm); `run_train`: the differentiable operator graph."""

    def __init__(self, arch_cfg, contact_dim: int, point_feat_dim: int, text_feat_dim: int, time_emb_dim: int, v2: bool = False) -> None:
        super().__init__()
        from .scene import PointTransformerBlock, TransitionDown, TransitionUp
        self.v2 = v2
        self.num_points = arch_cfg.num_points
        self.c = contact_dim + point_feat_dim + 3
        blocks, planes, share = list(arch_cfg.blocks), [64, 128, 256, 512], 8
        self.strides, self.nsamples = [1, 4, 4, 4], [8, 16, 16, 16]
        self.in_planes = self.c
        for i in range(4):
            layers = [TransitionDown(self.in_planes, planes[i], self.strides[i], self.nsamples[i])]
            self.in_planes = planes[i]
            layers += [PointTransformerBlock(planes[i], planes[i], share, nsample=self.nsamples[i]) for _ in range(1, blocks[i])]
            setattr(self, f"enc{i + 1}", nn.Sequential(*layers))
        for i in (3, 2, 1, 0):
            layers = [TransitionUp(self.in_planes, None if i == 3 else planes[i])]
            self.in_planes = planes[i]
            layers.append(PointTransformerBlock(planes[i], planes[i], share, nsample=self.nsamples[i]))
            setattr(self, f"dec{i + 1}", nn.Sequential(*layers))
        ctx_dim = text_feat_dim + time_emb_dim

        def ctx(c):
            return nn.Sequential(nn.Linear(c + ctx_dim, c), nn.BatchNorm1d(c), nn.ReLU(inplace=True), nn.Linear(c, c))
        if v2:
            self.ctx4, self.ctx3, self.ctx2 = ctx(planes[3]), ctx(planes[2]), ctx(planes[1])
            self.self_attn_layers = nn.TransformerEncoder(
                nn.TransformerEncoderLayer(d_model=planes[-1], nhead=8, dim_feedforward=1024, dropout=0.1, activation="relu", batch_first=True),
                num_layers=1, enable_nested_tensor=False)
        else:
            self.ctx = ctx(planes[3])

    @staticmethod
    def _ctx(seq: nn.Sequential, x, context, batch: int):
        """Linear -> BN(eval) -> ReLU -> Linear over [x | context of the sample] (cdm.py:236-243)."""
        from .scene import _bn_fold
        n = x.shape[0] // batch
        cat = torch.cat((x, context.repeat_interleave(n, dim=0)), 1)
        s, b = _bn_fold(seq[1], seq[0].bias)
        h = ops.linear(cat, seq[0].weight, b, scale=s, act=ffi.ACT_RELU)
        return ops.linear(h, seq[3].weight, seq[3].bias)

    def _encoder_layer_relu(self, x):
        """nn.TransformerEncoderLayer(post-LN, ReLU) over [B, G, 512] (V2 bottleneck, cdm.py:317-327)."""
        l = self.self_attn_layers.layers[0]
        B, G, d = x.shape
        flat = x.reshape(B * G, d)
        a = ops.mha(ops.linear(flat, l.self_attn.in_proj_weight, l.self_attn.in_proj_bias).view(B, G, 3 * d), None, l.self_attn.num_heads)
        y = ops.layernorm(ops.linear(a.view(B * G, d), l.self_attn.out_proj.weight, l.self_attn.out_proj.bias, residual=flat), l.norm1.weight,
                          l.norm1.bias, l.norm1.eps)
        h = ops.linear(y, l.linear1.weight, l.linear1.bias, act=ffi.ACT_RELU)
        return ops.layernorm(ops.linear(h, l.linear2.weight, l.linear2.bias, residual=y), l.norm2.weight, l.norm2.bias, l.norm2.eps).view(B, G, d)

    def run(self, x, point_feat, language_feat, time_embedding, xyz):
        """x [B,N,J], point_feat [B,N,F] | None, language_feat [B,Ft], time_embedding [B,Te], xyz [B,N,3] -> [B,N,64]."""
        from . import pointops
        B, N, _ = x.shape
        if point_feat is not None:
            x = torch.cat([x, point_feat], dim=-1)
        context = torch.cat([language_feat, time_embedding], dim=-1)                       # [B, Ft + Te]
        p0 = ffi.f32c(xyz).reshape(B * N, 3)
        x0 = torch.cat((p0, ffi.f32c(x).reshape(B * N, -1)), 1)
        ps, xs, knns = [], [], []
        for lvl in range(4):
            enc = getattr(self, f"enc{lvl + 1}")
            p0, x0 = enc[0].run(p0, x0, B)
            n = p0.shape[0] // B
            ki, _ = pointops.knn(self.nsamples[lvl], p0, p0, B, n, n)
            for blk in list(enc)[1:]:
                x0 = blk.run(p0, x0, ki)
            ps.append(p0); xs.append(x0); knns.append(ki)
        if self.v2:
            x4 = self._encoder_layer_relu(xs[3].view(B, -1, xs[3].shape[-1])).reshape(xs[3].shape)
            x4 = self._ctx(self.ctx4, x4, context, B)
        else:
            x4 = self._ctx(self.ctx, xs[3], context, B)
        y = self.dec4[1].run(ps[3], self.dec4[0].run_head(x4, B), knns[3])
        for lvl in (2, 1, 0):
            dec = getattr(self, f"dec{lvl + 1}")
            xl = xs[lvl]
            if self.v2 and lvl in (2, 1):
                xl = self._ctx(self.ctx3 if lvl == 2 else self.ctx2, xl, context, B)
            y = dec[1].run(ps[lvl], dec[0].run_fuse(ps[lvl], xl, ps[lvl + 1], y, B), knns[lvl])
        return y.view(B, N, -1)

    # ---- training (cdm.py:190-410 under autograd): the same graph from differentiable HIP operators, BatchNorm per `self.training`
    @staticmethod
    def _ctx_train(seq: nn.Sequential, x, context, batch: int):
        n = x.shape[0] // batch
        cat = torch.cat((x, AP.broadcast_rows(context, n)), 1)
        h = AP.batch_norm(AG.linear(cat, seq[0].weight, seq[0].bias), seq[1], relu=True)
        return AG.linear(h, seq[3].weight, seq[3].bias)

    def run_train(self, x, point_feat, language_feat, time_embedding, xyz, drop_seed: int = 0):
        from . import pointops
        B, N, _ = x.shape
        if point_feat is not None:
            x = torch.cat([x, point_feat], dim=-1)
        context = torch.cat([language_feat, time_embedding], dim=-1)
        p0 = ffi.f32c(xyz).reshape(B * N, 3)
        x0 = torch.cat((p0, ffi.f32c(x).reshape(B * N, -1)), 1)
        ps, xs, knns = [], [], []
        for lvl in range(4):
            enc = getattr(self, f"enc{lvl + 1}")
            p0, x0 = enc[0].run_train(p0, x0, B)
            n = p0.shape[0] // B
            with torch.no_grad():
                ki, _ = pointops.knn(self.nsamples[lvl], p0, p0, B, n, n)
            for blk in list(enc)[1:]:
                x0 = blk.run_train(p0, x0, ki)
            ps.append(p0); xs.append(x0); knns.append(ki)
        if self.v2:
            l = self.self_attn_layers.layers[0]
            p = float(l.dropout.p) if self.training else 0.0
            x4 = AG.encoder_layer(xs[3].view(B, -1, xs[3].shape[-1]), l, None, l.self_attn.num_heads, (p, drop_seed, 0), act=ffi.ACT_RELU)
            x4 = self._ctx_train(self.ctx4, x4.reshape(xs[3].shape), context, B)
        else:
            x4 = self._ctx_train(self.ctx, xs[3], context, B)
        y = self.dec4[1].run_train(ps[3], self.dec4[0].run_head_train(x4, B), knns[3])
        for lvl in (2, 1, 0):
            dec = getattr(self, f"dec{lvl + 1}")
            xl = xs[lvl]
            if self.v2 and lvl in (2, 1):
                xl = self._ctx_train(self.ctx3 if lvl == 2 else self.ctx2, xl, context, B)
            y = dec[1].run_train(ps[lvl], dec[0].run_fuse_train(ps[lvl], xl, ps[lvl + 1], y, B), knns[lvl])
        return y.view(B, N, -1)


class ContactPerceiver(nn.Module):
    """Parameter container with the reference's names (cdm.py:88-153)."""

    def __init__(self, arch_cfg, contact_dim: int, point_feat_dim: int, text_feat_dim: int, time_emb_dim: int) -> None:
        super().__init__()
        a = arch_cfg
        if a.encoder_widening_factor != 1 or a.decoder_widening_factor != 1:
            raise NotImplementedError("widening_factor != 1 is not used by any reference config")
        self.point_pos_emb = a.point_pos_emb
        self.dq, self.dkv = a.encoder_q_input_channels, a.encoder_kv_input_channels
        assert a.decoder_q_input_channels == self.dkv and a.decoder_kv_input_channels == self.dq
        self.enc_heads, self.dec_heads, self.n_self = a.encoder_num_heads, a.decoder_num_heads, a.encoder_self_attn_num_layers
        self.feat_dim = contact_dim + point_feat_dim + (3 if self.point_pos_emb else 0)
        self.language_adapter = nn.Linear(text_feat_dim, self.dq, bias=True)
        self.time_embedding_adapter = nn.Linear(time_emb_dim, self.dq, bias=True)
        self.encoder_adapter = nn.Linear(self.feat_dim, self.dkv, bias=True)
        self.decoder_adapter = nn.Linear(self.dkv, self.dkv, bias=True)
        self.encoder_cross_attn = nn.Sequential(_Wrap(_CrossAttention(self.dq, self.dkv)), _Wrap(_mlp(self.dq, 1)))
        self.encoder_self_attn = nn.Sequential(*[nn.Sequential(_Wrap(_SelfAttention(self.dq)), _Wrap(_mlp(self.dq, 1)))
                                                 for _ in range(self.n_self)])
        self.decoder_cross_attn = nn.Sequential(_Wrap(_CrossAttention(self.dkv, self.dq)), _Wrap(_mlp(self.dkv, 1)))


def _dist_rank() -> int:
    import torch.distributed as dist
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


@Model.register()
class CDM(_FlatParamsMixin, TextEncoderMixin, nn.Module):
    def __init__(self, cfg, *args, **kwargs):
        super().__init__()
        self.device = kwargs["device"] if "device" in kwargs else "cpu"
        self.contact_type = cfg.data_repr
        self.contact_dim = cfg.input_feats
        self.time_emb_dim = cfg.time_emb_dim
        self.timestep_embedder = TimestepEmbedder(self.time_emb_dim, self.time_emb_dim, max_len=1000)
        self.text_model_name = cfg.text_model.version
        self.text_max_length = cfg.text_model.max_length
        self.text_feat_dim, self.text_feat_type = lang_feat_dim_type(self.text_model_name)
        self._init_text_encoder()
        sm = cfg.scene_model
        if not sm.use_scene_model:
            self.point_feat_dim = 0
        elif sm.use_openscene:
            self.point_feat_dim = sm.point_feat_dim
        else:
            # frozen scene backbone (cdm.py:444-446): PointTransformerSeg over (xyz, colour); weights are excluded from
            # checkpoints by the reference's _save (keys containing 'scene_model', utils/training.py:97)
            if sm.name != "PointTransformerSeg":
                raise NotImplementedError(sm.name)
            from .scene import PointTransformerSeg
            self.scene_model_dim = 3 + int(sm.use_color) * 3
            self.scene_model = PointTransformerSeg(c=self.scene_model_dim, num_points=sm.num_points)
            pw = getattr(sm, "pretrained_weight", None)
            if pw:                                               # '' / None = explicit opt-out (tests, synthetic benches)
                import os
                if not os.path.exists(pw):                       # pointtransformer.py:203-205 raises too: a frozen, randomly
                    raise FileNotFoundError(                     # initialised backbone would produce garbage features silently
                        f"Can't find pretrained point-transformer weights: scene_model.pretrained_weight={pw!r} "
                        "(set it to '' to build the frozen backbone without loading weights)")
                sd = torch.load(pw, map_location="cpu")
                self.scene_model.load_state_dict({k: v for k, v in sd.items() if "enc" in k or "dec" in k})
            self.freeze_scene_model = bool(sm.freeze)          # read by TrainLoop._freeze_scene_model_batchnorm (utils/training.py:111-116)
            if not self.freeze_scene_model:
                raise NotImplementedError("scene_model.freeze=False: fine-tuning the PointTransformerSeg backbone is not built "
                                          "(every reference config freezes it)")
            self.scene_model.eval().requires_grad_(False)
            self.point_feat_dim = sm.point_feat_dim
            self._scene_cache = None
        self.arch = cfg.arch
        if self.arch == "Perceiver":
            self.arch_cfg, contact_model = cfg.arch_perceiver, ContactPerceiver
        elif self.arch == "MLP":
            self.arch_cfg, contact_model = cfg.arch_mlp, ContactMLP
        elif self.arch in ("PointTrans", "PointTransV2"):
            import functools
            self.arch_cfg, contact_model = cfg.arch_pointtrans, functools.partial(ContactPointTrans, v2=self.arch == "PointTransV2")
        else:
            raise NotImplementedError(f"arch={self.arch!r}: one of 'MLP', 'Perceiver', 'PointTrans', 'PointTransV2' (cdm.py:449-462)")
        self.contact_model = contact_model(self.arch_cfg, contact_dim=self.contact_dim, point_feat_dim=self.point_feat_dim,
                                           text_feat_dim=self.text_feat_dim, time_emb_dim=self.time_emb_dim)
        self.contact_layer = nn.Linear(self.arch_cfg.last_dim, self.contact_dim, bias=True)
        self._pack = None
        self._text_cache = None
        self._ws = {}
        # Host-side tuning attributes (plain attributes: set them on the instance; nothing is read from the environment).
        self.overlap_streams = True     # layer-by-layer form: decoder-adapter GEMM on a side stream under the latent chain
        # sub-batches of the native loop on their own streams (bit-identical results): 0 = automatic = ONE.  Round 2 ran two from B = 16 on
        # (one sub-batch's launch-latency-bound 2-latent chain under the other's full-chip kernels, +4.7 %); with round 3's chain a third
        # as long a second stream loses (1248 vs 1308 steps/s, profiles/r03_cdm_chain.md)
        self.loop_sub_batches = 0
        if self.arch != "Perceiver":
            self.afm_native_loop = None         # other archs sample step by step
        self.sub_batches = 1                    # per-call sub-batches of forward(): >1 costs more host time per step than it hides (measured)
        self._streams = []
        self.no_fold = False            # measurement: the layer-by-layer sampling form (what training-mode forward also runs)
        self.gemm_tile = 0              # measurement: AFM_TUNE_TILE code forced on the linear1 GEMM of the sampling forms (bit-neutral)
        self.no_gen = False             # measurement: round 2's folded form (per-point rows) instead of the row-less sampling form
        self.chain_side = False         # sub-batch streams: the latent chain of a sub-batch on its side stream (afm_cdm_weights.flags: AFM_CDM_CHAIN_SIDE)
        self.dec_chunks = 0             # dec_point workgroups per sample (0 = 16); bit-neutral tuning
        self.pipeline = False           # sub-batches as a fixed-phase pipeline (AFM_CDM_PIPELINE): all point kernels on one stream in round-robin order, each sub-batch's latent chain on its side stream
        self.chain_cu_mask = None       # CU masks (lists of uint32 words) of the side / main streams of the native loop; None = ordinary streams
        self.point_cu_mask = None

    # ------------------------------------------------------------------ weight pack
    def _weights(self) -> ffi.CdmWeights:
        ver = (_param_version(self), self.training, self.no_fold)
        if self._pack is not None and self._pack[0] == ver:
            w = self._pack[1]
            w.gemm_arith, w.gemm_arith_min_n = ops.gemm_arith()       # host arithmetic setting, per call (afm.ops.set_gemm_split)
            w.flags = self._flags()
            return w
        if self.contact_layer.weight.device.type != "cuda":
            raise ffi.AfmError("CDM parameters are on the CPU; move the model to the MI355X (`model.to('cuda')`)")
        keep: List[torch.Tensor] = []

        def P(t):
            t = ffi.f32c(t.detach())
            keep.append(t)
            return t.data_ptr()

        def lin(dst, m: nn.Linear):
            dst.w, dst.b = P(m.weight), P(m.bias)

        def ln(dst, m: nn.LayerNorm):
            dst.g, dst.b = P(m.weight), P(m.bias)

        def mha(dst, m: _MHA):
            lin(dst.q, m.q_proj); lin(dst.k, m.k_proj); lin(dst.v, m.v_proj); lin(dst.o, m.o_proj)

        def mlp(dst, m: nn.Sequential):
            ln(dst.norm, m[0]); lin(dst.fc1, m[1]); lin(dst.fc2, m[3])

        cm = self.contact_model
        w = ffi.CdmWeights()
        w.gemm_arith, w.gemm_arith_min_n = ops.gemm_arith()
        w.contact_dim, w.feat_dim, w.dq, w.dkv = self.contact_dim, cm.feat_dim, cm.dq, cm.dkv
        w.enc_heads, w.dec_heads, w.n_self = cm.enc_heads, cm.dec_heads, cm.n_self
        w.text_dim, w.time_dim, w.n_timesteps = self.text_feat_dim, self.time_emb_dim, self.timestep_embedder.pe.shape[0]
        lin(w.language_adapter, cm.language_adapter); lin(w.time_embedding_adapter, cm.time_embedding_adapter)
        lin(w.encoder_adapter, cm.encoder_adapter); lin(w.decoder_adapter, cm.decoder_adapter)
        ca = cm.encoder_cross_attn[0].module
        ln(w.enc_q_norm, ca.q_norm); ln(w.enc_kv_norm, ca.kv_norm); mha(w.enc_attn, ca.attention)
        mlp(w.enc_mlp, cm.encoder_cross_attn[1].module)
        for i, layer in enumerate(cm.encoder_self_attn):
            sa = layer[0].module
            ln(w.self_norm[i], sa.norm); mha(w.self_attn[i], sa.attention); mlp(w.self_mlp[i], layer[1].module)
        da = cm.decoder_cross_attn[0].module
        ln(w.dec_q_norm, da.q_norm); ln(w.dec_kv_norm, da.kv_norm); mha(w.dec_attn, da.attention)
        mlp(w.dec_mlp, cm.decoder_cross_attn[1].module)
        lin(w.contact_layer, self.contact_layer)
        # time latent of EVERY timestep (depends on t only): off the per-step path
        table = ffi.f32c(self.timestep_embedder.table())
        q0, u, cu = self._latent_tokens(w, 1, table)
        keep += [table, q0, u, cu]
        w.time_q0, w.time_u, w.time_cu = q0.data_ptr(), u.data_ptr(), cu.data_ptr()
        if not self.training and not self.no_fold and self.contact_dim <= 8 and cm.feat_dim > self.contact_dim and cm.dkv % 64 == 0:
            # eval: weight products of the folded sampling form (afm_cdm_weights.fold_*; float64 products, rounded once).  Only the
            # contact columns of the encoder input change between steps, and encoder_adapter -> decoder_adapter as well as
            # linear2 (+ residual) -> contact_layer are linear chains (cdm.py:176-186,509-510).
            cd = self.contact_dim
            f64 = lambda t: t.detach().double().cpu()        # a few 256 x 256 products: on the host, exactly reproducible
            we, wd = f64(cm.encoder_adapter.weight), f64(cm.decoder_adapter.weight)
            fc2, cl = cm.decoder_cross_attn[1].module[3], self.contact_layer
            wc = f64(cl.weight)
            xu = we[:, :cd].t().contiguous()                                       # [cd, dkv]
            xv = (wd @ we[:, :cd]).t().contiguous()                                # [cd, dkv]
            bo = f64(cm.decoder_cross_attn[0].module.attention.o_proj.bias)
            folds = dict(fold_xu=xu, fold_xv=xv, fold_w2=wc @ f64(fc2.weight), fold_q=wc @ xv.t(),
                         fold_c0=wc @ (f64(fc2.bias) + bo) + f64(cl.bias))
            if cm.feat_dim + 1 <= 44:
                K = 12 if cm.feat_dim + 1 <= 12 else 44                           # inputs of the row-less kernels, padded (RowLess<3> / RowLess<11> in perceiver.hip)
                # generator tables (afm_cdm_weights.gen_*): both adapters as maps of the K = feat_dim + 1 inputs [x_t | features | 1]
                F_, be, bd = cm.feat_dim, f64(cm.encoder_adapter.bias), f64(cm.decoder_adapter.bias)
                gen_enc = torch.zeros(K, cm.dkv, dtype=torch.float64)
                gen_enc[:F_] = we.t(); gen_enc[F_] = be
                gen_dec = torch.zeros(K, cm.dkv, dtype=torch.float64)
                gen_dec[:F_] = (wd @ we).t(); gen_dec[F_] = wd @ be + bd
                folds.update(gen_qe=wc @ gen_dec.t())
                # the decoder of a point in one kernel (afm_cdm_weights.dec_*): everything between the attention weights and linear1 is linear
                # in [a | inputs]; rows of the inputs (step-invariant) here, rows of the attention weights per step on the device
                mlp_m = cm.decoder_cross_attn[1].module
                g2, b2, w1, b1 = f64(mlp_m[0].weight), f64(mlp_m[0].bias), f64(mlp_m[1].weight), f64(mlp_m[1].bias)
                tx = gen_dec.clone()
                tx[F_] += bo                                                       # the attention's output bias rides on the constant input
                xc = tx - tx.mean(1, keepdim=True)
                w1g = w1 * g2[None, :]
                dc = gen_dec - gen_dec.mean(1, keepdim=True)                       # the query row's LayerNorm: var = x Qd x^T, in MFMA operand order
                def operand_order(q):                                              # [K, K] -> [K, 16 NT]: entry (k, 16 t + i) = q[4 (4 t + (i & 3)) + (i >> 2)][k]
                    nt = (K + 15) // 16
                    o = torch.zeros(K, 16 * nt, dtype=torch.float64)
                    for c in range(16 * nt):
                        t, i = divmod(c, 16)
                        src = 4 * (4 * t + (i & 3)) + (i >> 2)
                        if src < K:
                            o[:, c] = q[src, :]
                    return o
                qdd = operand_order(dc @ dc.t() / cm.dkv)
                ec = gen_enc - gen_enc.mean(1, keepdim=True)                       # the encoder side: rows the latents attend over
                folds.update(enc_ec=ec, enc_qee=operand_order(ec @ ec.t() / cm.dkv))
                # head of the latent chain: x1 = q0 + o_proj(v_proj(LayerNorm_kv-weighted sums)) is linear in the 8 x K numbers enc_point_kernel accumulates
                ea = cm.encoder_cross_attn[0].module
                gk, bk = f64(ea.kv_norm.weight), f64(ea.kv_norm.bias)
                wv, bv, wo, bo_e = f64(ea.attention.v_proj.weight), f64(ea.attention.v_proj.bias), f64(ea.attention.o_proj.weight), f64(ea.attention.o_proj.bias)
                hd = cm.dq // 8                                                    # the kernel is written for the reference's 8 encoder heads
                wve = (ec * gk[None, :]) @ wv.t()                                  # [K, dq]: v_proj of gamma * Ec[k]
                wove = torch.zeros(8 * K, cm.dq, dtype=torch.float64)
                for h in range(8):
                    blk = slice(h * hd, (h + 1) * hd)
                    wove[K * h:K * h + K] = wve[:, blk] @ wo[:, blk].t()
                folds.update(enc_wove=wove, enc_c1=bo_e + wo @ (wv @ bk + bv))
                # tail of the chain: the fused decoder's per-sample tables are linear / bilinear in the latents' decoder keys and values
                da_m = cm.decoder_cross_attn[0].module
                gq, bq_n = f64(da_m.q_norm.weight), f64(da_m.q_norm.bias)
                wq_d, bq_d, wo_d = f64(da_m.attention.q_proj.weight), f64(da_m.attention.q_proj.bias), f64(da_m.attention.o_proj.weight)
                woc = wo_d - wo_d.mean(0, keepdim=True)
                wco = torch.zeros(8, cm.dkv, dtype=torch.float64)
                wco[:cd] = wc @ wo_d
                folds.update(dec_dwq=(dc * gq[None, :]) @ wq_d.t(), dec_wqb=wq_d @ bq_n + bq_d, dec_wco=wco, dec_wow=woc.t() @ w1g.t(),
                             dec_wog=woc.t() @ woc / cm.dkv, dec_xwo=xc @ woc / cm.dkv)
                folds.update(dec_c=b1 + w1 @ b2, dec_twx=xc @ w1g.t(), dec_qxx=xc @ xc.t() / cm.dkv, dec_qdd=qdd)
            dev = cl.weight.device
            for name, t in folds.items():
                setattr(w, name, P(t.float().contiguous().to(dev)))
            # LayerNorm folded into the latent-chain stages that follow one (afm_cdm_weights.lat_fold: 19 slots of (W * gamma, row sums, b + W beta))
            slots = [None] * 19

            def fold_ln(slot, lin_m, ln_m):
                wl, bl, gl, btl = f64(lin_m.weight), f64(lin_m.bias), f64(ln_m.weight), f64(ln_m.bias)
                wg = wl * gl[None, :]
                slots[slot] = (wg, wg.sum(1), bl + wl @ btl)
            fold_ln(0, cm.encoder_cross_attn[1].module[1], cm.encoder_cross_attn[1].module[0])
            for li, layer in enumerate(cm.encoder_self_attn):
                sa_m, mlp_l = layer[0].module, layer[1].module
                fold_ln(1 + 4 * li, sa_m.attention.q_proj, sa_m.norm); fold_ln(2 + 4 * li, sa_m.attention.k_proj, sa_m.norm)
                fold_ln(3 + 4 * li, sa_m.attention.v_proj, sa_m.norm); fold_ln(4 + 4 * li, mlp_l[1], mlp_l[0])
            da_kv = cm.decoder_cross_attn[0].module
            fold_ln(17, da_kv.attention.k_proj, da_kv.kv_norm); fold_ln(18, da_kv.attention.v_proj, da_kv.kv_norm)
            table = (C.c_void_p * (3 * 19))()
            for si, trip in enumerate(slots):
                for j in range(3):
                    table[3 * si + j] = P(trip[j].float().contiguous().to(dev)) if trip is not None else None
            keep.append(table)
            w.lat_fold = C.cast(table, C.c_void_p)
        self._pack = (ver, w, keep)
        self._text_cache = None
        w.gemm_arith, w.gemm_arith_min_n = ops.gemm_arith()
        w.flags = self._flags()
        return w

    def _flags(self) -> int:
        return (ffi.CDM_NO_GEN if self.no_gen else 0) | ((int(self.gemm_tile) & 0xF) << 8) | (ffi.CDM_CHAIN_SIDE if self.chain_side else 0) | \
            ((int(self.dec_chunks) & 0x3F) << ffi.CDM_DEC_CHUNKS_SHIFT) | (ffi.CDM_PIPELINE if self.pipeline else 0)

    def _loop_streams(self, need: int, dev):
        """Streams of the native loop: [main_0, side_0, main_1, side_1, ...].  With `chain_cu_mask` / `point_cu_mask` (lists of 32-bit words,
        one bit per CU) the side / main streams are created with hipExtStreamCreateWithCUMask, so that the latent chain of one sub-batch owns
        a few CUs while the point kernels of the other sub-batch fill the rest (round 4 experiment; results do not depend on it)."""
        key = (tuple(self.chain_cu_mask or ()), tuple(self.point_cu_mask or ()), str(dev))
        if not self.chain_cu_mask and not self.point_cu_mask:
            if self.pipeline and need > 1:
                # AFM_CDM_PIPELINE: the point kernels run on the caller's stream, sub-batch s's chain on entry 2 s + 1: 1 + nsub streams in all
                pool = ffi.stream_pool(dev, need // 2)
                return [pool[i // 2] for i in range(need)]
            return ffi.stream_pool(dev, need)              # process-wide pool (hardware queues are few: ffi.stream_pool)
        if getattr(self, "_streams_key", None) != key:
            self._streams, self._streams_key = [], key
        while len(self._streams) < need:
            i = len(self._streams)
            mask = self.chain_cu_mask if (i & 1) else self.point_cu_mask
            self._streams.append(ffi.cu_masked_stream(dev, mask) if mask else torch.cuda.Stream(device=dev))
        return self._streams[:need]

    def _latent_tokens(self, w, which: int, rows: torch.Tensor):
        """afm_cdm_latent_tokens: rows [n, in_dim] -> (q0 [n,dq], u [n,He,dkv], cu [n,He])."""
        n, cm = rows.shape[0], self.contact_model
        q0 = torch.empty(n, cm.dq, device=rows.device); u = torch.empty(n, cm.enc_heads, cm.dkv, device=rows.device)
        cu = torch.empty(n, cm.enc_heads, device=rows.device)
        ffi.check(ffi.load().afm_cdm_latent_tokens(C.byref(w), which, rows.data_ptr(), n, q0.data_ptr(), u.data_ptr(), cu.data_ptr(),
                                                   ffi.stream_of(rows)), "afm_cdm_latent_tokens")
        return q0, u, cu

    def _text_latent(self, w, kwargs, device):
        """Text latent of every sample; cached while the same text tensor / strings are passed (the sampling loop)."""
        tf = kwargs.get("c_text_feat")
        extra = (None if isinstance(tf, torch.Tensor) else tuple(kwargs["c_text"]), _param_version(self))
        if getattr(self, "_text_cache", None) is not None and self._text_cache[0].matches((tf,), extra):
            return self._text_cache[1]
        text = ffi.f32c(self.encode_text(kwargs).to(device))
        lat = self._latent_tokens(w, 0, text)
        self._text_cache = (HeldKey((tf,), extra), lat, text)
        return lat

    def _text_similarity(self, pf: torch.Tensor, kwargs, like) -> torch.Tensor:
        """openscene text-similarity feature (cdm.py:500-503): pc_emb[b, n, 0] = <c_pc_feat[b, n, :], text_emb[b, :]> - one afm_linear per sample
        ([n, d] x [1, d]: the thin-layer stream kernel), step-invariant, so cached like the backbone features while the same tensors are passed."""
        tf = kwargs.get("c_text_feat")
        extra = (None if isinstance(tf, torch.Tensor) else tuple(kwargs["c_text"]),)
        cache = getattr(self, "_sim_cache", None)
        if cache is not None and cache[0].matches((pf, tf), extra):
            return cache[1]
        text = ffi.f32c(self.encode_text(kwargs).to(like.device))                     # [B, d]
        pfc = ffi.f32c(pf.to(like))
        if pfc.shape[-1] != text.shape[-1]:
            raise ffi.AfmError(f"openscene text-similarity: point features are {pfc.shape[-1]}-wide, the text feature {text.shape[-1]}-wide")
        sim = torch.stack([ops.linear(pfc[b], text[b:b + 1]) for b in range(pfc.shape[0])])      # [B, n, 1]
        self._sim_cache = (HeldKey((pf, tf), extra), sim)
        return sim

    def _scene_features(self, xyz, col, like) -> torch.Tensor:
        """Frozen PointTransformerSeg features of the scene batch; step-invariant (the reference re-runs the backbone in every step,
        cdm.py:508), so cached while the SAME live tensors are passed again (`HeldKey` keeps them alive: a freed batch's address
        handed to the next batch can therefore never match)."""
        ver = (_param_version(self.scene_model),)
        if self._scene_cache is None or not self._scene_cache[0].matches((xyz, col), ver):
            self._scene_cache = (HeldKey((xyz, col), ver), self.scene_model((xyz.to(like), None if col is None else col.to(like))))
        return self._scene_cache[1]

    def _features(self, x, kwargs) -> torch.Tensor:
        """cat(x_t, per-point features, xyz) exactly as cdm.py:495-505 + ContactPerceiver.forward :167-171."""
        parts = [x]
        if hasattr(self, "scene_model"):
            # step-invariant: the reference re-runs the frozen backbone in every step (cdm.py:508); cache per scene batch
            xyz, col = kwargs["c_pc_xyz"], kwargs.get("c_pc_feat")
            parts.append(self._scene_features(xyz, col, x))
        elif self.point_feat_dim > 0:
            pf = kwargs["c_pc_feat"]
            parts.append(self._text_similarity(pf, kwargs, x) if (self.point_feat_dim == 1 and pf.shape[-1] != 1) else pf.to(x))
        if self.contact_model.point_pos_emb:
            parts.append(kwargs["c_pc_xyz"].to(x))
        return torch.cat(parts, dim=-1).contiguous()

    def forward(self, x, timesteps, **kwargs):
        """x [B, N, contact_dim], timesteps [B] -> predicted x_0 (same shape)."""
        if self.arch == "MLP":
            return self.forward_mlp(x, timesteps, **kwargs)
        if self.arch in ("PointTrans", "PointTransV2"):
            return self.forward_pointtrans(x, timesteps, **kwargs)
        if torch.is_grad_enabled() and (self.training or any(p.requires_grad for p in self.parameters())):
            return self.forward_train(x, timesteps, **kwargs)
        ffi.require_gpu(x)
        with torch.no_grad():
            lib = ffi.load()
            x = ffi.f32c(x)
            B, N, _ = x.shape
            w = self._weights()
            feat = self._features(x, kwargs)
            tq0, tu, tcu = self._text_latent(w, kwargs, x.device)
            t = timesteps.to(device=x.device, dtype=torch.int64).contiguous()
            out = torch.empty_like(x)
            # Two levels of concurrency hide the per-sample latent chain (one workgroup per sample, ~0.5 ms serial): inside a
            # call the decoder-adapter GEMM runs on a side stream underneath it (AFM_CDM_OVERLAP=0 disables), and the batch is
            # can be split into sub-batches on their own streams (AFM_CDM_SUBBATCH=n; off by default: driven from Python it costs
            # more host time per step than it hides).  Results are identical: samples are independent.
            nsub = max(1, min(self.sub_batches, B))
            bounds = [(B * i // nsub, B * (i + 1) // nsub) for i in range(nsub)]
            cur = torch.cuda.current_stream(x.device)
            if len(self._streams) < 2 * nsub:
                self._streams = ffi.stream_pool(x.device, 2 * nsub)       # process-wide pool (hardware queues are few: ffi.stream_pool)
            fork = None
            if nsub > 1:
                fork = torch.cuda.Event()
                fork.record(cur)
            for i, (lo, hi) in enumerate(bounds):
                n = hi - lo
                key = (n, N, i, str(x.device))
                if key not in self._ws:
                    nbytes = lib.afm_cdm_workspace_bytes(C.byref(w), n, N)
                    if nbytes < 0:
                        ffi.check(int(nbytes), "afm_cdm_workspace_bytes")
                    self._ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
                ws = self._ws[key]
                main = cur if nsub == 1 else self._streams[2 * i]
                if fork is not None:
                    main.wait_event(fork)
                side = self._streams[2 * i + 1].cuda_stream if self.overlap_streams else None
                ffi.check(lib.afm_cdm_forward_overlap(C.byref(w), feat[lo:hi].data_ptr(), x[lo:hi].data_ptr(), t[lo:hi].data_ptr(),
                                                      tq0[lo:hi].data_ptr(), tu[lo:hi].data_ptr(), tcu[lo:hi].data_ptr(), out[lo:hi].data_ptr(), None,
                                                      n, N, ws.data_ptr(), ws.numel(), side, main.cuda_stream), "afm_cdm_forward_overlap")
                if fork is not None:
                    done = torch.cuda.Event()
                    done.record(main)
                    cur.wait_event(done)
        return out

    # ------------------------------------------------------------------ native sampling loop
    def afm_native_loop(self, diffusion, x, model_kwargs, *, step_noise=None, seed=0, sample_index0=0, progress=False, snapshots=None,
                        clip_denoised=False):
        """Whole p_sample_loop of the ADM on the device (afm_cdm_sample_loop): x holds x_T on entry, returns the sample.  ``progress``
        slices the chain (afm_cdm_sample_loop_range) so a tqdm bar can advance, with bit-identical results.  The batch
        runs as `loop_sub_batches` sub-batches on their own stream pairs (see __init__; bit-identical results).  ``snapshots`` =
        {executed step count: None} is filled with clones of x after those steps, as in CMDM.afm_native_loop."""
        if self.arch != "Perceiver":
            raise NotImplementedError("the native loop covers the Perceiver arch")
        lib = ffi.load()
        ffi.require_gpu(x)
        with torch.no_grad():
            x = ffi.f32c(x)
            B, N, _ = x.shape
            dev = x.device
            w = self._weights()
            feat = self._features(x, model_kwargs)
            tq0, tu, tcu = self._text_latent(w, model_kwargs, dev)
            tab = diffusion.tables(dev)
            n = diffusion.num_timesteps
            sched = ffi.sched_scratch(self, n, B, dev)
            nsub = int(self.loop_sub_batches) or 1
            nsub = max(1, min(nsub, B))
            need = 2 * nsub if nsub > 1 else (1 if self.overlap_streams else 0)
            streams = self._loop_streams(need, dev)
            handles = (C.c_void_p * max(need, 1))(*[s_.cuda_stream for s_ in streams]) if need else None
            nbytes = lib.afm_cdm_loop_workspace_bytes(C.byref(w), B, N, nsub)
            if nbytes < 0:
                ffi.check(int(nbytes), "afm_cdm_loop_workspace_bytes")
            key = ("loop", B, N, nsub, str(dev))
            if key not in self._ws or self._ws[key].numel() < nbytes:        # (the row-less form's workspace is smaller than the other forms')
                self._ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            ws = self._ws[key]
            if step_noise is not None:
                step_noise = ffi.f32c(step_noise.to(dev))
                assert step_noise.shape == (n,) + tuple(x.shape), step_noise.shape
            stream = ffi.stream_of(x)
            if clip_denoised:
                w.flags |= ffi.CDM_CLIP_X0                   # per call: the next _weights() rewrites the flags

            def enqueue(j0, j1):        # executed steps j0..j1-1 = timestep indices n-j1 .. n-1-j0
                lo, cnt = n - j1, j1 - j0
                ffi.check(lib.afm_cdm_sample_loop_range(
                    C.byref(w), x.data_ptr(), feat.data_ptr(), tq0.data_ptr(), tu.data_ptr(), tcu.data_ptr(),
                    None if step_noise is None else step_noise[j0:j1].data_ptr(), tab.timestep_map[lo:].data_ptr(),
                    tab.coef1[lo:].data_ptr(), tab.coef2[lo:].data_ptr(), tab.sigma[lo:].data_ptr(), cnt, j0, seed & (2**64 - 1),
                    sample_index0, B, N, sched.data_ptr(), ws.data_ptr(), ws.numel(), nsub, handles, stream), "afm_cdm_sample_loop_range")

            slices = ffi.progress_slices(n, progress)
            if snapshots is not None:
                slices = ffi.cut_slices(slices, sorted(k for k in snapshots if 0 < k < n))

                def enqueue_snap(j0, j1, _inner=enqueue):
                    _inner(j0, j1)
                    if j1 in snapshots:
                        snapshots[j1] = x.clone()
                ffi.run_slices(slices, enqueue_snap, progress, dev)
            else:
                ffi.run_slices(slices, enqueue, progress, dev)
            self._last_loop_scratch = (sched, step_noise, feat, tq0, tu, tcu)
        return x

    # ------------------------------------------------------------------ 'MLP' arch (per-operator composition, inference and training)
    def _point_features(self, x, kwargs):
        """pc_emb of CDM.forward (cdm.py:495-508): frozen scene backbone output, given per-point features, or None."""
        if hasattr(self, "scene_model"):
            xyz, col = kwargs["c_pc_xyz"], kwargs.get("c_pc_feat")
            return self._scene_features(xyz, col, x)
        if self.point_feat_dim > 0:
            pf = kwargs["c_pc_feat"]
            return self._text_similarity(pf, kwargs, x) if (self.point_feat_dim == 1 and pf.shape[-1] != 1) else pf.to(x)
        return None

    def forward_mlp(self, x, timesteps, **kwargs):
        """CDM.forward with ContactMLP (cdm.py:41-85,474-513): every op is a (differentiable) HIP operator; runs under
        torch.no_grad() for sampling and with the tape for training."""
        ffi.require_gpu(x)
        x = ffi.f32c(x)
        B, N, _ = x.shape
        dev = x.device
        te = self.timestep_embedder
        t_idx = timesteps.to(device=dev, dtype=torch.int64)
        time_emb = AG.linear(AG.linear(te.pe[t_idx, 0, :], te.time_embed[0].weight, te.time_embed[0].bias, act=ffi.ACT_SILU),
                             te.time_embed[2].weight, te.time_embed[2].bias).view(B, 1, -1)
        text = ffi.f32c(self.encode_text(kwargs).to(dev)).view(B, 1, -1)
        h = self.contact_model.run(x, self._point_features(x, kwargs), text, time_emb)
        return AG.linear(h, self.contact_layer.weight, self.contact_layer.bias)

    def forward_pointtrans(self, x, timesteps, **kwargs):
        """CDM.forward with ContactPointTrans / V2 (cdm.py:190-410,474-513): fused eval-mode kernels under no_grad, the differentiable
        operator graph with the tape otherwise."""
        ffi.require_gpu(x)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            x = ffi.f32c(x)
            dev = x.device
            te = self.timestep_embedder
            t_idx = timesteps.to(device=dev, dtype=torch.int64)
            time_emb = AG.linear(AG.linear(te.pe[t_idx, 0, :], te.time_embed[0].weight, te.time_embed[0].bias, act=ffi.ACT_SILU),
                                 te.time_embed[2].weight, te.time_embed[2].bias)                                 # [B, te]
            text = ffi.f32c(self.encode_text(kwargs).to(dev))
            self._train_calls = getattr(self, "_train_calls", 0) + 1
            h = self.contact_model.run_train(x, self._point_features(x, kwargs), text, time_emb, kwargs["c_pc_xyz"].to(x),
                                             drop_seed=self._train_calls)
            B, N, c = h.shape
            return AG.linear(h.reshape(B * N, c), self.contact_layer.weight, self.contact_layer.bias).view(B, N, -1)
        with torch.no_grad():
            x = ffi.f32c(x)
            dev = x.device
            t_idx = timesteps.to(device=dev, dtype=torch.int64)
            time_emb = self.timestep_embedder.table()[t_idx]                                   # [B, te]
            text = ffi.f32c(self.encode_text(kwargs).to(dev))
            h = self.contact_model.run(x, self._point_features(x, kwargs), text, time_emb, kwargs["c_pc_xyz"].to(x))
            return ops.linear(h, self.contact_layer.weight, self.contact_layer.bias)

    # ------------------------------------------------------------------ training forward (autograd tape over HIP kernels)
    def _mha_train(self, att: _MHA, xq, xkv, heads: int, drop, kind: str, residual):
        """MultiHeadAttention.forward (modules.py:301-381) on differentiable HIP ops; kind selects the attention kernel."""
        q = AG.linear(xq, att.q_proj.weight, att.q_proj.bias)
        k = AG.linear(xkv, att.k_proj.weight, att.k_proj.bias)
        v = AG.linear(xkv, att.v_proj.weight, att.v_proj.bias)
        if kind == "few_query":
            o = AG.few_query_attention(q, k, v, heads, drop)
        elif kind == "few_key":
            o = AG.few_key_attention(q, k, v, heads, drop)
        else:
            o = AG.self_attention(torch.cat([q, k, v], dim=-1), heads, drop)
        return AG.linear(o, att.o_proj.weight, att.o_proj.bias, residual=residual)           # Residual wrapper (modules.py:222-231)

    @staticmethod
    def _mlp_train(mlp: nn.Sequential, x):
        """x + MLP(x) (modules.py:651-661 inside Residual)."""
        h = AG.linear(AG.layer_norm(x, mlp[0]), mlp[1].weight, mlp[1].bias, act=ffi.ACT_GELU)
        return AG.linear(h, mlp[3].weight, mlp[3].bias, residual=x)

    def forward_train(self, x, timesteps, **kwargs):
        """CDM.forward + ContactPerceiver.forward (cdm.py:474-513,155-188) as written, composed from differentiable HIP ops so
        that `training_losses(...)['loss'].mean().backward()` fills `.grad` of every trainable parameter.  Train mode applies the
        attention-probability dropout of the three attention types (arch_perceiver.*_dropout); residual dropout must be 0
        (as in every reference config)."""
        ffi.require_gpu(x)
        a = self.arch_cfg
        if float(a.encoder_residual_dropout) != 0.0 or float(a.decoder_residual_dropout) != 0.0:
            raise NotImplementedError("residual dropout != 0 is not used by any reference config")
        x = ffi.f32c(x)
        B, N, _ = x.shape
        cm, dev = self.contact_model, x.device
        p_enc = float(a.encoder_dropout) if self.training else 0.0
        p_dec = float(a.decoder_dropout) if self.training else 0.0
        self._drop_calls = getattr(self, "_drop_calls", 0) + 1
        seed = (torch.initial_seed() + 0x9E3779B97F4A7C15 * self._drop_calls + 0xD1B54A32D192ED03 * _dist_rank()) & (2**64 - 1)   # per call, per rank
        te = self.timestep_embedder
        t_idx = timesteps.to(device=dev, dtype=torch.int64)
        time_emb = AG.linear(AG.linear(te.pe[t_idx, 0, :], te.time_embed[0].weight, te.time_embed[0].bias, act=ffi.ACT_SILU),
                             te.time_embed[2].weight, te.time_embed[2].bias)                                # [B, te]
        text = ffi.f32c(self.encode_text(kwargs).to(dev))
        feat = self._features(x, kwargs)                                                                     # [B, N, feat]
        enc_kv = AG.linear(feat, cm.encoder_adapter.weight, cm.encoder_adapter.bias)                         # [B, N, dkv]
        enc_q = torch.cat([AG.linear(text, cm.language_adapter.weight, cm.language_adapter.bias).view(B, 1, cm.dq),
                           AG.linear(time_emb, cm.time_embedding_adapter.weight, cm.time_embedding_adapter.bias).view(B, 1, cm.dq)], dim=1)
        ca = cm.encoder_cross_attn[0].module
        h = self._mha_train(ca.attention, AG.layer_norm(enc_q, ca.q_norm), AG.layer_norm(enc_kv, ca.kv_norm), cm.enc_heads, (p_enc, seed, 1),
                            "few_query", enc_q)
        enc_q = self._mlp_train(cm.encoder_cross_attn[1].module, h)
        for i, layer in enumerate(cm.encoder_self_attn):
            sa = layer[0].module
            xn = AG.layer_norm(enc_q, sa.norm)
            h = self._mha_train(sa.attention, xn, xn, cm.enc_heads, (p_enc, seed, 2 + i), "self", enc_q)
            enc_q = self._mlp_train(layer[1].module, h)
        dec_q = AG.linear(enc_kv, cm.decoder_adapter.weight, cm.decoder_adapter.bias)                        # [B, N, dkv]
        da = cm.decoder_cross_attn[0].module
        h = self._mha_train(da.attention, AG.layer_norm(dec_q, da.q_norm), AG.layer_norm(enc_q, da.kv_norm), cm.dec_heads, (p_dec, seed, 8),
                            "few_key", dec_q)
        dec_q = self._mlp_train(cm.decoder_cross_attn[1].module, h)
        return AG.linear(dec_q, self.contact_layer.weight, self.contact_layer.bias)
