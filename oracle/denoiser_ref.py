"""TEST INFRASTRUCTURE ONLY - CPU restatement of the two denoisers on the hot path:
CMDM (`trans_enc`) and CDM (`Perceiver`), eval mode, float32.

Functional style over a flat state dict with the reference's key names
(SURVEY.md §8b); citations are into /root/reference/models/.  The frozen CLIP
text encoder is outside the path: both functions take the pooled text feature
``text_feat [B, 512]`` (what `encode_text_clip` returns, functions.py:62-84).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import scene_ref

SD = Dict[str, torch.Tensor]


def sinusoid_table(max_len: int, d: int) -> torch.Tensor:
    """modules.py:10-26 -> [max_len, d] (the reference stores it as [max_len, 1, d])."""
    pe = torch.zeros(max_len, d)
    pos = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    div = torch.exp(torch.arange(0, d, 2).float() * (-math.log(10000.0) / d))
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def _lin(sd, pre, x):
    w = sd[pre + ".weight"]
    return F.linear(x.to(w.dtype), w, sd.get(pre + ".bias"))          # (float32 tables / features into a float64 state dict: train_ref's double runs)


def _ln(sd, pre, x):
    w = sd[pre + ".weight"]
    return F.layer_norm(x, (w.shape[0],), w, sd[pre + ".bias"], 1e-5)


def timestep_embed(sd: SD, pre: str, t: torch.Tensor, te_dim: int, max_len: int = 1000) -> torch.Tensor:
    """modules.py:38-53: pe[t] -> Linear -> SiLU -> Linear, [B, 1, d]."""
    e = sinusoid_table(max_len, te_dim)[t].unsqueeze(1)
    pre = pre + "." if pre else ""
    return _lin(sd, pre + "time_embed.2", F.silu(_lin(sd, pre + "time_embed.0", e)))


def encoder_layer(sd: SD, pre: str, x: torch.Tensor, key_padding_mask: Optional[torch.Tensor], nhead: int):
    """nn.TransformerEncoderLayer(batch_first, post-LN, exact-erf GELU, eps 1e-5) as the
    reference builds it (cmdm.py:66-77); masked keys get -inf before the softmax."""
    B, T, D = x.shape
    dh = D // nhead
    qkv = F.linear(x, sd[pre + ".self_attn.in_proj_weight"], sd[pre + ".self_attn.in_proj_bias"])
    q, k, v = (z.view(B, T, nhead, dh).transpose(1, 2) for z in qkv.chunk(3, dim=-1))
    s = (q * (dh ** -0.5)) @ k.transpose(-1, -2)                        # [B, h, T, T]
    if key_padding_mask is not None:
        s = s.masked_fill(key_padding_mask[:, None, None, :], float("-inf"))
    a = torch.softmax(s, dim=-1) @ v
    a = a.transpose(1, 2).reshape(B, T, D)
    x = _ln(sd, pre + ".norm1", x + _lin(sd, pre + ".self_attn.out_proj", a))
    ff = _lin(sd, pre + ".linear2", F.gelu(_lin(sd, pre + ".linear1", x)))
    return _ln(sd, pre + ".norm2", x + ff)


def cmdm_forward(sd: SD, x, t, text_feat, c_pc_xyz=None, c_pc_contact=None, x_mask=None, *,
                 time_emb_dim: int = 512, nhead: int = 8, num_layers: int = 5,
                 blocks=(2, 2, 2, 2), mask_motion: bool = True, cont_emb: Optional[torch.Tensor] = None,
                 return_tokens: bool = False, c_text_mask=None, c_text_erase=None, c_pc_mask=None, c_pc_erase=None):
    """CMDM.forward, `trans_enc` branch (cmdm.py:118-170,195).

    ``cont_emb`` (the SceneMapEncoder output, [B, G, planes[-1]]) may be passed in to
    skip the step-invariant point-cloud encoder; otherwise it is computed from
    (c_pc_xyz, c_pc_contact).
    """
    B, L, _ = x.shape
    d = sd["motion_adapter.weight"].shape[0]
    time_emb = timestep_embed(sd, "timestep_embedder", t, time_emb_dim)           # [B,1,d]
    text = text_feat.unsqueeze(1).float()
    if c_text_erase is not None:                                                   # cmdm.py:144-145
        text = text * (1.0 - c_text_erase.unsqueeze(-1).float())
    text_emb = _lin(sd, "language_adapter", text)                                  # [B,1,d]
    if cont_emb is None:
        cont_emb = scene_ref.scene_map_encoder(sd, "contact_encoder", c_pc_xyz, c_pc_contact, blocks=blocks)
    if c_pc_erase is not None:                                                     # cmdm.py:154-155
        cont_emb = cont_emb * (1.0 - c_pc_erase.unsqueeze(-1).float())
    cont = _lin(sd, "contact_adapter", cont_emb)                                   # [B,G,d]
    G = cont.shape[1]
    seq = torch.cat([time_emb, text_emb, cont, _lin(sd, "motion_adapter", x)], dim=1)
    T = seq.shape[1]
    seq = seq + sinusoid_table(5000, d)[:T].unsqueeze(0)                           # PositionalEncoding, dropout off
    mask = None
    if mask_motion:
        if x_mask is None:
            x_mask = torch.zeros(B, L, dtype=torch.bool)
        tm = torch.zeros(B, 1, dtype=torch.bool) if c_text_mask is None else c_text_mask.bool().reshape(B, 1)      # cmdm.py:142-143
        cm = torch.zeros(B, G, dtype=torch.bool) if c_pc_mask is None else c_pc_mask.bool().reshape(B, 1).repeat(1, G)   # :152-153
        mask = torch.cat([torch.zeros(B, 1, dtype=torch.bool), tm, cm, x_mask], dim=1)
    for i in range(num_layers):
        seq = encoder_layer(sd, f"self_attn_layer.layers.{i}", seq, mask, nhead)
    out = _lin(sd, "motion_layer", seq[:, 2 + G:, :])
    return (out, seq) if return_tokens else out


# ------------------------------------------------------------------ CDM / Perceiver

def _mha(sd, pre, x_q, x_kv, nhead):
    """modules.py:301-381 (no mask / rotary / cache / causal - never used by this repo)."""
    q, k, v = _lin(sd, pre + ".q_proj", x_q), _lin(sd, pre + ".k_proj", x_kv), _lin(sd, pre + ".v_proj", x_kv)
    B, Nq, C = q.shape
    split = lambda z: z.view(z.shape[0], z.shape[1], nhead, z.shape[2] // nhead).transpose(1, 2)
    q, k, v = split(q), split(k), split(v)
    q = q * ((C // nhead) ** -0.5)
    a = torch.softmax(torch.einsum("bhic,bhjc->bhij", q, k), dim=-1)
    o = torch.einsum("bhij,bhjc->bhic", a, v).transpose(1, 2).reshape(B, Nq, -1)
    return _lin(sd, pre + ".o_proj", o)


def _mlp(sd, pre, x):
    """modules.py:651-661: LN -> Linear -> GELU -> Linear (widening 1)."""
    return _lin(sd, pre + ".3", F.gelu(_lin(sd, pre + ".1", _ln(sd, pre + ".0", x))))


def cross_attention_layer(sd, pre, x_q, x_kv, nhead):
    """modules.py:504-541 + Residual (:222-231): pre-LN on q and kv, residual adds the raw q."""
    a = pre + ".0.module"
    h = _mha(sd, a + ".attention", _ln(sd, a + ".q_norm", x_q), _ln(sd, a + ".kv_norm", x_kv), nhead) + x_q
    return _mlp(sd, pre + ".1.module", h) + h


def self_attention_layer(sd, pre, x, nhead):
    """modules.py:544-578."""
    a = pre + ".0.module"
    xn = _ln(sd, a + ".norm", x)
    h = _mha(sd, a + ".attention", xn, xn, nhead) + x
    return _mlp(sd, pre + ".1.module", h) + h


def cdm_forward(sd: SD, x, t, text_feat, c_pc_xyz, pc_emb=None, *, time_emb_dim: int = 128,
                enc_heads: int = 8, dec_heads: int = 8, self_layers: int = 2, point_pos_emb: bool = True):
    """CDM.forward + ContactPerceiver.forward (cdm.py:474-513, 155-188).

    ``pc_emb`` = optional per-point scene feature [B, N, F] (the frozen scene model's
    output in the HUMANISE variant); None for the H3D variant (`use_scene_model=False`).
    """
    cm = "contact_model"
    time_emb = timestep_embed(sd, "timestep_embedder", t, time_emb_dim)           # [B,1,te]
    text = text_feat.unsqueeze(1).float()
    feat = x if pc_emb is None else torch.cat([x, pc_emb], dim=-1)
    if point_pos_emb:
        feat = torch.cat([feat, c_pc_xyz], dim=-1)
    enc_kv = _lin(sd, cm + ".encoder_adapter", feat)                               # [B,N,256]
    enc_q = torch.cat([_lin(sd, cm + ".language_adapter", text),
                       _lin(sd, cm + ".time_embedding_adapter", time_emb)], dim=1)  # [B,2,512]
    enc_q = cross_attention_layer(sd, cm + ".encoder_cross_attn", enc_q, enc_kv, enc_heads)
    for l in range(self_layers):
        enc_q = self_attention_layer(sd, f"{cm}.encoder_self_attn.{l}", enc_q, enc_heads)
    dec_q = _lin(sd, cm + ".decoder_adapter", enc_kv)                              # [B,N,256]
    dec_q = cross_attention_layer(sd, cm + ".decoder_cross_attn", dec_q, enc_q, dec_heads)
    return _lin(sd, "contact_layer", dec_q)


def cdm_mlp_forward(sd: SD, x, t, text_feat, pc_emb=None, *, time_emb_dim: int = 128, n_layers: int = 2):
    """CDM.forward with ContactMLP (cdm.py:13-85,474-513): per-point MLP over [x | point feature | text | time], each
    PointSceneMLP concatenating the sample's mean feature before its second MLP."""
    B, N, _ = x.shape
    time_emb = timestep_embed(sd, "timestep_embedder", t, time_emb_dim)           # [B,1,te]
    text = text_feat.unsqueeze(1).float()
    parts = [x] + ([pc_emb] if pc_emb is not None else []) + [text.repeat(1, N, 1), time_emb.repeat(1, N, 1)]
    h = torch.cat(parts, dim=-1)

    def mlp(pre, z):
        return _lin(sd, pre + ".3", F.gelu(_lin(sd, pre + ".1", _ln(sd, pre + ".0", z))))
    for i in range(n_layers):
        pre = f"contact_model.point_mlp.{i}"
        pf = mlp(pre + ".mlp_pre", h)
        scene = pf.mean(dim=1, keepdim=True).repeat(1, N, 1)
        h = mlp(pre + ".mlp_post", torch.cat([pf, scene], dim=-1))
    return _lin(sd, "contact_layer", h)


def _ctx_mlp(sd, pre, x, context, batch):
    """cdm.py:236-243: Linear -> BN (eval) -> ReLU -> Linear over [x | context of the sample]; x [(b n), c]."""
    n = x.shape[0] // batch
    h = torch.cat((x, context.repeat_interleave(n, dim=0)), 1)
    h = F.relu(scene_ref._bn(sd, pre + ".1", _lin(sd, pre + ".0", h)))
    return _lin(sd, pre + ".3", h)


def encoder_layer_relu(sd: SD, pre: str, x: torch.Tensor, nhead: int):
    """nn.TransformerEncoderLayer(batch_first, post-LN, ReLU) without mask (ContactPointTransV2 bottleneck)."""
    B, T, d = x.shape
    qkv = F.linear(x, sd[pre + ".self_attn.in_proj_weight"], sd[pre + ".self_attn.in_proj_bias"])
    q, k, v = [z.view(B, T, nhead, d // nhead).transpose(1, 2) for z in qkv.split(d, dim=-1)]
    a = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(d // nhead), dim=-1) @ v
    a = a.transpose(1, 2).reshape(B, T, d)
    x = _ln(sd, pre + ".norm1", x + _lin(sd, pre + ".self_attn.out_proj", a))
    return _ln(sd, pre + ".norm2", x + _lin(sd, pre + ".linear2", F.relu(_lin(sd, pre + ".linear1", x))))


def cdm_pointtrans_forward(sd: SD, x, t, text_feat, c_pc_xyz, pc_emb=None, *, v2: bool = False, time_emb_dim: int = 128,
                           blocks=(2, 2, 2, 2)):
    """CDM.forward with ContactPointTrans / ContactPointTransV2 (cdm.py:190-410, 474-513), eval mode."""
    sr = scene_ref
    cm = "contact_model"
    B, N, _ = x.shape
    time_emb = timestep_embed(sd, "timestep_embedder", t, time_emb_dim)[:, 0, :]
    context = torch.cat([text_feat.float(), time_emb], dim=-1)
    feat = x if pc_emb is None else torch.cat([x, pc_emb], dim=-1)
    stride, nsample = (1, 4, 4, 4), (8, 16, 16, 16)
    o = torch.arange(1, B + 1, dtype=torch.int32) * N
    p0 = c_pc_xyz.reshape(B * N, 3).contiguous()
    x0 = torch.cat((p0, feat.reshape(B * N, -1)), 1)
    ps, xs, os_, knn = [], [], [], []
    for lvl in range(4):
        e = f"{cm}.enc{lvl + 1}"
        p0, x0, o, _ = sr.transition_down(sd, e + ".0", p0, x0, o, stride[lvl], nsample[lvl])
        ki, _ = sr.po.knn_query(nsample[lvl], p0, p0, o, o)
        for j in range(1, blocks[lvl]):
            x0 = sr.point_transformer_block(sd, f"{e}.{j}", p0, x0, o, nsample[lvl], 8, ki)
        ps.append(p0); xs.append(x0); os_.append(o); knn.append(ki)
    if v2:
        x4 = encoder_layer_relu(sd, cm + ".self_attn_layers.layers.0", xs[3].view(B, -1, xs[3].shape[-1]), 8).reshape(xs[3].shape)
        x4 = _ctx_mlp(sd, cm + ".ctx4", x4, context, B)
    else:
        x4 = _ctx_mlp(sd, cm + ".ctx", xs[3], context, B)
    y = sr.transition_up(sd, cm + ".dec4.0", ps[3], x4, os_[3])
    y = sr.point_transformer_block(sd, cm + ".dec4.1", ps[3], y, os_[3], nsample[3], 8, knn[3])
    for lvl in (2, 1, 0):
        xl = xs[lvl]
        if v2 and lvl in (2, 1):
            xl = _ctx_mlp(sd, f"{cm}.ctx{lvl + 1}", xl, context, B)
        y = sr.transition_up(sd, f"{cm}.dec{lvl + 1}.0", ps[lvl], xl, os_[lvl], ps[lvl + 1], y, os_[lvl + 1])
        y = sr.point_transformer_block(sd, f"{cm}.dec{lvl + 1}.1", ps[lvl], y, os_[lvl], nsample[lvl], 8, knn[lvl])
    return _lin(sd, "contact_layer", y.view(B, N, -1))


def _attn(q, k, v, nhead, key_mask=None):
    B, Tq, d = q.shape
    sp = lambda z: z.view(B, z.shape[1], nhead, d // nhead).transpose(1, 2)
    s = sp(q) @ sp(k).transpose(-1, -2) / math.sqrt(d // nhead)
    if key_mask is not None:
        s = s.masked_fill(key_mask[:, None, None, :], float("-inf"))
    return (torch.softmax(s, dim=-1) @ sp(v)).transpose(1, 2).reshape(B, Tq, d)


def decoder_layer(sd: SD, pre: str, x, mem, tgt_mask, mem_mask, nhead: int):
    """nn.TransformerDecoderLayer(batch_first, post-LN, exact-erf GELU): self-attention, cross-attention over `mem`, FFN."""
    d = x.shape[-1]
    qkv = F.linear(x, sd[pre + ".self_attn.in_proj_weight"], sd[pre + ".self_attn.in_proj_bias"])
    q, k, v = qkv.split(d, dim=-1)
    x = _ln(sd, pre + ".norm1", x + _lin(sd, pre + ".self_attn.out_proj", _attn(q, k, v, nhead, tgt_mask)))
    W, bq = sd[pre + ".multihead_attn.in_proj_weight"], sd[pre + ".multihead_attn.in_proj_bias"]
    q = F.linear(x, W[:d], bq[:d])
    k, v = F.linear(mem, W[d:2 * d], bq[d:2 * d]), F.linear(mem, W[2 * d:], bq[2 * d:])
    x = _ln(sd, pre + ".norm2", x + _lin(sd, pre + ".multihead_attn.out_proj", _attn(q, k, v, nhead, mem_mask)))
    return _ln(sd, pre + ".norm3", x + _lin(sd, pre + ".linear2", F.gelu(_lin(sd, pre + ".linear1", x))))


def cmdm_trans_dec_forward(sd: SD, x, t, text_feat, c_pc_xyz, c_pc_contact, x_mask=None, *, time_emb_dim: int = 512, nhead: int = 8,
                           num_layers=(1, 1, 1, 1, 1), blocks=(2, 2, 2, 2)):
    """CMDM.forward, `trans_dec` branch (cmdm.py:118-133,171-195)."""
    B, L, _ = x.shape
    d = sd["motion_adapter.weight"].shape[0]
    time_emb = timestep_embed(sd, "timestep_embedder", t, time_emb_dim)
    text_emb = _lin(sd, "language_adapter", text_feat.unsqueeze(1).float())
    cont = scene_ref.scene_map_encoder_decoder(sd, "contact_encoder", c_pc_xyz, c_pc_contact, blocks=blocks)
    seq = torch.cat([time_emb, text_emb, _lin(sd, "motion_adapter", x)], dim=1)
    T = seq.shape[1]
    seq = seq + sinusoid_table(5000, d)[:T].unsqueeze(0)
    if x_mask is None:
        x_mask = torch.zeros(B, L, dtype=torch.bool)
    mask = torch.cat([torch.zeros(B, 2, dtype=torch.bool), x_mask], dim=1)
    for i, n in enumerate(num_layers):
        for j in range(n):
            seq = encoder_layer(sd, f"self_attn_layers.{i}.layers.{j}", seq, mask, nhead)
        if i != len(num_layers) - 1:
            mem = _ln(sd, f"kv_mappling_layers.{i}.1", _lin(sd, f"kv_mappling_layers.{i}.0", cont[i]))
            seq = decoder_layer(sd, f"cross_attn_layers.{i}", seq, mem, mask, None, nhead)
    return _lin(sd, "motion_layer", seq[:, 2:, :])
