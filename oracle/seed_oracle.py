"""CPU restatement of the reference's tokenize-and-generate hot path (the parity ORACLE).

TEST INFRASTRUCTURE.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module, and only as the checker / the timed CPU baseline.
The product path (``seed_amd/``, ``models/``) never imports it and fails loudly if the HIP
library is missing.

Pinning status: the reference ships NO tests, golden vectors or fixtures for this path
(SURVEY.md section 4 / 8c), so the pin is created here: ``oracle/make_golden.py`` runs the reference's
*own modules* (imported from /root/reference through ``oracle/ref_shims.py``) on seeded
synthetic weights and commits their outputs under ``tests/golden/``;
``tests/test_oracle_golden.py`` checks this restatement against those vectors.
Third-party arithmetic that is not under /root/reference (xformers >= 0.0.20 memory-efficient
attention, transformers 4.30.2 ACT2FN/GenerationMixin, torch GEMM/LN/softmax) is restated from
its published semantics and anchored on the reference call sites cited below.

Every function cites the reference lines it follows.  Two precisions:

* ``mode='fp32'``  — every op in fp32: the ground truth.
* ``mode='bf16'``  — the reference's GPU dtype choreography with bf16 as the half type
  (SURVEY.md appendix A): ViT under autocast (GEMM/conv/matmul outputs rounded to bf16 after fp32
  accumulation, LayerNorm and softmax computed in fp32 and re-rounded by the consuming GEMM,
  bf16 residual stream), ``ln_vision`` explicit fp32, Q-Former / task MLP / VQ entirely in bf16
  (each op's output rounded to bf16).  Tensors are carried as fp32 holding bf16-representable
  values; ``r()`` marks every point where the reference materialises a half tensor.
"""
import math
from typing import Dict, Optional, Tuple, List

import numpy as np
import torch
import torch.nn.functional as F


class Prec:
    """fp32 = ground truth; "bf16" / "fp16" = the same choreography (where the reference materialises a half tensor)
    with that element type - bf16 is what BASELINE.json's configs name, fp16 what the reference ships (seed_llama_tokenizer_hf.yaml:3)."""

    def __init__(self, mode: str):
        assert mode in ("fp32", "bf16", "fp16")
        self.mode = mode
        self.half = mode != "fp32"
        self.dtype = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[mode]

    def r(self, x: torch.Tensor) -> torch.Tensor:
        """Round to the model's half type (identity in fp32 mode)."""
        return x.to(self.dtype).float() if self.half else x


def _w(sd, name, prec: Prec):
    """Parameters live in the model dtype (``model.half()``, seed_llama_tokenizer.py:58-59)."""
    return prec.r(sd[name].float())


def linear(x, w, b, prec: Prec):
    """nn.Linear / F.linear: fp32 accumulate, bias added before the single output rounding."""
    y = x @ w.t()
    if b is not None:
        y = y + b
    return prec.r(y)


def layer_norm(x, w, b, eps):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def gelu_erf(x):
    """nn.GELU() (eva_vit.py:50) and ACT2FN['gelu'] (qformer_causual.py:316): exact erf form."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


# ----------------------------------------------------------------------------- EVA-ViT-g


def patch_embed(sd, image, cfg, prec: Prec):
    """PatchEmbed.forward (eva_vit.py:224-230): Conv2d(3->D, k=patch, s=patch)+bias as a GEMM over
    (c, kh, kw)-ordered patches, then cls/pos (forward_features, eva_vit.py:369-377)."""
    B = image.shape[0]
    p, g, D = cfg.patch, cfg.grid, cfg.vit_dim
    x = prec.r(image.float())
    # [B,3,g,p,g,p] -> [B, g*g, 3*p*p]
    cols = x.reshape(B, 3, g, p, g, p).permute(0, 2, 4, 1, 3, 5).reshape(B, g * g, 3 * p * p)
    w = _w(sd, "visual_encoder.patch_embed.proj.weight", prec).reshape(D, -1)
    b = _w(sd, "visual_encoder.patch_embed.proj.bias", prec)
    tok = linear(cols, w, b, prec)                                     # conv output (half)
    cls = _w(sd, "visual_encoder.cls_token", prec).expand(B, -1, -1)
    x = torch.cat((cls, tok), dim=1)                                   # :373-374
    x = prec.r(x + _w(sd, "visual_encoder.pos_embed", prec))           # :375-376
    return x


def vit_attention(sd, p, x_ln, cfg, prec: Prec):
    """Attention.forward (eva_vit.py:129-159)."""
    B, N, C = x_ln.shape
    H, hd = cfg.vit_heads, cfg.vit_head_dim
    qb = _w(sd, p + "attn.q_bias", prec)
    vb = _w(sd, p + "attn.v_bias", prec)
    qkv_bias = torch.cat((qb, torch.zeros_like(vb), vb))               # :133
    qkv = linear(x_ln, _w(sd, p + "attn.qkv.weight", prec), qkv_bias, prec)   # :135
    qkv = qkv.reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)           # :136
    q, k, v = qkv[0], qkv[1], qkv[2]
    q = prec.r(q * (hd ** -0.5))                                       # :139  (half * python float)
    attn = prec.r(q @ k.transpose(-2, -1))                             # :140  matmul -> half
    attn = attn.softmax(dim=-1)                                        # :153  autocast-fp32 op
    attn = prec.r(attn)                                                # re-cast by the next matmul
    x = prec.r(attn @ v).transpose(1, 2).reshape(B, N, C)              # :156
    return linear(x, _w(sd, p + "attn.proj.weight", prec), _w(sd, p + "attn.proj.bias", prec), prec)  # :157


def vit_block(sd, i, x, cfg, prec: Prec):
    """Block.forward with gamma_1 None (eva_vit.py:199-202); LayerNorm eps 1e-6 (:472) in fp32."""
    p = f"visual_encoder.blocks.{i}."
    h = prec.r(layer_norm(x, _w(sd, p + "norm1.weight", prec), _w(sd, p + "norm1.bias", prec), 1e-6))
    x = prec.r(x + vit_attention(sd, p, h, cfg, prec))
    h = prec.r(layer_norm(x, _w(sd, p + "norm2.weight", prec), _w(sd, p + "norm2.bias", prec), 1e-6))
    h = linear(h, _w(sd, p + "mlp.fc1.weight", prec), _w(sd, p + "mlp.fc1.bias", prec), prec)   # Mlp :60
    h = prec.r(gelu_erf(h))                                                                        # :61
    h = linear(h, _w(sd, p + "mlp.fc2.weight", prec), _w(sd, p + "mlp.fc2.bias", prec), prec)   # :64
    return prec.r(x + h)


def vit_forward(sd, image, cfg, prec: Prec, taps: Optional[dict] = None):
    x = patch_embed(sd, image, cfg, prec)
    if taps is not None:
        taps["vit_embed"] = x.clone()
    for i in range(cfg.vit_depth):
        x = vit_block(sd, i, x, cfg, prec)
        if taps is not None and i == 0:
            taps["vit_block0"] = x.clone()
    return x


# ----------------------------------------------------------------------------- Q-Former


def qf_attention(q_in, kv_in, sd, p, cfg, prec: Prec, causal: bool):
    """BertSelfAttention.forward (qformer_causual.py:148-241)."""
    B, nq, Q = q_in.shape
    H = cfg.qf_heads
    hd = Q // H

    def heads(t):  # transpose_for_scores :140-146
        return t.reshape(B, -1, H, hd).permute(0, 2, 1, 3)

    k = heads(linear(kv_in, _w(sd, p + "key.weight", prec), _w(sd, p + "key.bias", prec), prec))
    v = heads(linear(kv_in, _w(sd, p + "value.weight", prec), _w(sd, p + "value.bias", prec), prec))
    q = heads(linear(q_in, _w(sd, p + "query.weight", prec), _w(sd, p + "query.bias", prec), prec))
    s = prec.r(q @ k.transpose(-1, -2))                                # :189
    s = prec.r(s / math.sqrt(hd))                                      # :212
    if causal:
        # get_extended_attention_mask (:712-714, 765-766): (1 - tril) * -10000 in the model dtype
        ids = torch.arange(nq)
        allow = (ids[None, :] <= ids[:, None]).float()
        mask = prec.r((1.0 - allow) * prec.r(torch.tensor(-10000.0)))
        s = prec.r(s + mask)                                           # :215
    # cross-attention mask: invert_attention_mask(ones) == 0 everywhere (:883) -> adding it is a no-op
    pr = prec.r(torch.softmax(s, dim=-1))                              # :218 (half softmax, fp32 inside)
    ctx = prec.r(pr @ v)                                               # :232
    return ctx.permute(0, 2, 1, 3).reshape(B, nq, Q)


def qf_self_output(x, res, sd, p, prec: Prec):
    """BertSelfOutput / BertOutput.forward (qformer_causual.py:251-255, 333-337), LN eps 1e-12."""
    h = linear(x, _w(sd, p + "dense.weight", prec), _w(sd, p + "dense.bias", prec), prec)
    h = prec.r(h + res)
    return prec.r(layer_norm(h, _w(sd, p + "LayerNorm.weight", prec), _w(sd, p + "LayerNorm.bias", prec), 1e-12))


def qformer_forward(sd, image_embeds, cfg, prec: Prec, taps: Optional[dict] = None):
    """BertModel.forward with query_embeds only (qformer_causual.py:769-931) -> BertEncoder (:453-543)
    -> BertLayer.forward (:359-434) with query_length == 32."""
    B = image_embeds.shape[0]
    qt = _w(sd, "query_tokens", prec).expand(B, -1, -1)                # qformer_quantizer.py:293
    pe = "Qformer.bert.embeddings.LayerNorm."
    x = prec.r(layer_norm(qt, _w(sd, pe + "weight", prec), _w(sd, pe + "bias", prec), 1e-12))   # :94-98
    for i in range(cfg.qf_layers):
        p = f"Qformer.bert.encoder.layer.{i}."
        ctx = qf_attention(x, x, sd, p + "attention.self.", cfg, prec, causal=True)
        x = qf_self_output(ctx, x, sd, p + "attention.output.", prec)
        if i % cfg.cross_freq == 0:                                    # :348-350, 395-406
            ctx = qf_attention(x, image_embeds, sd, p + "crossattention.self.", cfg, prec, causal=False)
            x = qf_self_output(ctx, x, sd, p + "crossattention.output.", prec)
        # feed_forward_chunk_query (:441-444)
        h = linear(x, _w(sd, p + "intermediate_query.dense.weight", prec),
                   _w(sd, p + "intermediate_query.dense.bias", prec), prec)
        h = prec.r(gelu_erf(h))
        x = qf_self_output(h, x, sd, p + "output_query.", prec)
        if taps is not None and i == 0:
            taps["qf_layer0"] = x.clone()
    return x


# ----------------------------------------------------------------------------- task head + VQ


def encode_task_layer(sd, x, prec: Prec):
    """nn.Sequential(Linear, Tanh, Linear) (qformer_quantizer.py:219-223, 301)."""
    h = linear(x, _w(sd, "encode_task_layer.0.weight", prec), _w(sd, "encode_task_layer.0.bias", prec), prec)
    h = prec.r(torch.tanh(h))
    return linear(h, _w(sd, "encode_task_layer.2.weight", prec), _w(sd, "encode_task_layer.2.bias", prec), prec)


def _bf16_round_np(x: np.ndarray) -> np.ndarray:
    """fp32 -> bf16 (round-to-nearest-even) -> fp32, bit-exact with torch / gfx950 v_cvt_pk_bf16_f32."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    rounded = (u + (np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1)))) & np.uint32(0xFFFF0000)
    return rounded.view(np.float32)


def vq_distances_fixed_order(z: np.ndarray, e: np.ndarray, half: bool) -> np.ndarray:
    """VectorQuantizer2.forward distance (qformer_quantizer.py:94-96) with a FIXED fp32 summation order:

        d = (sum(z**2, 1, keepdim) + sum(e**2, 1)) - 2 * (z @ e.T)

    In the model dtype every term is materialised as a half tensor: z**2 and e**2 elementwise, the two
    row sums (fp32 accumulate, one rounding), their broadcast sum, the einsum output, 2*einsum (exact)
    and the final subtraction.  The only unspecified thing in the reference is the order of the fp32
    accumulation inside torch's reductions/BLAS; the oracle (and the HIP kernel, and oracle/vq_oracle.c)
    fix it to a sequential k = 0..D-1 chain of separately rounded fp32 multiply and add (no FMA
    contraction: products of two bf16 values are exact in fp32, so mul+add == fma bit for bit in half
    mode).
    """
    z = np.ascontiguousarray(z, dtype=np.float32)
    e = np.ascontiguousarray(e, dtype=np.float32)
    # half: False (fp32) | True or "bf16" | "fp16" (numpy's float16 conversion rounds to nearest even, subnormals kept: what
    # torch's .half() and v_cvt_f16_f32 do)
    if half == "fp16":
        rnd = lambda a: np.asarray(a, dtype=np.float32).astype(np.float16).astype(np.float32)      # noqa: E731
    else:
        rnd = _bf16_round_np if half else (lambda a: a)
    D = z.shape[1]
    zz = np.zeros(z.shape[0], np.float32)
    ee = np.zeros(e.shape[0], np.float32)
    for k in range(D):
        zz = (zz + rnd(z[:, k] * z[:, k])).astype(np.float32)
        ee = (ee + rnd(e[:, k] * e[:, k])).astype(np.float32)
    zz, ee = rnd(zz), rnd(ee)
    dot = np.zeros((z.shape[0], e.shape[0]), np.float32)
    for k in range(D):
        dot = (dot + (z[:, k:k + 1] * e[None, :, k]).astype(np.float32)).astype(np.float32)
    dot = rnd(dot)
    s = rnd((zz[:, None] + ee[None, :]).astype(np.float32))
    return rnd((s - (np.float32(2.0) * dot)).astype(np.float32))


def vq_argmin(z: torch.Tensor, codebook: torch.Tensor, prec: Prec, return_gap: bool = False):
    """argmin over the codebook with torch.argmin's first-index tie-break (qformer_quantizer.py:98)."""
    zf = prec.r(z.float()).reshape(-1, z.shape[-1]).numpy()
    ef = prec.r(codebook.float()).numpy()
    ids = np.empty(zf.shape[0], np.int64)
    gaps = np.empty(zf.shape[0], np.float32)
    step = 1024
    for s0 in range(0, zf.shape[0], step):
        d = vq_distances_fixed_order(zf[s0:s0 + step], ef, prec.mode if prec.half else False)
        ids[s0:s0 + step] = np.argmin(d, axis=1)                      # first minimal index
        if return_gap:
            part = np.partition(d, 1, axis=1)
            gaps[s0:s0 + step] = part[:, 1] - part[:, 0]
    ids_t = torch.from_numpy(ids)
    return (ids_t, torch.from_numpy(gaps)) if return_gap else ids_t


def get_codebook_indices(sd: Dict[str, torch.Tensor], image: torch.Tensor, cfg, mode: str = "fp32",
                         taps: Optional[dict] = None) -> torch.Tensor:
    """Blip2QformerQuantizer.get_codebook_indices (qformer_quantizer.py:288-307) followed by
    ImageTokenizer.encode's ``id.view(B, -1)`` (seed_llama_tokenizer.py:75-90). Returns int64 [B, 32]."""
    prec = Prec(mode)
    if image.dim() == 3:                                               # seed_llama_tokenizer.py:81-82
        image = image.unsqueeze(0)
    with torch.no_grad():
        x = vit_forward(sd, image, cfg, prec, taps)
        # ln_vision: fp32 LayerNorm eps 1e-5, cast back (blip2.py:179-184)
        emb = prec.r(layer_norm(x, _w(sd, "ln_vision.weight", prec), _w(sd, "ln_vision.bias", prec), 1e-5))
        qo = qformer_forward(sd, emb, cfg, prec, taps)
        z = encode_task_layer(sd, qo, prec)
        if taps is not None:
            taps.update(image_embeds=emb, qformer_out=qo, z=z)
        ids, gap = vq_argmin(z, sd["quantize.embedding.weight"], prec, return_gap=True)
        if taps is not None:
            taps["vq_gap"] = gap.reshape(image.shape[0], -1)
    return ids.reshape(image.shape[0], -1)


# ----------------------------------------------------------------------------- de-tokenizer front half


def detok_block(sd, i, x, cfg, prec: Prec):
    """vit.Block.forward (models/seed_qformer/vit.py:147-150) with Attention.forward (:85-105) and Mlp (:41-47); the module
    is .half()'ed and runs outside autocast (seed_llama_tokenizer.py:62-63, 92-93), so LayerNorm / softmax / GELU outputs
    are half tensors too."""
    p = f"blocks_image.{i}."
    B, N, C = x.shape
    H = cfg.dec_heads
    hd = C // H
    h = prec.r(layer_norm(x, _w(sd, p + "norm1.weight", prec), _w(sd, p + "norm1.bias", prec), 1e-6))
    qkv = linear(h, _w(sd, p + "attn.qkv.weight", prec), _w(sd, p + "attn.qkv.bias", prec), prec)      # :87
    qkv = qkv.reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = prec.r(prec.r(q @ k.transpose(-2, -1)) * (hd ** -0.5))                                         # :94
    attn = prec.r(attn.softmax(dim=-1))                                                                  # :95
    a = prec.r(attn @ v).transpose(1, 2).reshape(B, N, C)                                                # :102
    a = linear(a, _w(sd, p + "attn.proj.weight", prec), _w(sd, p + "attn.proj.bias", prec), prec)      # :103
    x = prec.r(x + a)                                                                                    # :148
    h = prec.r(layer_norm(x, _w(sd, p + "norm2.weight", prec), _w(sd, p + "norm2.bias", prec), 1e-6))
    h = linear(h, _w(sd, p + "mlp.fc1.weight", prec), _w(sd, p + "mlp.fc1.bias", prec), prec)          # :42
    h = prec.r(gelu_erf(h))                                                                              # :43
    h = linear(h, _w(sd, p + "mlp.fc2.weight", prec), _w(sd, p + "mlp.fc2.bias", prec), prec)          # :45
    return prec.r(x + h)                                                                                 # :149


def get_codebook_entry(sd: Dict[str, torch.Tensor], ids: torch.Tensor, cfg, mode: str = "fp32", taps: Optional[dict] = None):
    """Blip2QformerQuantizer.get_codebook_entry, use_qformer_image=False (qformer_quantizer.py:309-338):
    ids int64 [B, n_query] -> image embeds [B, image_features_dim] (fp32 tensor holding half-representable values in
    bf16 mode)."""
    prec = Prec(mode)
    z_q = _w(sd, "quantize.embedding.weight", prec)[ids]                                                 # :310 / :133
    h = linear(z_q, _w(sd, "decode_task_layer.0.weight", prec), _w(sd, "decode_task_layer.0.bias", prec), prec)
    h = prec.r(torch.tanh(h))
    x = linear(h, _w(sd, "decode_task_layer.2.weight", prec), _w(sd, "decode_task_layer.2.bias", prec), prec)   # :314
    x = prec.r(x + _w(sd, "pos_embed_image", prec))                                                      # :316-317
    for i in range(cfg.decode_depth):                                                                    # :318-319
        x = detok_block(sd, i, x, cfg, prec)
    if taps is not None:
        taps["hidden"] = x.clone()
    d = prec.r(torch.relu(linear(x, _w(sd, "image_down.0.weight", prec), None, prec)))                   # :333
    d = prec.r(torch.relu(linear(d, _w(sd, "image_down.2.weight", prec), None, prec)))
    d = linear(d, _w(sd, "image_down.4.weight", prec), None, prec)
    d = d.reshape(d.shape[0], -1)                                                                        # :334
    return linear(d, _w(sd, "distill_image_proj.weight", prec), _w(sd, "distill_image_proj.bias", prec), prec)   # :335


# ----------------------------------------------------------------------------- LLaMA


def rms_norm(x, w, eps, prec: Prec):
    """LlamaRMSNorm.forward (llama_xformer.py:105-113): fp32 variance, x*rsqrt rounded to the weight
    dtype BEFORE the multiply by weight."""
    var = x.pow(2).mean(-1, keepdim=True)
    h = prec.r(x * torch.rsqrt(var + eps))
    return prec.r(w * h)


def rope_tables(cfg, prec: Prec):
    """LlamaRotaryEmbedding (llama_xformer.py:118-134, 147-150): fp32 cache cast to the activation dtype."""
    hd = cfg.head_dim
    inv_freq = 1.0 / (cfg.rope_base ** (torch.arange(0, hd, 2).float() / hd))
    t = torch.arange(cfg.max_pos, dtype=torch.float32)
    freqs = torch.einsum("i,j->ij", t, inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    return prec.r(emb.cos()), prec.r(emb.sin())


def apply_rope(x, cos, sin, prec: Prec):
    """rotate_half / apply_rotary_pos_emb (llama_xformer.py:153-168); x [B,H,T,hd], cos/sin [B,1,T,hd]."""
    h = x.shape[-1] // 2
    rot = torch.cat((-x[..., h:], x[..., :h]), dim=-1)
    return prec.r(prec.r(x * cos) + prec.r(rot * sin))


def llama_forward(sd, cfg, input_ids: Optional[torch.Tensor], past: Optional[List[Tuple[torch.Tensor, torch.Tensor]]] = None,
                  position_ids: Optional[torch.Tensor] = None, mode: str = "fp32", inputs_embeds: Optional[torch.Tensor] = None,
                  hidden_out: Optional[list] = None):
    """LlamaForCausalLM.forward in eval mode with use_cache=True on an UNPADDED equal-length batch
    (llama_xformer.py:661-743 -> LlamaModel.forward :496-627 -> LlamaDecoderLayer :280-332 ->
    LlamaAttention :212-263).  Attention = xformers.memory_efficient_attention semantics: scale
    1/sqrt(hd), fp32 softmax, causal (top-left aligned LowerTriangularMask) when q_len > 1, no bias when
    q_len == 1 (:251-256).  Returns (logits [B,T,V], past list of (k,v) [B,H,T,hd] post-RoPE).
    ``inputs_embeds`` replaces the embedding gather (:519-520, 543-544); ``hidden_out`` (a list) receives the
    output_hidden_states tuple: the input of every layer (:569-570) and the final-norm output (:613-617)."""
    prec = Prec(mode)
    B, T = input_ids.shape if input_ids is not None else inputs_embeds.shape[:2]
    H, hd = cfg.heads, cfg.head_dim
    past_len = 0 if past is None else past[0][0].shape[2]
    if position_ids is None:                                           # :531-539
        position_ids = torch.arange(past_len, past_len + T).unsqueeze(0).expand(B, -1)
    cos_t, sin_t = rope_tables(cfg, prec)
    cos = cos_t[position_ids].unsqueeze(1)                             # :164-165
    sin = sin_t[position_ids].unsqueeze(1)
    with torch.no_grad():
        if inputs_embeds is not None:
            x = prec.r(inputs_embeds.float())
        else:
            x = _w(sd, "model.embed_tokens.weight", prec)[input_ids]   # :544
        new_past = []
        for i in range(cfg.layers):
            p = f"model.layers.{i}."
            if hidden_out is not None:
                hidden_out.append(x)                                   # :569-570
            h = rms_norm(x, _w(sd, p + "input_layernorm.weight", prec), cfg.rms_eps, prec)
            q = linear(h, _w(sd, p + "self_attn.q_proj.weight", prec), None, prec)
            k = linear(h, _w(sd, p + "self_attn.k_proj.weight", prec), None, prec)
            v = linear(h, _w(sd, p + "self_attn.v_proj.weight", prec), None, prec)
            q = q.view(B, T, H, hd).transpose(1, 2)
            k = k.view(B, T, H, hd).transpose(1, 2)
            v = v.view(B, T, H, hd).transpose(1, 2)
            q = apply_rope(q, cos, sin, prec)
            k = apply_rope(k, cos, sin, prec)
            if past is not None:                                       # :234-237
                k = torch.cat([past[i][0], k], dim=2)
                v = torch.cat([past[i][1], v], dim=2)
            new_past.append((k, v))
            s = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(hd))      # fp32 scores inside the fused kernel
            if T > 1:
                Tk = k.shape[2]
                keep = torch.ones(T, Tk, dtype=torch.bool).tril()      # top-left aligned (xformers)
                s = s.masked_fill(~keep, float("-inf"))
            pr = prec.r(torch.softmax(s, dim=-1))
            o = prec.r(pr @ v).transpose(1, 2).reshape(B, T, H * hd)
            o = linear(o, _w(sd, p + "self_attn.o_proj.weight", prec), None, prec)
            x = prec.r(x + o)                                          # :316
            h = rms_norm(x, _w(sd, p + "post_attention_layernorm.weight", prec), cfg.rms_eps, prec)
            g = linear(h, _w(sd, p + "mlp.gate_proj.weight", prec), None, prec)
            u = linear(h, _w(sd, p + "mlp.up_proj.weight", prec), None, prec)
            a = prec.r(prec.r(F.silu(g)) * u)                          # :186
            d = linear(a, _w(sd, p + "mlp.down_proj.weight", prec), None, prec)
            x = prec.r(x + d)                                          # :322
        x = rms_norm(x, _w(sd, "model.norm.weight", prec), cfg.rms_eps, prec)     # :613
        if hidden_out is not None:
            hidden_out.append(x)                                                  # :615-617
        logits = linear(x, _w(sd, "lm_head.weight", prec), None, prec)            # :718 (model dtype)
    return logits, new_past


def llama_greedy_decode(sd, cfg, prompt_ids: torch.Tensor, n_new: int, mode: str = "fp32"):
    """Greedy loop over forward(input_ids=tok, past_key_values=pkv, use_cache=True) — the reference's
    HF generate() cannot run on its class under transformers 5.x, EOS ignored."""
    logits, past = llama_forward(sd, cfg, prompt_ids, mode=mode)
    step_logits = [logits[:, -1]]
    tok = logits[:, -1].argmax(-1, keepdim=True)
    out = [tok]
    for _ in range(n_new - 1):
        logits, past = llama_forward(sd, cfg, tok, past=past, mode=mode)
        step_logits.append(logits[:, -1])
        tok = logits[:, -1].argmax(-1, keepdim=True)
        out.append(tok)
    return torch.cat(out, dim=1), torch.stack(step_logits, dim=1)
