"""Host side of the SEED-2 image tokenizer on MI355X: weight packing + one C-ABI call per batch.

Mirrors ``Blip2QformerQuantizer`` (models/seed_qformer/qformer_quantizer.py:161-307) for the encode
path only.  ``load_state_dict``-compatible: it consumes the reference's key names
(``visual_encoder.blocks.N.attn.qkv.weight`` ...) and repacks them once:

* every tensor -> the engine's 16-bit dtype (bf16 by default, fp16 on request), contiguous, device resident (2.18 GB at full size);
* ``patch_embed.proj.weight`` [D,3,14,14] -> [D, 640] (K = 588 zero padded to the GEMM's K-tile);
* ViT qkv bias = cat(q_bias, 0, v_bias) built once instead of per call (eva_vit.py:133);
* Q-Former self-attention q/k/v stacked to one [3Q,Q] GEMM, cross-attention k/v to one [2Q,D] GEMM;
* input-independent work folded at load: ``cls + pos_embed[0]`` and ``LayerNorm(query_tokens)``
  (qformer_causual.py:94-98) and the codebook row norms (qformer_quantizer.py:95).

PyTorch is used as the allocator / stream owner only; all arithmetic on the measured path happens inside
``seedmi_tokenize``.
"""
import ctypes as C
from typing import Dict, Optional

import torch

from . import lib as L
from .config import TokenizerConfig


def _round_up(x, m):
    return (x + m - 1) // m * m


class TokenizerEngine:
    def __init__(self, state_dict: Dict[str, torch.Tensor], cfg: TokenizerConfig, device="cuda", fold_layernorm: bool = True,
                 dtype: torch.dtype = torch.bfloat16):
        # dtype: the 16-bit element the whole path computes in - torch.bfloat16 (BASELINE.json's configs; libseedmi.so) or torch.float16
        # (the reference's shipped `fp16: True`, configs/tokenizer/seed_llama_tokenizer_hf.yaml:3; libseedmi_f16.so: same kernels, same
        # rounding places, fp16 as the element)
        if dtype not in (torch.bfloat16, torch.float16):
            raise L.SeedmiError(f"TokenizerEngine computes in bfloat16 or float16, not {dtype}")
        self.dtype = dtype
        self.lib = L.load(dtype)
        self.cfg = cfg
        # fold_layernorm: norm1 / norm2 of every ViT block are applied inside the qkv / fc1 GEMMs (seedmi_gemm_bf16_ext) instead of
        # as separate passes over the token stream; False keeps the explicit LayerNorm launches (A/B, tests)
        self.fold_layernorm = bool(fold_layernorm)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise L.SeedmiError("TokenizerEngine needs a HIP device (cuda:N); there is no CPU path")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        with torch.cuda.device(self.device):
            L.check(self.lib.seedmi_check_device(), "seedmi_check_device", self.lib)
        self._keep = []          # tensors owning the device memory referenced by the C structs
        self._ws = None
        self._ws_batch = 0
        self.max_batch = max(1, min(1024, (2 ** 31 - 1) // (cfg.n_tokens * max(cfg.vit_ffn, 3 * cfg.vit_dim)) - 1))
        self._pack(state_dict)

    # ------------------------------------------------------------------ packing
    def _dev(self, t: torch.Tensor) -> torch.Tensor:
        t = t.detach().to(device=self.device, dtype=self.dtype).contiguous()
        self._keep.append(t)
        return t

    def _dev32(self, t: torch.Tensor) -> torch.Tensor:
        t = t.detach().to(device=self.device, dtype=torch.float32).contiguous()
        self._keep.append(t)
        return t

    def _fold_ln(self, weight, bias, gamma, beta):
        """LayerNorm folded into the Linear behind it (eva_vit.py:199-202 + :135 / :60):
        LN(x) W^T + b = rstd * (x (W * gamma)^T - mean * colsum) + (b + W beta).  The GEMM multiplies the half residual stream by
        W' = half(W * gamma); colsum is taken over the ROUNDED W' so that the mean term cancels exactly what the MFMAs summed.
        Everything on the parameters' device in fp32 (the weights are the model's half parameters)."""
        w32 = weight.detach().to(self.device).to(self.dtype).float()
        g32 = gamma.detach().to(self.device).to(self.dtype).float()
        b32 = beta.detach().to(self.device).to(self.dtype).float()
        wg = (w32 * g32.unsqueeze(0)).to(self.dtype)
        colsum = wg.float().sum(dim=1)
        bias32 = bias.detach().to(self.device).to(self.dtype).float() + w32 @ b32
        self._keep += [wg]
        return wg.contiguous(), self._dev32(colsum), self._dev32(bias32)

    def _pack(self, sd):
        cfg = self.cfg
        D, Q = cfg.vit_dim, cfg.qf_dim
        kpad = _round_up(cfg.patch_k, 64)
        g = lambda k: sd[k]  # noqa: E731
        p = L.ptr

        w = L.TokenizerWeights()
        for name in ("img_size", "patch", "vit_dim", "vit_depth", "vit_heads", "qf_dim", "qf_layers", "qf_heads",
                     "qf_ffn", "n_query", "n_embed", "code_dim"):
            setattr(w, name, getattr(cfg, name))
        w.vit_ffn = cfg.vit_ffn
        w.kpad = kpad

        pw = torch.zeros(D, kpad, dtype=torch.float32)
        pw[:, :cfg.patch_k] = g("visual_encoder.patch_embed.proj.weight").float().reshape(D, -1).cpu()
        w.patch_w = p(self._dev(pw))
        w.patch_b = p(self._dev(g("visual_encoder.patch_embed.proj.bias")))
        pos = g("visual_encoder.pos_embed").reshape(cfg.n_tokens, D)
        w.pos_embed = p(self._dev(pos))
        cls = g("visual_encoder.cls_token").reshape(D)
        # half(cls) + half(pos[0]) rounded once more: what `x + self.pos_embed` does on the cls row (eva_vit.py:373-376)
        cls_pos0 = (cls.to(self.dtype).float() + pos[0].to(self.dtype).float())
        w.cls_pos0 = p(self._dev(cls_pos0))

        vit = (L.VitLayer * cfg.vit_depth)()
        for i in range(cfg.vit_depth):
            pre = f"visual_encoder.blocks.{i}."
            l = vit[i]
            l.ln1_w, l.ln1_b = p(self._dev(g(pre + "norm1.weight"))), p(self._dev(g(pre + "norm1.bias")))
            l.qkv_w = p(self._dev(g(pre + "attn.qkv.weight")))
            qb, vb = g(pre + "attn.q_bias"), g(pre + "attn.v_bias")
            l.qkv_b = p(self._dev(torch.cat((qb, torch.zeros_like(vb), vb))))
            l.proj_w, l.proj_b = p(self._dev(g(pre + "attn.proj.weight"))), p(self._dev(g(pre + "attn.proj.bias")))
            l.ln2_w, l.ln2_b = p(self._dev(g(pre + "norm2.weight"))), p(self._dev(g(pre + "norm2.bias")))
            l.fc1_w, l.fc1_b = p(self._dev(g(pre + "mlp.fc1.weight"))), p(self._dev(g(pre + "mlp.fc1.bias")))
            l.fc2_w, l.fc2_b = p(self._dev(g(pre + "mlp.fc2.weight"))), p(self._dev(g(pre + "mlp.fc2.bias")))
            if self.fold_layernorm:
                wg, cs, bf = self._fold_ln(g(pre + "attn.qkv.weight"), torch.cat((qb, torch.zeros_like(vb), vb)),
                                           g(pre + "norm1.weight"), g(pre + "norm1.bias"))
                l.qkv_wg, l.qkv_cs, l.qkv_bf = p(wg), p(cs), p(bf)
                wg, cs, bf = self._fold_ln(g(pre + "mlp.fc1.weight"), g(pre + "mlp.fc1.bias"), g(pre + "norm2.weight"),
                                           g(pre + "norm2.bias"))
                l.fc1_wg, l.fc1_cs, l.fc1_bf = p(wg), p(cs), p(bf)
        self._vit = vit
        w.vit = C.cast(vit, C.POINTER(L.VitLayer))
        w.ln_vision_w, w.ln_vision_b = p(self._dev(g("ln_vision.weight"))), p(self._dev(g("ln_vision.bias")))

        qf = (L.QfLayer * cfg.qf_layers)()
        for i in range(cfg.qf_layers):
            pre = f"Qformer.bert.encoder.layer.{i}."
            l = qf[i]
            a = pre + "attention.self."
            l.qkv_w = p(self._dev(torch.cat([g(a + f"{n}.weight") for n in ("query", "key", "value")], 0)))
            l.qkv_b = p(self._dev(torch.cat([g(a + f"{n}.bias") for n in ("query", "key", "value")], 0)))
            o = pre + "attention.output."
            l.ao_w, l.ao_b = p(self._dev(g(o + "dense.weight"))), p(self._dev(g(o + "dense.bias")))
            l.ao_ln_w, l.ao_ln_b = p(self._dev(g(o + "LayerNorm.weight"))), p(self._dev(g(o + "LayerNorm.bias")))
            l.has_cross = 1 if (pre + "crossattention.self.query.weight") in sd else 0
            if l.has_cross:
                c = pre + "crossattention.self."
                l.cq_w, l.cq_b = p(self._dev(g(c + "query.weight"))), p(self._dev(g(c + "query.bias")))
                l.ckv_w = p(self._dev(torch.cat([g(c + "key.weight"), g(c + "value.weight")], 0)))
                l.ckv_b = p(self._dev(torch.cat([g(c + "key.bias"), g(c + "value.bias")], 0)))
                co = pre + "crossattention.output."
                l.co_w, l.co_b = p(self._dev(g(co + "dense.weight"))), p(self._dev(g(co + "dense.bias")))
                l.co_ln_w, l.co_ln_b = p(self._dev(g(co + "LayerNorm.weight"))), p(self._dev(g(co + "LayerNorm.bias")))
            l.ffn_w1 = p(self._dev(g(pre + "intermediate_query.dense.weight")))
            l.ffn_b1 = p(self._dev(g(pre + "intermediate_query.dense.bias")))
            oq = pre + "output_query."
            l.ffn_w2, l.ffn_b2 = p(self._dev(g(oq + "dense.weight"))), p(self._dev(g(oq + "dense.bias")))
            l.ffn_ln_w, l.ffn_ln_b = p(self._dev(g(oq + "LayerNorm.weight"))), p(self._dev(g(oq + "LayerNorm.bias")))
        self._qf = qf
        w.qf = C.cast(qf, C.POINTER(L.QfLayer))

        # LayerNorm(query_tokens) is input independent -> computed once by the HIP LayerNorm kernel
        qt = self._dev(g("query_tokens").reshape(cfg.n_query, Q))
        q_ln = torch.empty_like(qt)
        self._keep.append(q_ln)
        e = "Qformer.bert.embeddings.LayerNorm."
        ew, eb = self._dev(g(e + "weight")), self._dev(g(e + "bias"))
        with torch.cuda.device(self.device):
            L.check(self.lib.seedmi_layernorm_bf16(L.ptr(qt), Q, L.ptr(ew), L.ptr(eb), 1e-12, L.ptr(q_ln), Q,
                                                   cfg.n_query, Q, L.stream_ptr()), "layernorm(query_tokens)", self.lib)
        w.query_ln = p(q_ln)

        w.head_w0, w.head_b0 = p(self._dev(g("encode_task_layer.0.weight"))), p(self._dev(g("encode_task_layer.0.bias")))
        w.head_w1, w.head_b1 = p(self._dev(g("encode_task_layer.2.weight"))), p(self._dev(g("encode_task_layer.2.bias")))
        self.w = w
        self.set_codebook(g("quantize.embedding.weight"))

    def set_codebook(self, codebook: torch.Tensor):
        cfg = self.cfg
        assert tuple(codebook.shape) == (cfg.n_embed, cfg.code_dim)
        self.codebook = self._dev(codebook)
        self.code_sqnorm = torch.empty(cfg.n_embed, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            L.check(self.lib.seedmi_vq_code_sqnorm(L.ptr(self.codebook), L.ptr(self.code_sqnorm), cfg.n_embed,
                                                   cfg.code_dim, L.stream_ptr()), "seedmi_vq_code_sqnorm", self.lib)
        self.w.codebook = L.ptr(self.codebook)
        self.w.code_sqnorm = L.ptr(self.code_sqnorm)

    # ------------------------------------------------------------------ run
    def _workspace(self, batch: int) -> torch.Tensor:
        if self._ws is None or batch > self._ws_batch:
            nbytes = self.lib.seedmi_tokenize_workspace_bytes(C.byref(self.w), batch)
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self._ws_batch = batch
        return self._ws

    def encode(self, images: torch.Tensor, taps: Optional[dict] = None) -> torch.Tensor:
        """images [B,3,S,S] float (fp32 or the engine's 16-bit dtype; other float types are widened to fp32) on this engine's device.
        Returns int64 [B, n_query] on the same device (stream ordered, no host sync)."""
        cfg = self.cfg
        if images.dim() == 3:                       # seed_llama_tokenizer.py:81-82
            images = images.unsqueeze(0)
        if images.dim() != 4 or images.shape[1] != 3 or images.shape[2] != cfg.img_size or images.shape[3] != cfg.img_size:
            # eva_vit.py:227-228
            raise AssertionError(f"Input image size ({tuple(images.shape)}) doesn't match model "
                                 f"(3x{cfg.img_size}x{cfg.img_size}).")
        if images.device != self.device:
            raise L.SeedmiError(f"images on {images.device}, engine on {self.device} (the caller places the tensor, "
                                "seed_llama_tokenizer.py:84-85)")
        if images.dtype not in (torch.float32, self.dtype):
            images = images.float()
        images = images.contiguous()
        B = images.shape[0]
        if B == 0:
            # an empty batch is a valid input of the reference (every module maps over dim 0: get_codebook_indices returns [0, 32]) and what a
            # rank with an empty shard passes (dist.tokenize_data_parallel with fewer images than ranks): no launch, no workspace
            return torch.empty(0, cfg.n_query, dtype=torch.int64, device=self.device)
        if B > self.max_batch and taps is None:
            # one C call addresses its matrices with 32-bit element offsets (B * n_tokens * ffn < 2^31): larger batches are a
            # sequence of calls on the same stream (the batch is a pure map over images)
            return torch.cat([self.encode(images[i:i + self.max_batch]) for i in range(0, B, self.max_batch)], dim=0)
        ids = torch.empty(B, cfg.n_query, dtype=torch.int64, device=self.device)
        ws = self._workspace(B)
        tp = None
        if taps is not None:
            t = L.TokenizerTaps()
            emb = torch.empty(B * cfg.n_tokens, cfg.vit_dim, dtype=self.dtype, device=self.device)
            qo = torch.empty(B * cfg.n_query, cfg.qf_dim, dtype=self.dtype, device=self.device)
            z = torch.empty(B * cfg.n_query, cfg.code_dim, dtype=self.dtype, device=self.device)
            t.image_embeds, t.qformer_out, t.z = L.ptr(emb), L.ptr(qo), L.ptr(z)
            taps.update(image_embeds=emb.view(B, cfg.n_tokens, -1), qformer_out=qo.view(B, cfg.n_query, -1),
                        z=z.view(B, cfg.n_query, -1))
            tp = C.byref(t)
        with torch.cuda.device(self.device):
            rc = self.lib.seedmi_tokenize(C.byref(self.w), L.ptr(images), 1 if images.dtype == torch.float32 else 0, B,
                                          L.ptr(ids), tp, L.ptr(ws), ws.numel(), L.stream_ptr())
        L.check(rc, "seedmi_tokenize", self.lib)
        return ids
