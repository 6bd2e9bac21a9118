"""Host side of the de-tokenizer front half: ids -> unCLIP image embeds through seedmi_detokenize (libseedmi.so).

Mirrors Blip2QformerQuantizer.get_codebook_entry for use_qformer_image=False
(models/seed_qformer/qformer_quantizer.py:309-338).  The state dict (reference key names) is repacked
once: the engine's 16-bit element (bf16 or fp16), codebook and the two 32-wide ``decode_task_layer`` Linears zero padded to the GEMM's K granularity of 64.
There is no CPU path: construction raises without a gfx950 device.
"""
import ctypes as C
from typing import Dict, Optional

import torch

from . import lib as L
from .config import TokenizerConfig

DETOK_KEYS = ("quantize.embedding.weight", "decode_task_layer.0.weight", "pos_embed_image", "image_down.0.weight",
              "distill_image_proj.weight")


def has_detokenizer_weights(sd: Dict[str, torch.Tensor]) -> bool:
    return all(k in sd for k in DETOK_KEYS)


class DetokenizerEngine:
    def __init__(self, state_dict: Dict[str, torch.Tensor], cfg: TokenizerConfig, device="cuda", dtype: torch.dtype = torch.bfloat16):
        # dtype: the 16-bit element the path computes in and returns - torch.bfloat16 (libseedmi.so) or torch.float16 (libseedmi_f16.so, the
        # reference's shipped setting: the de-tokenizer modules are .half()'ed, seed_llama_tokenizer.py:62-63)
        if dtype not in (torch.bfloat16, torch.float16):
            raise L.SeedmiError(f"DetokenizerEngine computes in bfloat16 or float16, not {dtype}")
        self.dtype = dtype
        self.lib = L.load(dtype)
        self.cfg = cfg
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise L.SeedmiError("DetokenizerEngine needs a HIP device (cuda:N); there is no CPU path")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        with torch.cuda.device(self.device):
            L.check(self.lib.seedmi_check_device(), "seedmi_check_device", self.lib)
        missing = [k for k in DETOK_KEYS if k not in state_dict]
        if missing:
            raise KeyError(f"state dict has no de-tokenizer weights (missing {missing})")
        self._keep = []
        self._ws = None
        self._ws_batch = 0
        self._pack(state_dict)

    def _dev(self, t: torch.Tensor) -> torch.Tensor:
        t = t.detach().to(device=self.device, dtype=self.dtype).contiguous()
        self._keep.append(t)
        return t

    def _padded(self, t: torch.Tensor, rows: int, cols: int) -> torch.Tensor:
        out = torch.zeros(rows, cols, dtype=torch.float32)
        t = t.detach().float().cpu()
        if t.dim() == 1:
            out = torch.zeros(rows, dtype=torch.float32)
            out[:t.shape[0]] = t
        else:
            out[:t.shape[0], :t.shape[1]] = t
        return self._dev(out)

    def _pack(self, sd):
        cfg = self.cfg
        Q, cd = cfg.qf_dim, cfg.code_dim
        cp = (cd + 63) // 64 * 64
        p = L.ptr
        w = L.DetokWeights()
        w.n_embed, w.code_dim, w.code_pad = cfg.n_embed, cd, cp
        w.dim, w.heads, w.ffn, w.depth = Q, cfg.dec_heads, cfg.dec_ffn, cfg.decode_depth
        w.n_query, w.down1, w.down2, w.down3, w.out_dim = cfg.n_query, cfg.down1, cfg.down2, cfg.down3, cfg.image_features_dim
        assert tuple(sd["quantize.embedding.weight"].shape) == (cfg.n_embed, cd)
        assert tuple(sd["distill_image_proj.weight"].shape) == (cfg.image_features_dim, cfg.n_query * cfg.down3)
        w.codebook_pad = p(self._padded(sd["quantize.embedding.weight"], cfg.n_embed, cp))
        w.dec_w0 = p(self._padded(sd["decode_task_layer.0.weight"], cp, cp))
        w.dec_b0 = p(self._padded(sd["decode_task_layer.0.bias"], cp, 0))
        w.dec_w1 = p(self._padded(sd["decode_task_layer.2.weight"], Q, cp))
        w.dec_b1 = p(self._dev(sd["decode_task_layer.2.bias"]))
        w.pos_embed_image = p(self._dev(sd["pos_embed_image"].reshape(cfg.n_query, Q)))
        self._blocks = (L.VitLayer * max(cfg.decode_depth, 1))()
        for i in range(cfg.decode_depth):
            b, pre = self._blocks[i], f"blocks_image.{i}."
            for field, key in (("ln1_w", "norm1.weight"), ("ln1_b", "norm1.bias"), ("qkv_w", "attn.qkv.weight"),
                               ("qkv_b", "attn.qkv.bias"), ("proj_w", "attn.proj.weight"), ("proj_b", "attn.proj.bias"),
                               ("ln2_w", "norm2.weight"), ("ln2_b", "norm2.bias"), ("fc1_w", "mlp.fc1.weight"),
                               ("fc1_b", "mlp.fc1.bias"), ("fc2_w", "mlp.fc2.weight"), ("fc2_b", "mlp.fc2.bias")):
                setattr(b, field, p(self._dev(sd[pre + key])))
        w.blocks = C.cast(self._blocks, C.POINTER(L.VitLayer))
        w.down_w0 = p(self._dev(sd["image_down.0.weight"]))
        w.down_w1 = p(self._dev(sd["image_down.2.weight"]))
        w.down_w2 = p(self._dev(sd["image_down.4.weight"]))
        w.distill_w = p(self._dev(sd["distill_image_proj.weight"]))
        w.distill_b = p(self._dev(sd["distill_image_proj.bias"]))
        self.w = w

    def _workspace(self, batch: int) -> torch.Tensor:
        if self._ws is None or batch > self._ws_batch:
            nbytes = self.lib.seedmi_detokenize_workspace_bytes(C.byref(self.w), batch)
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self._ws_batch = batch
        return self._ws

    def codebook_entry(self, indices: torch.Tensor, taps: Optional[dict] = None) -> torch.Tensor:
        """indices int64 [B, n_query] (or [n_query]) on this engine's device, values in [0, n_embed).
        Returns [B, image_features_dim] in the engine's dtype on the same device (stream ordered, no host sync)."""
        cfg = self.cfg
        if indices.dim() == 1:
            indices = indices.unsqueeze(0)
        if indices.dim() != 2 or indices.shape[1] != cfg.n_query:
            raise ValueError(f"indices must be [B, {cfg.n_query}], got {tuple(indices.shape)}")
        if indices.device != self.device:
            raise L.SeedmiError(f"indices on {indices.device}, engine on {self.device}")
        ids = indices.to(torch.int64).contiguous()
        B = ids.shape[0]
        out = torch.empty(B, cfg.image_features_dim, dtype=self.dtype, device=self.device)
        hid = None
        if taps is not None:
            hid = torch.empty(B * cfg.n_query, cfg.qf_dim, dtype=self.dtype, device=self.device)
            taps["hidden"] = hid.view(B, cfg.n_query, -1)
        ws = self._workspace(B)
        with torch.cuda.device(self.device):
            rc = self.lib.seedmi_detokenize(C.byref(self.w), L.ptr(ids), B, L.ptr(out), L.ptr(hid), L.ptr(ws), ws.numel(),
                                            L.stream_ptr())
        L.check(rc, "seedmi_detokenize", self.lib)
        return out
