"""Model-shape constants of the SEED tokenize-and-generate hot path.

Every number is taken from the reference's code (not from a checkpoint):
* EVA-ViT-g/14: models/seed_qformer/eva_vit.py:461-474
* causal Q-Former = bert-base-uncased + cross-attn every 2nd layer, 32 queries:
  models/seed_qformer/blip2.py:52-63, qformer_quantizer.py:161-173
* codebook 8192 x 32, task head 768->768->32: qformer_quantizer.py:217-223
* LLaMA bodies: the HF checkpoints named in README.md:138 (Vicuna-7B for SEED-LLaMA-8B,
  LLaMA-2-13B for 14B); vocab = 32000 text + 8192 image codes + <img>,</img> (+pad)
  (scripts/seed_llama_inference_8B.py:21-23).
"""
from dataclasses import dataclass, asdict


@dataclass(frozen=True)
class TokenizerConfig:
    img_size: int = 224
    patch: int = 14
    vit_dim: int = 1408
    vit_depth: int = 39
    vit_heads: int = 16
    vit_mlp_ratio: float = 4.3637
    qf_dim: int = 768
    qf_layers: int = 12
    qf_heads: int = 12
    qf_ffn: int = 3072
    cross_freq: int = 2
    n_query: int = 32
    n_embed: int = 8192
    code_dim: int = 32
    # de-tokenizer front half (qformer_quantizer.py:176-181, 249-286): blocks_image + image_down + distill_image_proj
    decode_depth: int = 4
    dec_heads: int = 12
    dec_mlp_ratio: float = 4.0
    down1: int = 256
    down2: int = 128
    down3: int = 32
    image_features_dim: int = 1024

    @property
    def dec_ffn(self) -> int:  # vit.py:135  int(dim * mlp_ratio)
        return int(self.qf_dim * self.dec_mlp_ratio)

    @property
    def vit_ffn(self) -> int:  # eva_vit.py:190  int(dim * mlp_ratio)
        return int(self.vit_dim * self.vit_mlp_ratio)

    @property
    def vit_head_dim(self) -> int:
        return self.vit_dim // self.vit_heads

    @property
    def grid(self) -> int:
        return self.img_size // self.patch

    @property
    def n_patches(self) -> int:
        return self.grid * self.grid

    @property
    def n_tokens(self) -> int:  # cls + patches
        return self.n_patches + 1

    @property
    def patch_k(self) -> int:  # conv-as-GEMM contraction length, (c, kh, kw) order
        return 3 * self.patch * self.patch

    def flops_per_image(self) -> float:
        """Algorithmic 2*M*N*K FLOPs per image (533.52 GF at full size)."""
        D, N, F, H = self.vit_dim, self.n_tokens, self.vit_ffn, self.vit_heads
        hd = self.vit_head_dim
        vit = 2 * self.n_patches * self.patch_k * D
        per_layer = 2 * N * D * 3 * D + 2 * 2 * H * N * N * hd + 2 * N * D * D + 2 * 2 * N * D * F
        vit += self.vit_depth * per_layer
        q, Q, FF = self.n_query, self.qf_dim, self.qf_ffn
        qhd = Q // self.qf_heads
        qf = 0.0
        for layer in range(self.qf_layers):
            qf += 2 * q * Q * 3 * Q + 2 * 2 * self.qf_heads * q * q * qhd + 2 * q * Q * Q
            if layer % self.cross_freq == 0:
                qf += 2 * q * Q * Q + 2 * 2 * N * D * Q + 2 * 2 * self.qf_heads * q * N * qhd + 2 * q * Q * Q
            qf += 2 * 2 * q * Q * FF
        head = 2 * q * Q * Q + 2 * q * Q * self.code_dim + 2 * q * self.code_dim * self.n_embed
        return float(vit + qf + head)

    def to_dict(self):
        return asdict(self)


# full-size SEED-2 tokenizer
SEED2 = TokenizerConfig()
# reduced shapes for CPU-speed parity tests (same structure: odd head dim 88-like handled by 'mid')
TINY = TokenizerConfig(img_size=56, patch=14, vit_dim=128, vit_depth=2, vit_heads=2, vit_mlp_ratio=4.0,
                       qf_dim=128, qf_layers=2, qf_heads=2, qf_ffn=256, n_query=32, n_embed=512, code_dim=32,
                       decode_depth=2, dec_heads=2, down1=128, down2=64, down3=32, image_features_dim=256)
# keeps the MFMA-hostile dims of the real model (hd = 88, 257 tokens, 64-wide Q-Former heads) at small depth
MID = TokenizerConfig(img_size=224, patch=14, vit_dim=704, vit_depth=2, vit_heads=8, vit_mlp_ratio=4.0,
                      qf_dim=256, qf_layers=2, qf_heads=4, qf_ffn=512, n_query=32, n_embed=8192, code_dim=32,
                      decode_depth=2, dec_heads=4)


@dataclass(frozen=True)
class LlamaConfig:
    hidden: int = 4096
    layers: int = 32
    heads: int = 32
    ffn: int = 11008
    vocab: int = 40194
    rms_eps: float = 1e-6
    max_pos: int = 2048
    rope_base: float = 10000.0

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads

    def linear_params(self) -> int:
        """Parameters in the per-token linear path (all layers + lm_head)."""
        h, f = self.hidden, self.ffn
        return self.layers * (4 * h * h + 3 * h * f) + self.vocab * h

    def kv_bytes_per_token(self, itemsize: int = 2) -> int:
        return 2 * self.layers * self.hidden * itemsize

    def to_dict(self):
        return asdict(self)


LLAMA_8B = LlamaConfig()
LLAMA_14B = LlamaConfig(hidden=5120, layers=40, heads=40, ffn=13824, rms_eps=1e-5, max_pos=4096)
LLAMA_TINY = LlamaConfig(hidden=256, layers=2, heads=2, ffn=512, vocab=1024 + 130, rms_eps=1e-6, max_pos=256)
