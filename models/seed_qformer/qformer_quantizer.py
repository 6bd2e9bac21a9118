"""Drop-in for the reference's models/seed_qformer/qformer_quantizer.py (encode side), backed by libseedmi.so.

``Blip2QformerQuantizer`` keeps the reference's construction and call surface
(``from_pretrained(pretrained_model_path, **kwargs)``, ``get_codebook_indices(image) -> (embed_ind, query_output_up)``,
``.to()/.half()/.eval()``, attributes ``n_embed``, ``visual_encoder``) — qformer_quantizer.py:143-375 — but holds
no torch modules: the state dict is repacked once into the HIP engine (seed_amd/tokenizer_engine.py) and
``get_codebook_indices`` is a single C-ABI call (seedmi_tokenize).  The de-tokenizer front half
(``get_codebook_entry``, qformer_quantizer.py:309-338: ids -> unCLIP image embeds) is a single C-ABI call too
(seedmi_detokenize) when the checkpoint carries ``blocks_image`` / ``image_down`` / ``distill_image_proj``; the diffusers
pipeline behind it stays outside this library.
"""
import warnings

import torch

from seed_amd.config import TokenizerConfig, SEED2
from seed_amd.tokenizer_engine import TokenizerEngine
from seed_amd.detokenizer_engine import DetokenizerEngine, has_detokenizer_weights


class _DeviceHandle:
    """Stands in for sub-modules that serving code moves between devices
    (gradio_demo/seed_llama_flask.py:72-80 calls ``model.visual_encoder.to(...)``): weights stay resident in HBM
    (2.18 GB of 288 GB), so offload requests are accepted and ignored."""

    def to(self, *a, **k):
        return self

    def cpu(self):
        return self

    def cuda(self, *a, **k):
        return self


def _norm_device(device):
    """torch.device('cuda') and torch.device('cuda', current) name the same device: compare with the index filled in."""
    dev = torch.device(device)
    if dev.type == "cuda" and dev.index is None and torch.cuda.is_available():
        dev = torch.device("cuda", torch.cuda.current_device())
    return dev


class Blip2QformerQuantizer:
    def __init__(self, state_dict=None, cfg: TokenizerConfig = SEED2, device="cuda", **kwargs):
        self.cfg = cfg
        self.n_embed = cfg.n_embed
        self.codebook_embed_dim = cfg.code_dim
        self.visual_encoder = _DeviceHandle()
        self._state_dict = state_dict
        self._device = _norm_device(device) if device is not None else None
        self._engine = None
        self._out_dtype = torch.bfloat16              # dtype handed to the diffusers pipeline by get_codebook_entry
        self._compute_dtype = torch.bfloat16          # the encode path's 16-bit element: bf16 (BASELINE.json) until .half() asks for fp16
        self._detok = None

    # -- reference constructor path (qformer_quantizer.py:340-375)
    @classmethod
    def from_pretrained(cls, pretrained_model_path, **kwargs):
        cfg = kwargs.pop("cfg", SEED2)
        device = kwargs.pop("device", "cuda")
        kwargs.pop("vit_precision", None)             # qformer_quantizer.py:341: 'fp16' | 'fp32'; the compute type is chosen by .half() / .bfloat16()
        if isinstance(pretrained_model_path, dict):
            ckpt = pretrained_model_path
        elif str(pretrained_model_path).startswith("http"):
            raise RuntimeError(f"offline: cannot download {pretrained_model_path}; pass a local seed_quantizer.pt")
        else:
            ckpt = torch.load(pretrained_model_path, map_location="cpu")
        return cls(state_dict=ckpt, cfg=cfg, device=device)

    # -- nn.Module-ish plumbing used by ImageTokenizer (seed_llama_tokenizer.py:38,48,58-59)
    def eval(self):
        return self

    def _set_compute(self, dtype):
        """The tokenizer engine computes in ONE 16-bit element type, chosen here: changing it drops the packed engine (rebuilt on demand)."""
        if dtype != self._compute_dtype:
            self._compute_dtype = dtype
            self._engine = None
            self._detok = None

    def half(self):
        """The reference's shipped setting (configs/tokenizer/seed_llama_tokenizer_hf.yaml:3 `fp16: True`, seed_llama_tokenizer.py:58-59):
        the encode path then runs through libseedmi_f16.so - the same kernels and rounding places with IEEE fp16 as the 16-bit element
        (round 5; until then .half() only warned and kept bf16).  The de-tokenizer front half follows (seed_llama_tokenizer.py:62-63
        .half()s it too)."""
        self._set_compute(torch.float16)
        self._out_dtype = torch.float16               # get_codebook_entry feeds an fp16 diffusers pipeline (:309-338)
        return self

    def bfloat16(self):
        self._set_compute(torch.bfloat16)
        self._out_dtype = torch.bfloat16
        return self

    def float(self):
        """fp32 callers get the bf16 engine and fp32 OUTPUTS - and are told so: the reference with ``fp16=False`` keeps fp32 parameters,
        runs only the ViT under autocast (blip2.py:40-48, qformer_quantizer.py:290-291) and the Q-Former, task MLP and VQ distances in
        fp32 (qformer_quantizer.py:293-303, seed_llama_tokenizer.py:35-37,58-59); this library has no fp32 compute path (BASELINE.json's
        dtype is bf16, the shipped yaml is fp16: both are served).  Ids can differ from such a reference run on near-tie rows
        (bf16 vs fp32: ~7 % of ids on the i.i.d. synthetic codebook, 0 % on the peaked case - profiles/r06_tokenizer_margin_coverage.json)."""
        warnings.warn("Blip2QformerQuantizer.float(): no fp32 compute path - the Q-Former, task MLP and VQ, which the reference runs in fp32 "
                      "when fp16=False (only its ViT is under autocast), run in bf16 here; outputs are returned as float32. "
                      "Use .half() (the reference's shipped fp16 setting) or .bfloat16() to choose the 16-bit type explicitly.",
                      RuntimeWarning, stacklevel=2)
        self._set_compute(torch.bfloat16)
        self._out_dtype = torch.float32
        return self

    def to(self, device=None, *a, **k):
        if device is not None and not isinstance(device, torch.dtype):
            dev = _norm_device(device)
            if self._engine is not None and dev != self._engine.device:
                self._engine = None
            if self._detok is not None and dev != self._detok.device:
                self._detok = None
            self._device = dev
        return self

    @property
    def engine(self) -> TokenizerEngine:
        if self._engine is None:
            if self._state_dict is None:
                raise RuntimeError("Blip2QformerQuantizer has no weights (use from_pretrained or pass state_dict)")
            self._engine = TokenizerEngine(self._state_dict, self.cfg, device=self._device, dtype=self._compute_dtype)   # raises without a GPU
        return self._engine

    def get_codebook_indices(self, image):
        """qformer_quantizer.py:288-307.  Returns (embed_ind int64 [B,32], None): ``query_output_up`` feeds only the
        de-tokenizer and is dead work for encode_image, so it is not computed."""
        with torch.no_grad():
            return self.engine.encode(image), None

    @property
    def detokenizer(self) -> DetokenizerEngine:
        if self._detok is None:
            if self._state_dict is None or not has_detokenizer_weights(self._state_dict):
                raise RuntimeError("this checkpoint carries no de-tokenizer weights (blocks_image / image_down / "
                                   "distill_image_proj)")
            self._detok = DetokenizerEngine(self._state_dict, self.cfg, device=self._device, dtype=self._compute_dtype)     # raises without a GPU
        return self._detok

    def get_codebook_entry(self, indices):
        """qformer_quantizer.py:309-338 (use_qformer_image=False): ids [B,32] -> image embeds [B,1024] in the dtype last asked for with .half() / .bfloat16() / .float() (computed in fp16 after .half(), in bf16 otherwise)."""
        with torch.no_grad():
            return self.detokenizer.codebook_entry(indices).to(self._out_dtype)
