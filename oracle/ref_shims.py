"""Import the reference's OWN modules (test infrastructure only).

TEST INFRASTRUCTURE — never imported by the product path (seed_amd/, models/).
Source of the modules: ``/root/reference`` where that tree exists (the build container), otherwise the sourceless
bytecode that ``oracle/build_ref.py`` compiled from it into the git-ignored ``oracle/_ref/`` (which travels to the GPU
box with the snapshot, like the built ``.so`` files).  Used by ``oracle/make_golden.py`` (golden vectors), by
``bench.py``'s ``cpu_baseline`` leg (the reference's own CPU path timed on the bench node) and by the ``-m gpu`` tests
that compare the HIP path with live reference modules, and to pin ``oracle/seed_oracle.py`` (the CPU restatement)
against the reference's real code.

The reference cannot be imported as-is under torch 2.10 / transformers 5.15
(SURVEY.md §8c): ``timm`` and ``xformers`` are not installed and a few
transformers symbols moved.  The shims below are injected into ``sys.modules``
*before* the reference modules are imported; none of them changes arithmetic:

* ``timm.models.layers``: ``drop_path`` (identity at p=0/eval), ``to_2tuple``,
  ``trunc_normal_`` (reference use: models/seed_qformer/eva_vit.py:15).
* ``timm.models.hub``: ``download_cached_file``/``get_cache_dir`` stubs
  (models/seed_qformer/utils.py:14) — never called, we bypass the downloaders.
* ``transformers.modeling_utils.{apply_chunking_to_forward, prune_linear_layer,
  find_pruneable_heads_and_indices}`` (models/seed_qformer/qformer_causual.py:38-43).
* ``xformers.ops``: ``LowerTriangularMask`` + ``memory_efficient_attention``
  restated with explicit matmul/softmax in [B,T,H,D] layout, scale 1/sqrt(D),
  top-left aligned causal mask (models/llama_xformer.py:244-256).
"""
import importlib
import importlib.util
import math
import os
import sys
import types

import torch
import torch.nn as nn

REFERENCE_ROOT = os.environ.get("SEED_REFERENCE_ROOT", "/root/reference")
COMPILED_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")


def _source_tree() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "models", "seed_qformer"))


def reference_available() -> bool:
    """The reference tree itself, or its compiled modules under oracle/_ref (oracle/build_ref.py)."""
    if _source_tree():
        return True
    from . import build_ref
    return build_ref.available()


def reference_origin() -> str:
    return REFERENCE_ROOT if _source_tree() else "oracle/_ref (bytecode compiled from /root/reference by oracle/build_ref.py)"


def _install_shims():
    import transformers  # noqa: F401  (must be imported before we alias into it)
    import transformers.modeling_utils as mu
    import transformers.pytorch_utils as pu

    if "timm" not in sys.modules:
        timm = types.ModuleType("timm")
        timm_models = types.ModuleType("timm.models")
        layers = types.ModuleType("timm.models.layers")
        hub = types.ModuleType("timm.models.hub")

        def drop_path(x, drop_prob: float = 0.0, training: bool = False):
            assert drop_prob == 0.0 or not training
            return x

        def to_2tuple(x):
            return tuple(x) if isinstance(x, (tuple, list)) else (x, x)

        layers.drop_path = drop_path
        layers.to_2tuple = to_2tuple
        layers.trunc_normal_ = nn.init.trunc_normal_
        hub.download_cached_file = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("offline"))
        hub.get_cache_dir = lambda *a, **k: "/tmp"
        # models/seed_qformer/vit.py:17-20 (the de-tokenizer's Block lives there) imports a few more timm names; only
        # DropPath is ever instantiated on our path, and only with drop_prob == 0 (vit.py:133 picks nn.Identity then)
        class DropPath(nn.Module):
            def __init__(self, drop_prob=0.0):
                super().__init__()
                assert drop_prob == 0.0
            def forward(self, x):
                return x

        layers.DropPath = DropPath
        vt = types.ModuleType("timm.models.vision_transformer")
        vt._cfg = lambda **kw: dict(kw)
        vt.PatchEmbed = type("PatchEmbed", (nn.Module,), {})
        reg = types.ModuleType("timm.models.registry")
        reg.register_model = lambda fn: fn
        helpers = types.ModuleType("timm.models.helpers")
        helpers.named_apply = lambda *a, **k: None
        helpers.adapt_input_conv = lambda *a, **k: None
        timm.models = timm_models
        timm_models.layers = layers
        timm_models.hub = hub
        timm_models.vision_transformer = vt
        timm_models.registry = reg
        timm_models.helpers = helpers
        sys.modules.update({"timm": timm, "timm.models": timm_models,
                            "timm.models.layers": layers, "timm.models.hub": hub,
                            "timm.models.vision_transformer": vt, "timm.models.registry": reg,
                            "timm.models.helpers": helpers})

    for name in ("apply_chunking_to_forward", "prune_linear_layer"):
        if not hasattr(mu, name):
            setattr(mu, name, getattr(pu, name))
    if not hasattr(mu, "find_pruneable_heads_and_indices"):
        def find_pruneable_heads_and_indices(*a, **k):
            raise NotImplementedError("head pruning is not on the hot path")
        mu.find_pruneable_heads_and_indices = find_pruneable_heads_and_indices

    # transformers' @add_start_docstrings_to_model_forward (llama_xformer.py:495, 660) reads the decorated function's SOURCE
    # to indent a docstring; bytecode compiled into oracle/_ref has none.  Docstrings only - a method of a class sits at depth 4.
    import transformers.utils.doc as tdoc
    if not getattr(tdoc.get_docstring_indentation_level, "_seed_sourceless_ok", False):
        _orig_level = tdoc.get_docstring_indentation_level

        def get_docstring_indentation_level(func):
            try:
                return _orig_level(func)
            except OSError:
                return 8 if "." in getattr(func, "__qualname__", "") else 4
        get_docstring_indentation_level._seed_sourceless_ok = True
        tdoc.get_docstring_indentation_level = get_docstring_indentation_level

    if "xformers" not in sys.modules:
        xf = types.ModuleType("xformers")
        xops = types.ModuleType("xformers.ops")

        class LowerTriangularMask:  # marker type, xformers semantics: top-left aligned
            pass

        def memory_efficient_attention(query, key, value, attn_bias=None, p=0.0, scale=None):
            # [B, Tq, H, D] / [B, Tk, H, D]; fp32 softmax like the CUDA kernels
            q = query.transpose(1, 2).float()
            k = key.transpose(1, 2).float()
            v = value.transpose(1, 2).float()
            d = q.shape[-1]
            s = torch.matmul(q, k.transpose(-1, -2)) * (scale if scale is not None else 1.0 / math.sqrt(d))
            if isinstance(attn_bias, LowerTriangularMask):
                tq, tk = s.shape[-2:]
                keep = torch.ones(tq, tk, dtype=torch.bool).tril()
                s = s.masked_fill(~keep, float("-inf"))
            else:
                assert attn_bias is None
            p_ = torch.softmax(s, dim=-1)
            o = torch.matmul(p_.to(value.dtype).float(), v)
            return o.transpose(1, 2).to(query.dtype)

        xops.LowerTriangularMask = LowerTriangularMask
        xops.memory_efficient_attention = memory_efficient_attention
        xf.ops = xops
        sys.modules.update({"xformers": xf, "xformers.ops": xops})


_PKG = "_seedref"


def _load(relpath: str, modname: str):
    """Load one reference source file as ``_seedref.<modname>`` (relative imports work)."""
    full = f"{_PKG}.{modname}"
    if full in sys.modules:
        return sys.modules[full]
    path = os.path.join(REFERENCE_ROOT, relpath) if _source_tree() else os.path.join(COMPILED_ROOT, relpath + "c")
    spec = importlib.util.spec_from_file_location(full, path)       # ".pyc" selects the SourcelessFileLoader
    mod = importlib.util.module_from_spec(spec)
    sys.modules[full] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference_modules():
    """Returns a namespace with the reference's eva_vit, qformer_causual, VectorQuantizer2, LayerNorm, llama."""
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT} and no compiled copy under {COMPILED_ROOT}")
    _install_shims()
    if _PKG not in sys.modules:
        pkg = types.ModuleType(_PKG)
        pkg.__path__ = []  # mark as package
        sys.modules[_PKG] = pkg
    utils = _load("models/seed_qformer/utils.py", "utils")
    eva = _load("models/seed_qformer/eva_vit.py", "eva_vit")
    qf = _load("models/seed_qformer/qformer_causual.py", "qformer_causual")
    llama = _load("models/llama_xformer.py", "llama_xformer")

    # BertPreTrainedModel.init_weights / get_head_mask drifted in transformers 5.x
    qf.BertPreTrainedModel.init_weights = lambda self: self.apply(self._init_weights)
    qf.BertModel.get_head_mask = lambda self, hm, n, **k: [None] * n

    # qformer_quantizer.py and blip2.py import diffusers-free but network-bound helpers;
    # we only need two small classes from them.  Load them through stub siblings.
    for stub in ("clip_vit", "vit"):
        full = f"{_PKG}.{stub}"
        if full not in sys.modules:
            try:
                _load(f"models/seed_qformer/{stub}.py", stub)
            except Exception:  # pragma: no cover
                m = types.ModuleType(full)
                m.create_clip_vit_L = None
                m.Block = None
                sys.modules[full] = m
    blip2 = _load("models/seed_qformer/blip2.py", "blip2")
    qq = _load("models/seed_qformer/qformer_quantizer.py", "qformer_quantizer")

    ns = types.SimpleNamespace(eva_vit=eva, qformer=qf, blip2=blip2, quantizer=qq, llama=llama, utils=utils)
    return ns


def build_reference_tokenizer_modules(ref, cfg, dtype=torch.float32):
    """Instantiate the reference sub-modules with the network constructors bypassed.

    cfg: oracle.seed_oracle.TokenizerConfig.  Mirrors
    models/seed_qformer/eva_vit.py:461-474, blip2.py:52-63, qformer_quantizer.py:204-223.
    """
    from functools import partial
    vit = ref.eva_vit.VisionTransformer(
        img_size=cfg.img_size, patch_size=cfg.patch, use_mean_pooling=False, embed_dim=cfg.vit_dim,
        depth=cfg.vit_depth, num_heads=cfg.vit_heads, mlp_ratio=cfg.vit_mlp_ratio, qkv_bias=True,
        drop_path_rate=0.0, norm_layer=partial(nn.LayerNorm, eps=1e-6))
    ln_vision = ref.blip2.LayerNorm(cfg.vit_dim)
    bcfg = ref.qformer.BertConfig(hidden_size=cfg.qf_dim, num_hidden_layers=cfg.qf_layers,
                                  num_attention_heads=cfg.qf_heads, intermediate_size=cfg.qf_ffn)
    bcfg.encoder_width = cfg.vit_dim
    bcfg.add_cross_attention = True
    bcfg.cross_attention_freq = cfg.cross_freq
    bcfg.query_length = cfg.n_query
    qformer = ref.qformer.BertLMHeadModel(bcfg)
    # strip exactly like qformer_quantizer.py:206-211
    qformer.cls = None
    qformer.bert.embeddings.word_embeddings = None
    qformer.bert.embeddings.position_embeddings = None
    for layer in qformer.bert.encoder.layer:
        layer.output = None
        layer.intermediate = None
    quantize = ref.quantizer.VectorQuantizer2(cfg.n_embed, cfg.code_dim, beta=0.25, remap=None, sane_index_shape=False)
    encode_task_layer = nn.Sequential(nn.Linear(cfg.qf_dim, cfg.qf_dim), nn.Tanh(), nn.Linear(cfg.qf_dim, cfg.code_dim))
    mods = types.SimpleNamespace(visual_encoder=vit.eval(), ln_vision=ln_vision.eval(), Qformer=qformer.eval(),
                                 quantize=quantize.eval(), encode_task_layer=encode_task_layer.eval())
    return mods


def reference_get_codebook_indices(mods, query_tokens, image):
    """Line-for-line re-assembly of Blip2QformerQuantizer.get_codebook_indices
    (models/seed_qformer/qformer_quantizer.py:288-307) on the reference sub-modules;
    ``maybe_autocast`` is a nullcontext on CPU (blip2.py:45-50)."""
    with torch.no_grad():
        image_embeds = mods.ln_vision(mods.visual_encoder(image))
        image_atts = torch.ones(image_embeds.size()[:-1], dtype=torch.long)
        q = query_tokens.expand(image_embeds.shape[0], -1, -1)
        query_output = mods.Qformer.bert(query_embeds=q, encoder_hidden_states=image_embeds,
                                         encoder_attention_mask=image_atts, return_dict=True)
        query_output_down = mods.encode_task_layer(query_output.last_hidden_state)
        quant, loss_embed, embed_ind = mods.quantize(query_output_down)
        embed_ind = embed_ind.reshape(quant.shape[0], -1)
    return embed_ind, dict(image_embeds=image_embeds, qformer_out=query_output.last_hidden_state, z=query_output_down)


def build_reference_detokenizer_modules(ref, cfg):
    """The reference sub-modules get_codebook_entry touches, instantiated as Blip2QformerQuantizer.__init__ does
    (models/seed_qformer/qformer_quantizer.py:217, 225-229, 249-262, 279-286) with the network constructors bypassed."""
    from functools import partial
    vit_mod = sys.modules[f"{_PKG}.vit"]
    Q = cfg.qf_dim
    quantize = ref.quantizer.VectorQuantizer2(cfg.n_embed, cfg.code_dim, beta=0.25, remap=None, sane_index_shape=False)
    decode_task_layer = nn.Sequential(nn.Linear(cfg.code_dim, cfg.code_dim), nn.Tanh(), nn.Linear(cfg.code_dim, Q))
    blocks_image = nn.ModuleList([
        vit_mod.Block(dim=Q, num_heads=cfg.dec_heads, mlp_ratio=cfg.dec_mlp_ratio, qkv_bias=True, qk_scale=None, drop=0.0,
                      attn_drop=0.0, drop_path=0.0, norm_layer=partial(nn.LayerNorm, eps=1e-6))
        for _ in range(cfg.decode_depth)])
    image_down = nn.Sequential(nn.Linear(Q, cfg.down1, bias=False), nn.ReLU(), nn.Linear(cfg.down1, cfg.down2, bias=False),
                               nn.ReLU(), nn.Linear(cfg.down2, cfg.down3, bias=False))
    distill_image_proj = nn.Linear(cfg.n_query * cfg.down3, cfg.image_features_dim)
    holder = nn.Module()
    holder.quantize = quantize
    holder.decode_task_layer = decode_task_layer
    holder.pos_embed_image = nn.Parameter(torch.zeros(1, cfg.n_query, Q))
    holder.blocks_image = blocks_image
    holder.image_down = image_down
    holder.distill_image_proj = distill_image_proj
    return holder.eval()


def reference_get_codebook_entry(mods, indices):
    """Line-for-line re-assembly of Blip2QformerQuantizer.get_codebook_entry for use_qformer_image=False
    (models/seed_qformer/qformer_quantizer.py:309-320, 332-338) on the reference sub-modules."""
    with torch.no_grad():
        quant_embedding = mods.quantize.get_codebook_entry(indices)
        query_output_up = mods.decode_task_layer(quant_embedding)
        pos_embed_image = mods.pos_embed_image.repeat(query_output_up.shape[0], 1, 1)
        query_output_up_pos_image = query_output_up + pos_embed_image
        for blk in mods.blocks_image:
            query_output_up_pos_image = blk(query_output_up_pos_image)
        query_output_up = query_output_up_pos_image
        reverse_output = mods.image_down(query_output_up)
        reverse_output = reverse_output.reshape(reverse_output.shape[0], -1)
        reverse_output_proj = mods.distill_image_proj(reverse_output)
    return reverse_output_proj, query_output_up
