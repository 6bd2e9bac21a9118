"""Drop-in for the reference's models/llama_xformer.py: ``LlamaForCausalLM`` with the same constructor, state-dict
key names, ``forward`` signature/outputs, ``prepare_inputs_for_generation`` and ``_reorder_cache``, whose forward
runs on the MI355X-native path (one ``seedmi_llama_forward`` C-ABI call) instead of PyTorch + xformers.

Reference lines mirrored: LlamaForCausalLM.forward 661-743, LlamaModel.forward 496-627 (argument checks 514-541),
prepare_inputs_for_generation 745-776, _reorder_cache 778-783, _init_weights 363-372.

* The torch modules below only *hold* parameters under the HF names (``model.layers.N.self_attn.q_proj.weight`` ...) so
  ``from_pretrained`` / ``load_state_dict`` work unchanged; at the first forward they are repacked into the HIP engine
  (seed_amd/llama_engine.py) and, by default, released.
* ``past_key_values`` keeps the reference's legacy layout — a tuple over layers of ``(k, v)`` ``[B,H,T,128]`` with
  post-RoPE keys — as zero-copy views of the engine's static KV cache.
* transformers >= 4.50 no longer mixes ``GenerationMixin`` into ``PreTrainedModel`` and hands ``generate()`` a
  ``Cache`` object; this class inherits the mixin explicitly and opts out of the default
  ``DynamicCache`` so the scripts' ``model.generate(...)`` calls run unchanged.
* Semantics differences, by design: attention is causal + correct under the reference's unpadded equal-length
  batches (its eval path ignores padding masks); compute dtype is bf16 with fp32 accumulation.
"""
from typing import List, Optional, Tuple, Union

import warnings

import torch
import torch.nn as nn
from torch.nn import CrossEntropyLoss
from transformers import GenerationMixin, PreTrainedModel
from transformers.modeling_outputs import CausalLMOutputWithPast
from transformers.models.llama.configuration_llama import LlamaConfig

from seed_amd.config import LlamaConfig as EngineConfig


class _Weight(nn.Module):
    """Parameter holder with an nn.Linear / nn.Embedding / RMSNorm compatible ``.weight``."""

    def __init__(self, *shape):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(*shape))


class LlamaAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        h = config.hidden_size
        self.q_proj, self.k_proj, self.v_proj, self.o_proj = _Weight(h, h), _Weight(h, h), _Weight(h, h), _Weight(h, h)


class LlamaMLP(nn.Module):
    def __init__(self, config):
        super().__init__()
        h, f = config.hidden_size, config.intermediate_size
        self.gate_proj, self.down_proj, self.up_proj = _Weight(f, h), _Weight(h, f), _Weight(f, h)


class LlamaDecoderLayer(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.self_attn = LlamaAttention(config)
        self.mlp = LlamaMLP(config)
        self.input_layernorm = _Weight(config.hidden_size)
        self.post_attention_layernorm = _Weight(config.hidden_size)


class LlamaModel(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.embed_tokens = _Weight(config.vocab_size, config.hidden_size)
        self.layers = nn.ModuleList([LlamaDecoderLayer(config) for _ in range(config.num_hidden_layers)])
        self.norm = _Weight(config.hidden_size)


class PastKeyValues(tuple):
    """The reference's legacy cache layout (tuple over layers of (k, v) [B,H,T,hd]) plus the two ``Cache`` methods
    newer ``generate()`` loops query."""

    def get_seq_length(self, layer_idx: int = 0) -> int:
        return self[0][0].shape[2] if len(self) else 0

    def get_max_cache_shape(self, layer_idx: int = 0) -> int:
        return -1


class LlamaForCausalLM(PreTrainedModel, GenerationMixin):
    config_class = LlamaConfig
    base_model_prefix = "model"
    supports_gradient_checkpointing = False
    _no_split_modules = ["LlamaDecoderLayer"]
    _skip_keys_device_placement = "past_key_values"

    def __init__(self, config):
        super().__init__(config)
        self.model = LlamaModel(config)
        self.lm_head = _Weight(config.vocab_size, config.hidden_size)
        self._engine = None
        self._weights_released = False
        self._engine_opts = {"batch_cap": 32, "tmax": None, "free_unpacked": True}
        self.post_init()

    # ------------------------------------------------------------------ HF plumbing
    def _init_weights(self, module):
        std = self.config.initializer_range           # llama_xformer.py:363-372
        if isinstance(module, _Weight):
            if module.weight.dim() == 1:
                module.weight.data.fill_(1.0)          # RMSNorm
            else:
                module.weight.data.normal_(mean=0.0, std=std)

    @classmethod
    def _supports_default_dynamic_cache(cls) -> bool:
        return False                                   # the KV cache lives in the HIP engine (static, in place)

    def get_input_embeddings(self):
        return self.model.embed_tokens

    def get_output_embeddings(self):
        return None                                    # untied, nothing to resize/tie

    def configure_engine(self, batch_cap: int = 32, tmax: Optional[int] = None, free_unpacked: bool = True):
        """Static KV-cache capacity (batch rows x positions) of the HIP engine.  Before the first forward it sets the options
        the engine is built with; afterwards it resizes the cache of the existing engine (the packed weights stay in place —
        with free_unpacked the torch-side copies are gone, so the engine is never rebuilt from them)."""
        self._engine_opts = {"batch_cap": batch_cap, "tmax": tmax, "free_unpacked": free_unpacked}
        if self._engine is not None:
            self._engine.resize_cache(batch_cap=max(batch_cap, 1), tmax=tmax)
        return self

    def _engine_config(self) -> EngineConfig:
        c = self.config
        return EngineConfig(hidden=c.hidden_size, layers=c.num_hidden_layers, heads=c.num_attention_heads,
                            ffn=c.intermediate_size, vocab=c.vocab_size, rms_eps=c.rms_norm_eps,
                            max_pos=c.max_position_embeddings,
                            rope_base=float(getattr(c, "rope_theta", None) or 10000.0))

    def _make_engine(self, device, batch):
        from seed_amd.llama_engine import LlamaEngine
        opts = self._engine_opts
        sd = {k: v for k, v in self.state_dict().items()}
        # the model dtype picks the library build: float16 parameters (the reference's `torch_dtype: fp16`, configs/llm/seed_llama_8b.yaml:4,
        # model_tools.py:7-8) -> libseedmi_f16.so; bfloat16 (and fp32 checkpoints, narrowed once) -> the default bf16 build
        pdtype = next(iter(sd.values())).dtype if sd else torch.bfloat16
        eng = LlamaEngine(sd, self._engine_config(), device=device, batch_cap=max(opts["batch_cap"], batch),
                          tmax=opts["tmax"], dtype=torch.float16 if pdtype == torch.float16 else torch.bfloat16)
        if opts["free_unpacked"]:
            for p in self.parameters():
                p.data = torch.empty(0, dtype=p.dtype, device=p.device)
            self._weights_released = True
        return eng

    def state_dict(self, *args, **kwargs):
        if getattr(self, "_weights_released", False):
            raise RuntimeError("the torch-side parameters were released after packing into the HIP engine "
                               "(configure_engine(free_unpacked=False) before the first forward keeps them for "
                               "state_dict()/save_pretrained())")
        return super().state_dict(*args, **kwargs)

    def _ensure_engine(self, device, batch):
        """Build the engine once; a later, larger batch grows its KV cache instead of repacking (the unpacked parameters may
        already have been released)."""
        if self._engine is None:
            if getattr(self, "_weights_released", False):
                raise RuntimeError("LlamaForCausalLM: engine dropped after its weights were released; reload the checkpoint")
            self._engine = self._make_engine(device, batch)
        elif self._engine.batch_cap < batch:
            self._engine.resize_cache(batch_cap=batch)
        return self._engine

    @property
    def engine(self):
        return self._engine

    # ------------------------------------------------------------------ forward
    def forward(
        self,
        input_ids: torch.LongTensor = None,
        attention_mask: Optional[torch.Tensor] = None,
        position_ids: Optional[torch.LongTensor] = None,
        past_key_values: Optional[List[torch.FloatTensor]] = None,
        inputs_embeds: Optional[torch.FloatTensor] = None,
        labels: Optional[torch.LongTensor] = None,
        use_cache: Optional[bool] = None,
        output_attentions: Optional[bool] = None,
        output_hidden_states: Optional[bool] = None,
        return_dict: Optional[bool] = None,
        **kwargs,
    ) -> Union[Tuple, CausalLMOutputWithPast]:
        use_cache = use_cache if use_cache is not None else getattr(self.config, "use_cache", True)
        return_dict = True if return_dict is None else return_dict
        if input_ids is not None and inputs_embeds is not None:
            raise ValueError("You cannot specify both decoder_input_ids and decoder_inputs_embeds at the same time")
        if input_ids is None and inputs_embeds is None:
            raise ValueError("You have to specify either decoder_input_ids or decoder_inputs_embeds")
        if output_attentions:
            # the reference itself cannot serve this on its xformers path: LlamaAttention.forward never assigns
            # `attn_weights` before returning it (llama_xformer.py:244-263 -> UnboundLocalError); the fused kernels never
            # materialise the [B,H,T,T] probabilities either
            raise NotImplementedError("output_attentions=True: attention maps are not materialised (the reference's xformers "
                                      "path raises UnboundLocalError for the same request, llama_xformer.py:260-263)")
        if input_ids is not None:
            B, T = input_ids.shape
            dev = input_ids.device
        else:
            B, T, _ = inputs_embeds.shape                                       # llama_xformer.py:519-520
            dev = inputs_embeds.device
        eng = self._ensure_engine(dev, B)

        past_len = 0
        if past_key_values is not None and len(past_key_values) > 0:
            past_len = past_key_values[0][0].shape[2]                          # llama_xformer.py:527-529
            k0 = past_key_values[0][0]
            if k0.data_ptr() != eng.k_cache[0].data_ptr():                     # foreign cache: copy it in
                for l, (k, v) in enumerate(past_key_values):
                    eng.k_cache[l][:B, :, :past_len].copy_(k)
                    eng.v_cache[l][:B, :, :past_len].copy_(v)
        if attention_mask is not None and past_len == 0 and attention_mask.numel() > 0 and bool((attention_mask == 0).any()):
            # the reference's eval path ignores padding masks too (xformers gets a LowerTriangularMask or nothing,
            # llama_xformer.py:244-256), so a padded batch is computed as if every position were a token - in both.  Parity is
            # defined for unpadded, equal-length batches; say so instead of silently returning answers for padding.  Checked on
            # prefill calls only (one host sync), not on every cached decode step.
            warnings.warn("LlamaForCausalLM.forward: attention_mask contains zeros (a padded batch); padding masks are not applied - "
                          "like the reference's xformers path - so padded rows attend to their padding.  Use unpadded, equal-length "
                          "batches (or one sequence per call).", RuntimeWarning, stacklevel=2)
        if position_ids is not None:
            position_ids = position_ids.view(-1, T).long()                      # :541
            if position_ids.shape[0] == 1 and B > 1:
                position_ids = position_ids.expand(B, T)
        hidden = [] if output_hidden_states else None
        logits = eng.forward(input_ids, position_ids=position_ids, past_len=past_len, last_only=False,
                             inputs_embeds=inputs_embeds, hidden_states_out=hidden)

        loss = None
        if labels is not None:                                                  # :721-731
            shift_logits = logits[..., :-1, :].float().contiguous().view(-1, self.config.vocab_size)
            shift_labels = labels[..., 1:].contiguous().view(-1).to(shift_logits.device)
            loss = CrossEntropyLoss()(shift_logits, shift_labels)

        past = None
        if use_cache:
            total = past_len + T
            past = PastKeyValues((eng.k_cache[l][:B, :, :total], eng.v_cache[l][:B, :, :total])
                                 for l in range(len(eng.k_cache)))
        if not return_dict:
            out = (logits,) + ((past,) if past is not None else ()) + ((tuple(hidden),) if hidden is not None else ())
            return ((loss,) + out) if loss is not None else out
        return CausalLMOutputWithPast(loss=loss, logits=logits, past_key_values=past,
                                      hidden_states=tuple(hidden) if hidden is not None else None, attentions=None)

    # ------------------------------------------------------------------ generation glue (llama_xformer.py:745-783)
    def prepare_inputs_for_generation(self, input_ids, past_key_values=None, attention_mask=None, inputs_embeds=None,
                                      **kwargs):
        if past_key_values:
            input_ids = input_ids[:, -1:]
        position_ids = kwargs.get("position_ids", None)
        if attention_mask is not None and position_ids is None:
            # create position_ids on the fly for batch generation
            position_ids = attention_mask.long().cumsum(-1) - 1
            position_ids.masked_fill_(attention_mask == 0, 1)
            if past_key_values:
                position_ids = position_ids[:, -1].unsqueeze(-1)
        elif position_ids is not None:
            position_ids = position_ids[:, -input_ids.shape[1]:]
        if inputs_embeds is not None and past_key_values is None:
            model_inputs = {"inputs_embeds": inputs_embeds}
        else:
            model_inputs = {"input_ids": input_ids}
        model_inputs.update({
            "position_ids": position_ids,
            "past_key_values": past_key_values,
            "use_cache": kwargs.get("use_cache"),
            "attention_mask": attention_mask,
        })
        return model_inputs

    def _reorder_cache(self, past_key_values, beam_idx):
        """Beam search support: rows of the static cache are permuted in place (the reference index_selects copies)."""
        eng = self._engine
        n = beam_idx.shape[0]
        T = past_key_values[0][0].shape[2]
        for l in range(len(eng.k_cache)):
            eng.k_cache[l][:n, :, :T] = eng.k_cache[l][:n, :, :T].index_select(0, beam_idx)
            eng.v_cache[l][:n, :, :T] = eng.v_cache[l][:n, :, :T].index_select(0, beam_idx)
        return PastKeyValues((eng.k_cache[l][:n, :, :T], eng.v_cache[l][:n, :, :T]) for l in range(len(eng.k_cache)))
