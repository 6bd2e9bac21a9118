"""Host side of the SEED-LLaMA forward on MI355X: weight packing, static KV cache, one C-ABI call per step.

Mirrors ``LlamaForCausalLM.forward`` (models/llama_xformer.py:661-743) in eval mode with ``use_cache=True``.
Consumes the HF state-dict key names and repacks once:

* q/k/v_proj stacked into one [3h,h] GEMM; gate/up_proj row-interleaved into one [2F,h] GEMM whose epilogue
  applies SiLU(gate)*up (llama_xformer.py:186);
* RoPE cos/sin tables built exactly like ``LlamaRotaryEmbedding`` (fp32, cat(freqs,freqs), cast to the
  activation dtype on read — llama_xformer.py:118-150);
* a static KV cache [B][H][tmax][128] per layer replaces the reference's ``torch.cat`` per step (:234-239):
  keys are stored post-RoPE, like the reference's ``past_key_value``.

The reference's eval-mode mask quirk means parity is defined for unpadded, equal-length
batches; this engine implements causal attention over positions ``past_len .. past_len+T-1`` for all rows.
"""
import ctypes as C
from typing import Dict, Optional

import torch

from . import lib as L
from .config import LlamaConfig


def _round_up(x, m):
    return (x + m - 1) // m * m


class LlamaEngine:
    def __init__(self, state_dict: Dict[str, torch.Tensor], cfg: LlamaConfig, device="cuda", batch_cap: int = 32,
                 tmax: Optional[int] = None, decode_packed: bool = True, fold_norm: bool = True, dtype: torch.dtype = torch.bfloat16):
        # dtype: the model dtype (weights, activations, KV cache, logits): torch.bfloat16 (BASELINE.json; libseedmi.so) or torch.float16
        # (the reference's `torch_dtype: fp16`, configs/llm/seed_llama_8b.yaml:4; libseedmi_f16.so - same kernels, fp16 as the element)
        if dtype not in (torch.bfloat16, torch.float16):
            raise L.SeedmiError(f"LlamaEngine computes in bfloat16 or float16, not {dtype}")
        self.dtype = dtype
        self.lib = L.load(dtype)
        self.cfg = cfg
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise L.SeedmiError("LlamaEngine needs a HIP device (cuda:N); there is no CPU path")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        if cfg.head_dim != 128:
            raise L.SeedmiError(f"head_dim {cfg.head_dim}: the attention kernels are built for 128")
        with torch.cuda.device(self.device):
            L.check(self.lib.seedmi_check_device(), "seedmi_check_device", self.lib)
        self.batch_cap = batch_cap
        self.decode_packed = decode_packed     # second, fragment-major copy of every weight for the M <= 64 path
        # the fragment-major qkv / gate_up / lm_head copies carry weight * gamma of the RMSNorm in front of them: decode steps
        # then need no norm launches (the GEMM derives the row scale from the activations it streams)
        self.fold_norm = bool(fold_norm and decode_packed)
        self.tmax = tmax or cfg.max_pos
        self.vocab_pad = _round_up(cfg.vocab, 16)
        self._keep = []
        self._ws = None
        self._ws_key = (0, 0)
        self.past_len = 0
        self.cache_generation = 0              # bumped by resize_cache: consumers that baked cache addresses in compare it
        self._pack(state_dict)

    def _dev(self, t):
        t = t.detach().to(device=self.device, dtype=self.dtype).contiguous()
        self._keep.append(t)
        return t

    def _packed(self, w: torch.Tensor, gamma: Optional[torch.Tensor] = None):
        """Fragment-major copy for the weight-streaming decode GEMM (288 GB of HBM: the 2x copy is cheap).  With ``gamma``
        (and fold_norm) the copy holds round_bf16(w * gamma): the RMSNorm weight of the norm feeding this projection."""
        if not self.decode_packed:
            return None
        if gamma is not None and self.fold_norm:
            w = (w.float() * gamma.to(self.device).float().unsqueeze(0)).to(self.dtype).contiguous()
        N, K = w.shape
        out = torch.empty(self.lib.seedmi_pack_skinny_weights_bytes(N, K) // 2, dtype=self.dtype, device=self.device)
        with torch.cuda.device(self.device):
            L.check(self.lib.seedmi_pack_skinny_weights(L.ptr(w), K, N, K, L.ptr(out), L.stream_ptr()), "pack_skinny", self.lib)
        self._keep.append(out)
        return out

    def _pack(self, sd):
        cfg = self.cfg
        h, F = cfg.hidden, cfg.ffn
        p = L.ptr
        w = L.LlamaWeights()
        w.hidden, w.layers, w.heads, w.ffn, w.vocab = h, cfg.layers, cfg.heads, F, cfg.vocab
        w.vocab_pad, w.max_pos, w.tmax, w.batch_cap = self.vocab_pad, cfg.max_pos, self.tmax, self.batch_cap
        w.rms_eps = cfg.rms_eps
        w.embed = p(self._dev(sd["model.embed_tokens.weight"]))
        layers = (L.LlamaLayer * cfg.layers)()
        self.k_cache, self.v_cache = [], []
        for i in range(cfg.layers):
            pre = f"model.layers.{i}."
            l = layers[i]
            l.ln1_w = p(self._dev(sd[pre + "input_layernorm.weight"]))
            qkv = self._dev(torch.cat([sd[pre + f"self_attn.{n}_proj.weight"] for n in "qkv"], 0))
            o = self._dev(sd[pre + "self_attn.o_proj.weight"])
            l.qkv_w, l.o_w = p(qkv), p(o)
            l.qkv_wp, l.o_wp = p(self._packed(qkv, sd[pre + "input_layernorm.weight"])), p(self._packed(o))
            l.ln2_w = p(self._dev(sd[pre + "post_attention_layernorm.weight"]))
            gate, up = sd[pre + "mlp.gate_proj.weight"], sd[pre + "mlp.up_proj.weight"]
            gu = self._dev(torch.stack((gate, up), dim=1).reshape(2 * F, h))
            dn = self._dev(sd[pre + "mlp.down_proj.weight"])
            l.gate_up_w, l.down_w = p(gu), p(dn)
            l.gate_up_wp, l.down_wp = p(self._packed(gu, sd[pre + "post_attention_layernorm.weight"])), p(self._packed(dn))
            kc = torch.zeros(self.batch_cap, cfg.heads, self.tmax, cfg.head_dim, dtype=self.dtype, device=self.device)
            vc = torch.zeros_like(kc)
            self.k_cache.append(kc)
            self.v_cache.append(vc)
            l.k_cache, l.v_cache = p(kc), p(vc)
        self._layers = layers
        w.layer = C.cast(layers, C.POINTER(L.LlamaLayer))
        w.norm_w = p(self._dev(sd["model.norm.weight"]))
        lm = torch.zeros(self.vocab_pad, h, dtype=self.dtype)
        lm[:cfg.vocab] = sd["lm_head.weight"].to(self.dtype).cpu()
        lm = self._dev(lm)
        w.lm_head = p(lm)
        w.lm_head_p = p(self._packed(lm, sd["model.norm.weight"]))
        w.norm_folded = 1 if self.fold_norm else 0
        # LlamaRotaryEmbedding.__init__ (llama_xformer.py:118-134)
        hd = cfg.head_dim
        inv_freq = 1.0 / (cfg.rope_base ** (torch.arange(0, hd, 2).float() / hd))
        t = torch.arange(cfg.max_pos, dtype=torch.float32)
        freqs = torch.einsum("i,j->ij", t, inv_freq)
        emb = torch.cat((freqs, freqs), dim=-1)
        w.cos_t, w.sin_t = p(self._dev(emb.cos())), p(self._dev(emb.sin()))
        self.w = w

    def _workspace(self, B, T):
        # sized by what the library asks for THIS shape (a decode step carves regions a prefill does not: B * T alone does not order them)
        nbytes = self.lib.seedmi_llama_workspace_bytes(C.byref(self.w), B, T)
        if self._ws is None or nbytes > self._ws.numel():
            # initialised once at allocation (include/seedmi.h: seedmi_llama_workspace_init): the split-K flag words start at zero, the
            # sticky error word among them is never cleared by a decode step, and the status call recognises the workspace by its tag
            self._ws = self.new_workspace(nbytes)
            self._ws_key = (B, T)
        return self._ws

    def new_workspace(self, nbytes: int) -> torch.Tensor:
        """A llama workspace as the C ABI wants it handed over: allocated, then seedmi_llama_workspace_init (stream-ordered)."""
        ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            L.check(self.lib.seedmi_llama_workspace_init(L.ptr(ws), ws.numel(), L.stream_ptr()), "seedmi_llama_workspace_init", self.lib)
        return ws

    def reset(self):
        self.past_len = 0

    def decode_status(self, B: int, ws: Optional[torch.Tensor] = None):
        """Raise if a split-K decode GEMM of this batch gave up waiting for a partner workgroup since the last check (its tile of the
        logits is then wrong): seedmi_llama_decode_status reads the sticky error word of the decode workspace.  Synchronises the
        stream - called once at the END of a decode loop, never inside one."""
        ws = self._ws if ws is None else ws
        if ws is None:
            return
        with torch.cuda.device(self.device):
            L.check(self.lib.seedmi_llama_decode_status(C.byref(self.w), B, L.ptr(ws), ws.numel(), L.stream_ptr()),
                    "seedmi_llama_decode_status", self.lib)

    def forward(self, input_ids: Optional[torch.Tensor], position_ids: Optional[torch.Tensor] = None,
                past_len: Optional[int] = None, last_only: bool = False, inputs_embeds: Optional[torch.Tensor] = None,
                hidden_states_out: Optional[list] = None) -> torch.Tensor:
        """input_ids int64 [B,T] on device (or ``inputs_embeds`` [B,T,hidden], llama_xformer.py:502-544).  Appends T positions to
        the static cache starting at ``past_len`` (default: continue from the previous call).  Returns bf16 logits
        [B, T or 1, vocab] (a view of a padded buffer).  ``hidden_states_out`` (a list, last_only False) receives the
        layers+1 tensors [B,T,hidden] of ``output_hidden_states=True`` (:569-570, 613-617)."""
        cfg = self.cfg
        if (input_ids is None) == (inputs_embeds is None):
            raise ValueError("You have to specify exactly one of input_ids and inputs_embeds")       # llama_xformer.py:515-522
        if input_ids is not None:
            if input_ids.dim() != 2:
                raise ValueError("You have to specify input_ids of shape [batch, seq]")
            B, T = input_ids.shape
            input_ids = input_ids.to(device=self.device, dtype=torch.int64).contiguous()
        else:
            if inputs_embeds.dim() != 3 or inputs_embeds.shape[2] != cfg.hidden:
                raise ValueError(f"inputs_embeds must be [batch, seq, {cfg.hidden}]")
            B, T = inputs_embeds.shape[:2]
            inputs_embeds = inputs_embeds.to(device=self.device, dtype=self.dtype).contiguous()
        if past_len is None:
            past_len = self.past_len
        if position_ids is None:                                                          # llama_xformer.py:531-539
            position_ids = torch.arange(past_len, past_len + T, dtype=torch.int64, device=self.device).unsqueeze(0).expand(B, T)
        position_ids = position_ids.reshape(B, T).to(device=self.device, dtype=torch.int64).contiguous()
        Tout = 1 if last_only else T
        logits = torch.empty(B * Tout, self.vocab_pad, dtype=self.dtype, device=self.device)
        hidden = None
        if hidden_states_out is not None:
            if last_only:
                raise ValueError("hidden states are produced for all positions: last_only must be False")
            hidden = torch.empty(cfg.layers + 1, B, T, cfg.hidden, dtype=self.dtype, device=self.device)
        ws = self._workspace(B, T)
        with torch.cuda.device(self.device):
            rc = self.lib.seedmi_llama_forward_io(C.byref(self.w), L.ptr(input_ids), L.ptr(inputs_embeds), L.ptr(position_ids),
                                                  B, T, past_len, None, 1 if last_only else 0, L.ptr(logits), self.vocab_pad,
                                                  L.ptr(hidden), L.ptr(ws), ws.numel(), L.stream_ptr())
        L.check(rc, "seedmi_llama_forward", self.lib)
        self.past_len = past_len + T
        if hidden is not None:
            hidden_states_out.extend(hidden[i] for i in range(cfg.layers + 1))
        return logits.view(B, Tout, self.vocab_pad)[:, :, :cfg.vocab]

    def resize_cache(self, batch_cap: Optional[int] = None, tmax: Optional[int] = None):
        """Grow (or shrink) the static KV cache in place of a rebuild: the packed weights stay where they are, the cached
        positions that fit are carried over.  The cache tensors are REPLACED: everything that baked their addresses in (captured decode
        graphs, ContinuousBatcher slot structs) is tied to ``cache_generation`` and refuses / rebuilds itself after a resize."""
        cfg = self.cfg
        nb, nt = batch_cap or self.batch_cap, tmax or self.tmax
        if (nb, nt) == (self.batch_cap, self.tmax):
            return self
        cb, ct = min(nb, self.batch_cap), min(nt, self.tmax)
        for i in range(cfg.layers):
            for name, caches in (("k_cache", self.k_cache), ("v_cache", self.v_cache)):
                new = torch.zeros(nb, cfg.heads, nt, cfg.head_dim, dtype=self.dtype, device=self.device)
                new[:cb, :, :ct] = caches[i][:cb, :, :ct]
                caches[i] = new
                setattr(self._layers[i], name, L.ptr(new))
        self.batch_cap, self.tmax = nb, nt
        self.w.batch_cap, self.w.tmax = nb, nt
        self.past_len = min(self.past_len, nt)
        self._ws, self._ws_key = None, (0, 0)
        self.cache_generation += 1
        return self

    # ------------------------------------------------------------------ hipGraph decode loop
    def _decode_step_graphed(self, tok: torch.Tensor, logits: torch.Tensor, counter: torch.Tensor, ws: torch.Tensor):
        """One single-token step whose launch arguments never change: the cache length is read from ``counter``
        (device int32) by the RoPE/KV-append and attention kernels, then advanced on the stream."""
        B = tok.shape[0]
        rc = self.lib.seedmi_llama_forward_ex(C.byref(self.w), L.ptr(tok), None, B, 1, self.tmax - 1, L.ptr(counter), 1,
                                              L.ptr(logits), self.vocab_pad, L.ptr(ws), ws.numel(), L.stream_ptr())
        L.check(rc, "seedmi_llama_forward_ex", self.lib)
        L.check(self.lib.seedmi_add_i32(L.ptr(counter), 1, L.stream_ptr()), "seedmi_add_i32", self.lib)

    def select_token(self, logits: torch.Tensor, tok_out: torch.Tensor, top_p: float = 0.0, temperature: float = 1.0,
                     uniforms: Optional[torch.Tensor] = None, step_dev: Optional[torch.Tensor] = None, step_offset: int = 0,
                     history: Optional[torch.Tensor] = None):
        """seedmi_sample_token_bf16 on logits [B, ld] (bf16): greedy (uniforms None / top_p 0) or temperature + top-p with
        the uniform for (step, row) taken from ``uniforms`` [steps, B]; writes tok_out [B] and history[:, step]."""
        B = logits.shape[0]
        if logits.dtype != self.dtype or logits.stride(-1) != 1 or tok_out.dtype != torch.int64:
            raise ValueError("select_token: logits must be bf16 rows, tok_out int64")
        # rows of ``uniforms`` / columns of ``history`` the device-side step may address (0 = no step-indexed buffer)
        n_steps = 0
        if history is not None:
            n_steps = history.shape[1]
        if uniforms is not None:
            n_steps = uniforms.shape[0] if n_steps == 0 else min(n_steps, uniforms.shape[0])
        with torch.cuda.device(self.device):
            rc = self.lib.seedmi_sample_token_bf16(L.ptr(logits), logits.stride(0), B, self.cfg.vocab, float(temperature),
                                                   float(top_p), L.ptr(uniforms), L.ptr(step_dev), int(step_offset),
                                                   L.ptr(tok_out), L.ptr(history), 0 if history is None else history.stride(0),
                                                   n_steps, L.stream_ptr())
        L.check(rc, "seedmi_sample_token_bf16", self.lib)

    def capture_decode_graph(self, first_tok: torch.Tensor, n_new: int, top_p: float = 0.0, temperature: float = 1.0,
                             uniforms: Optional[torch.Tensor] = None):
        """Capture one single-token step (forward, token selection, bookkeeping — all seedmi kernels) as a hipGraph,
        starting from the engine's current cache length.  Returns (replay, out): ``replay(k)`` runs k further steps
        (stream ordered, no host sync) and ``out`` [B, n_new] receives the tokens (column 0 = first_tok).  Greedy by
        default; with ``uniforms`` [n_new, B] (fp32 in [0,1), row s drives column s) the step samples top-p."""
        B = first_tok.shape[0]
        if self.past_len + n_new - 1 > self.tmax:
            raise L.SeedmiError(f"decode would exceed the KV cache ({self.past_len}+{n_new - 1} > {self.tmax})")
        if uniforms is not None and (tuple(uniforms.shape) != (n_new, B) or uniforms.dtype != torch.float32 or
                                     uniforms.device != self.device or not uniforms.is_contiguous()):
            raise ValueError("uniforms must be a contiguous float32 [n_new, B] tensor on the engine's device")
        tok = first_tok.to(torch.int64).reshape(B, 1).contiguous().clone()
        out = torch.zeros(B, n_new, dtype=torch.int64, device=self.device)
        out[:, 0:1] = tok
        past0 = self.past_len
        counter = torch.tensor([past0], dtype=torch.int32, device=self.device)
        logits = torch.empty(B, self.vocab_pad, dtype=self.dtype, device=self.device)
        ws = self._workspace(B, 1)

        def body():
            # after the step the counter reads past0 + s for the s-th new token: column / uniform row s = counter - past0
            self._decode_step_graphed(tok, logits, counter, ws)
            self.select_token(logits, tok, top_p, temperature, uniforms, counter, -past0, out)

        # warm-up on a side stream (required before capture), then rewind its side effects
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        tok0, cnt0, out0 = tok.clone(), counter.clone(), out.clone()
        with torch.cuda.stream(side):
            body()
        torch.cuda.current_stream(self.device).wait_stream(side)
        tok.copy_(tok0)
        counter.copy_(cnt0)
        out.copy_(out0)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            body()
        keep = (tok, counter, logits, ws, uniforms)          # buffers referenced by the graph
        left = [n_new - 1]                                   # steps the capture was sized for (cache rows, uniforms, out)
        generation = self.cache_generation                   # the graph holds raw K/V cache addresses of THIS generation

        def replay(k: int):
            if self.cache_generation != generation:
                raise L.SeedmiError("replay(): the KV cache was resized (resize_cache) after this decode graph was captured; its kernels "
                                    "would write the freed cache - capture a new graph")
            if k < 0 or k > left[0]:
                raise L.SeedmiError(f"replay({k}): only {left[0]} of the {n_new - 1} captured decode steps remain "
                                    "(the graph writes the KV cache, `out` and reads `uniforms` by a device-side counter)")
            for _ in range(k):
                graph.replay()
            left[0] -= k
            self.past_len += k
            return keep and out
        return replay, out

    def greedy_decode_graph(self, prompt_ids: torch.Tensor, n_new: int):
        """Greedy decode with the per-token step captured once as a hipGraph and replayed: ~260 kernel launches per
        step collapse into one graph launch (the decode step is launch/HBM bound; the reference additionally syncs
        the host once per layer per step, llama_xformer.py:255).  Returns tokens [B, n_new]."""
        return self.sample_decode_graph(prompt_ids, n_new, top_p=0.0)

    def sample_decode_graph(self, prompt_ids: torch.Tensor, n_new: int, top_p: float = 0.5, temperature: float = 1.0,
                            generator: Optional[torch.Generator] = None):
        """generate(do_sample=True, top_p, temperature) for the scripts' generation_config
        (scripts/seed_llama_inference_8B.py:81-87) entirely on the device: prefill, then the captured step
        (forward + top-p draw) replayed n_new - 1 times.  top_p == 0 selects greedily.  The uniforms are drawn up front
        from ``generator`` (device generator; default = torch's global one), one per (step, row)."""
        self.reset()
        B = prompt_ids.shape[0]
        uniforms = None
        if top_p > 0.0:
            uniforms = torch.rand(n_new, B, dtype=torch.float32, device=self.device, generator=generator)
        logits0 = self.forward(prompt_ids, last_only=True)
        tok = torch.empty(B, dtype=torch.int64, device=self.device)
        self.select_token(logits0[:, 0], tok, top_p, temperature, uniforms, None, 0, None)
        tok = tok.view(B, 1)
        if n_new == 1:
            return tok
        replay, out = self.capture_decode_graph(tok, n_new, top_p, temperature, uniforms)
        replay(n_new - 1)
        self.decode_status(B)
        return out

    def greedy_decode(self, prompt_ids: torch.Tensor, n_new: int):
        """Greedy loop (argmax on device, no host sync inside the loop). Returns (tokens [B,n_new], per-step logits)."""
        self.reset()
        logits = self.forward(prompt_ids, last_only=True)
        steps = [logits[:, 0]]
        tok = logits[:, 0].float().argmax(-1, keepdim=True)
        out = [tok]
        for _ in range(n_new - 1):
            logits = self.forward(tok, last_only=True)
            steps.append(logits[:, 0])
            tok = logits[:, 0].float().argmax(-1, keepdim=True)
            out.append(tok)
        self.decode_status(prompt_ids.shape[0])
        return torch.cat(out, dim=1), torch.stack(steps, dim=1)
