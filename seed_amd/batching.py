"""Continuous-batching decode loop on the static KV cache.

The reference's Flask handler runs one ``model.generate`` per request (gradio_demo/seed_llama_flask.py:166-174, Flask dev server:
one request at a time).  Here the KV cache's batch rows are SLOTS: a request is prefilled into a free slot (its own cache row,
positions 0..T0-1), after which all slots advance together, one token per step, each at its own cache length
(``seedmi_llama_decode_slots``).  The step - forward over all slots, token selection, length bookkeeping - is captured once as a
hipGraph and replayed ``chunk`` times between host visits; the host then reads the chunk's tokens, retires rows that hit EOS or
their budget and admits waiting requests into the freed slots.  Rows are independent in every kernel of the step, so a request's
tokens do not depend on what its neighbours are doing: greedy output equals the request run alone (tests/test_batching.py).
"""
import ctypes as C
from collections import deque
from typing import Dict, List, Optional

import torch

from . import lib as L


class ContinuousBatcher:
    def __init__(self, engine, slots: Optional[int] = None, chunk: int = 8, top_p: float = 0.0, temperature: float = 1.0,
                 eos_token_id: Optional[int] = None, generator: Optional[torch.Generator] = None):
        self.eng = engine
        self.lib = engine.lib
        self.B = slots or engine.batch_cap
        if self.B > engine.batch_cap:
            raise ValueError(f"{self.B} slots but the engine's KV cache holds {engine.batch_cap} rows")
        self.chunk, self.top_p, self.temperature = int(chunk), float(top_p), float(temperature)
        self.eos = eos_token_id
        self.gen = generator
        dev = engine.device
        self.tok = torch.zeros(self.B, dtype=torch.int64, device=dev)            # current token of every slot
        self.lens = torch.zeros(self.B, dtype=torch.int32, device=dev)           # cache length of every slot
        self.inc = torch.zeros(self.B, dtype=torch.int32, device=dev)            # 1 for active slots
        self.step = torch.zeros(1, dtype=torch.int32, device=dev)                # column of `hist` / row of `uniforms` within the chunk
        self.hist = torch.zeros(self.B, self.chunk, dtype=torch.int64, device=dev)
        self.uniforms = torch.zeros(self.chunk, self.B, dtype=torch.float32, device=dev) if top_p > 0.0 else None
        self.logits = torch.empty(self.B, engine.vocab_pad, dtype=engine.dtype, device=dev)
        # the step's workspace is baked into the captured graph: owned here, never reallocated; prefills use their own
        self._ws = engine.new_workspace(self.lib.seedmi_llama_workspace_bytes(C.byref(engine.w), self.B, 1))
        self._ws_prefill = None
        self._graph = None
        self._slot_w = {}
        self._generation = engine.cache_generation                               # of the cache addresses baked into _graph / _slot_w
        self.free = deque(range(self.B))
        self.waiting = deque()
        self.active: Dict[int, dict] = {}                                        # slot -> request state
        self.done: Dict[int, List[int]] = {}
        self._next_id = 0

    # ------------------------------------------------------------------ the captured step
    def _step_body(self):
        eng = self.eng
        with torch.cuda.device(eng.device):
            L.check(self.lib.seedmi_llama_decode_slots(C.byref(eng.w), L.ptr(self.tok), L.ptr(self.lens), self.B, L.ptr(self.logits),
                                                       eng.vocab_pad, L.ptr(self._ws), self._ws.numel(), L.stream_ptr()),
                    "seedmi_llama_decode_slots")
            eng.select_token(self.logits, self.tok, self.top_p, self.temperature, self.uniforms, self.step, 0, self.hist)
            L.check(self.lib.seedmi_add_i32_vec(L.ptr(self.lens), L.ptr(self.inc), self.B, L.stream_ptr()), "seedmi_add_i32_vec")
            L.check(self.lib.seedmi_add_i32(L.ptr(self.step), 1, L.stream_ptr()), "seedmi_add_i32")

    def _sync_generation(self):
        """``LlamaEngine.resize_cache`` replaces the K/V cache tensors (a larger batch or context arrived elsewhere).  The captured
        step and the per-slot weight structs hold raw addresses of the OLD tensors: drop them and rebuild on demand (the cached
        positions were carried over, so requests in flight continue); refuse if the new cache no longer holds this batcher's slots."""
        eng = self.eng
        if eng.cache_generation == self._generation:
            return
        if self.B > eng.batch_cap:
            raise L.SeedmiError(f"the KV cache was resized to {eng.batch_cap} rows but this batcher serves {self.B} slots")
        for req in self.active.values():
            if req["prompt"].numel() + req["max_new"] - 1 > eng.tmax:
                raise L.SeedmiError("the KV cache was shrunk below the context of a request in flight")
        # requests still waiting were clamped in submit() against the OLD context: clamp again (generate() stops at the context
        # limit); one whose prompt no longer fits finishes empty instead of failing a prefill with a slot in hand
        for req in list(self.waiting):
            room = eng.tmax - req["prompt"].numel() + 1
            if room < 1:
                self.waiting.remove(req)
                self.done[req["id"]] = []
            else:
                req["max_new"] = min(req["max_new"], room)
        self._graph, self._slot_w = None, {}
        self._ws = eng.new_workspace(self.lib.seedmi_llama_workspace_bytes(C.byref(eng.w), self.B, 1))
        self._ws_prefill = None
        self._generation = eng.cache_generation

    def _ensure_graph(self):
        self._sync_generation()
        if self._graph is not None:
            return
        eng = self.eng
        # (the warm-up step writes each slot's K/V row at its current length: exactly what the first replay writes again)
        saved = [t.clone() for t in (self.tok, self.lens, self.step, self.hist)]
        side = torch.cuda.Stream(device=eng.device)
        side.wait_stream(torch.cuda.current_stream(eng.device))
        with torch.cuda.stream(side):
            self._step_body()                                                    # warm-up (required before capture)
        torch.cuda.current_stream(eng.device).wait_stream(side)
        for t, s in zip((self.tok, self.lens, self.step, self.hist), saved):
            t.copy_(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._step_body()
        self._graph = g

    # ------------------------------------------------------------------ slots
    def _slot_weights(self, s: int):
        """The engine's weight struct with every layer's cache pointers moved to row s: a batch-1 prefill through it fills slot s."""
        self._sync_generation()
        if s not in self._slot_w:
            eng, cfg = self.eng, self.eng.cfg
            layers = (L.LlamaLayer * cfg.layers)()
            row_bytes = cfg.heads * eng.tmax * cfg.head_dim * 2
            for i in range(cfg.layers):
                for name, _ in L.LlamaLayer._fields_:
                    setattr(layers[i], name, getattr(eng._layers[i], name))
                layers[i].k_cache = eng.k_cache[i].data_ptr() + s * row_bytes
                layers[i].v_cache = eng.v_cache[i].data_ptr() + s * row_bytes
            w = L.LlamaWeights()
            for name, _ in L.LlamaWeights._fields_:
                setattr(w, name, getattr(eng.w, name))
            w.layer = C.cast(layers, C.POINTER(L.LlamaLayer))
            w.batch_cap = 1
            self._slot_w[s] = (w, layers)
        return self._slot_w[s][0]

    def submit(self, prompt_ids, max_new_tokens: int) -> int:
        """prompt_ids: 1-D int64 token ids (BOS included, as the scripts build them).  Returns the request id."""
        p = torch.as_tensor(prompt_ids, dtype=torch.int64).reshape(-1)
        if p.numel() + max_new_tokens - 1 > self.eng.tmax:
            max_new_tokens = self.eng.tmax - p.numel() + 1                     # the reference's generate() stops at the context limit
        if p.numel() < 1 or max_new_tokens < 1:
            raise ValueError("empty prompt or no room to generate in the KV cache")
        rid = self._next_id
        self._next_id += 1
        self.waiting.append({"id": rid, "prompt": p, "max_new": int(max_new_tokens)})
        return rid

    def cancel(self, rid: int) -> bool:
        """Remove a request that is still waiting (not yet prefilled into a slot)."""
        for req in list(self.waiting):
            if req["id"] == rid:
                self.waiting.remove(req)
                return True
        return False

    def abort(self):
        """Drop EVERYTHING this batcher holds - waiting requests, requests already prefilled into slots, finished results nobody collected -
        and return every slot.  For the caller whose run() raised midway: without it the next run() would keep decoding the
        failed call's orphaned slots and hand their stale ids to whoever asks next."""
        self.waiting.clear()
        for s in list(self.active):
            self.active.pop(s)
            self.inc[s:s + 1].fill_(0)
            self.lens[s:s + 1].fill_(0)
            self.free.append(s)
        self.done = {}

    @property
    def idle(self) -> bool:
        return not self.waiting and not self.active

    def _admit(self):
        self._sync_generation()                                                  # re-clamp waiting requests before any slot is taken
        while self.waiting and self.free:
            req = self.waiting.popleft()
            s = self.free.popleft()
            try:
                self._prefill_into(req, s)
            except Exception:
                # the slot goes back and the request is dropped with an empty result: a failed prefill must not leak the KV row (after
                # batch_cap leaks run() would spin with waiting requests and no free slot) nor poison the requests behind it
                self.free.appendleft(s)
                self.done[req["id"]] = []
                raise

    def _prefill_into(self, req, s: int):
        eng = self.eng
        ids = req["prompt"].to(eng.device).view(1, -1)
        T0 = ids.shape[1]
        pos = torch.arange(T0, dtype=torch.int64, device=eng.device).view(1, T0)
        lg = torch.empty(1, eng.vocab_pad, dtype=eng.dtype, device=eng.device)
        need = self.lib.seedmi_llama_workspace_bytes(C.byref(eng.w), 1, T0)
        if self._ws_prefill is None or self._ws_prefill.numel() < need:
            self._ws_prefill = eng.new_workspace(need)
        ws = self._ws_prefill
        with torch.cuda.device(eng.device):
            L.check(self.lib.seedmi_llama_forward_io(C.byref(self._slot_weights(s)), L.ptr(ids), None, L.ptr(pos), 1, T0, 0, None, 1,
                                                     L.ptr(lg), eng.vocab_pad, None, L.ptr(ws), ws.numel(), L.stream_ptr()),
                    "prefill into a slot")
        first = torch.empty(1, dtype=torch.int64, device=eng.device)
        u = None
        if self.top_p > 0.0:
            u = torch.rand(1, 1, dtype=torch.float32, device=eng.device, generator=self.gen)
        eng.select_token(lg, first, self.top_p, self.temperature, u, None, 0, None)
        self.tok[s:s + 1].copy_(first)
        self.lens[s:s + 1].fill_(T0)
        self.inc[s:s + 1].fill_(1)
        req.update(slot=s, tokens=[], first=first, emitted=0)
        self.active[s] = req

    def _retire(self, s: int):
        req = self.active.pop(s)
        self.done[req["id"]] = req["tokens"]
        self.inc[s:s + 1].fill_(0)
        self.lens[s:s + 1].fill_(0)
        self.free.append(s)

    def _take(self, req, toks: List[int]) -> bool:
        """Append generated tokens to a request; True when it is finished (EOS kept, like generate())."""
        for t in toks:
            req["tokens"].append(int(t))
            if (self.eos is not None and int(t) == self.eos) or len(req["tokens"]) >= req["max_new"]:
                return True
        return False

    # ------------------------------------------------------------------ driver
    def run(self) -> Dict[int, List[int]]:
        """Serve every submitted request to completion; returns {request id: generated token ids}."""
        eng = self.eng
        while self.waiting or self.active:
            self._admit()
            # the token chosen at prefill is the request's first generated token
            for s, req in list(self.active.items()):
                if req["first"] is not None:
                    fin = self._take(req, req["first"].cpu().tolist())
                    req["first"] = None
                    if fin:
                        self._retire(s)
            if not self.active:
                continue
            self._ensure_graph()                                                 # (re-captured if the engine's cache was resized)
            self.step.zero_()
            if self.uniforms is not None:
                self.uniforms.copy_(torch.rand(self.uniforms.shape, dtype=torch.float32, device=eng.device, generator=self.gen))
            for _ in range(self.chunk):
                self._graph.replay()
            hist = self.hist.cpu()                                               # the only host sync: once per chunk
            for s, req in list(self.active.items()):
                if self._take(req, hist[s].tolist()):
                    self._retire(s)
        eng.decode_status(self.B, self._ws)                                       # a cut split-K tile that lost its partner is an error, not a token
        out, self.done = self.done, {}
        return out
