"""Serving shell around the hot path: the request/response contract of the reference's Flask
``/generate`` endpoint (gradio_demo/seed_llama_flask.py:93-226) on top of the device engines.

Same JSON keys, defaults, assertion and ``error_msg`` texts as the reference handler; what changes is underneath:

* raw images go PIL -> ``DevicePreprocessor`` (bit-exact with the reference's transform) -> ``TokenizerEngine.encode``;
* the prompt is spliced by id arithmetic (``32000 + code`` between ``<img>`` / ``</img>``) instead of formatting
  ``<img_XXXXX>`` strings and re-tokenising them (:149-164);
* ``model.generate(do_sample=True, top_p, temperature)`` (:166-174) runs on the continuous-batching loop of seed_amd/batching.py:
  the request is prefilled into a free slot of the static KV cache and decoded together with whatever else is in flight by a
  hipGraph-captured step (forward over all slots + top-p draw + bookkeeping) that is captured ONCE per sampling configuration and
  replayed in chunks; a request stops at its first EOS (kept, like generate()) or at the context limit, whichever comes first.
  ``handle_many`` serves several requests concurrently (the reference's Flask dev server handles one at a time);
* generated image spans are turned into unCLIP embeds on the device (``seedmi_detokenize``); rendering pixels needs the
  diffusers pipeline, which is not part of this library: pass ``image_renderer`` (embeds -> PIL.Image) to get base64 PNGs,
  otherwise the slot stays '' as it does in the reference when decoding fails.

Transport is left to the caller: ``GenerateService.handle(dict) -> dict`` is the whole handler; ``create_app`` wraps it in a
FastAPI route when fastapi is installed (the reference uses Flask's dev server, one request at a time; so does this).
"""
import base64
import io
from collections import OrderedDict
from typing import Callable, List, Optional

import torch

BOI_TOKEN = '<img>'
EOI_TOKEN = '</img>'
IMG_FLAG = '<image>'
NUM_IMG_TOKNES = 32
NUM_IMG_CODES = 8192
IMAGE_ID_SHIFT = 32000


def decode_image(encoded_image: str):
    from PIL import Image
    return Image.open(io.BytesIO(base64.b64decode(encoded_image.encode('utf-8'))))


def encode_image(image, format: str = 'PNG') -> str:
    with io.BytesIO() as buffer:
        image.save(buffer, format=format)
        return base64.b64encode(buffer.getvalue()).decode('utf-8')


class GenerateService:
    """text_tokenizer needs ``bos_token_id``, ``eos_token_id``, ``eos_token`` (str), ``encode(str) -> List[int]`` (no special
    tokens added) and ``decode(List[int]) -> str`` (special tokens kept: BOI / EOI ids decode to '<img>' / '</img>')."""

    def __init__(self, text_tokenizer, encode_images: Callable[[torch.Tensor], torch.Tensor], llama_engine,
                 preprocess: Callable, boi_token_id: int = IMAGE_ID_SHIFT + NUM_IMG_CODES,
                 eoi_token_id: int = IMAGE_ID_SHIFT + NUM_IMG_CODES + 1, codebook_entry: Optional[Callable] = None,
                 image_renderer: Optional[Callable] = None, device="cuda", batcher_factory: Optional[Callable] = None):
        self.tok = text_tokenizer
        self.encode_images = encode_images          # [B,3,S,S] device tensor -> int64 [B,32]
        self.llm = llama_engine
        self.preprocess = preprocess                # PIL.Image -> [3,S,S] device tensor
        self.boi_token_id, self.eoi_token_id = boi_token_id, eoi_token_id
        self.codebook_entry = codebook_entry        # int64 [B,32] -> image embeds (seedmi_detokenize)
        self.image_renderer = image_renderer        # embeds [1,D] -> PIL.Image (the diffusers pipeline, external)
        self.image_id_shift = IMAGE_ID_SHIFT
        self.device = device
        self._batchers = OrderedDict()              # (top_p, temperature) -> ContinuousBatcher (the captured step bakes them in), LRU
        self.max_batchers = 4                       # each holds a logits buffer, a decode workspace and a captured hipGraph: bounded
        self.decode_chunk = 16
        self._batcher_factory = batcher_factory     # (top_p, temperature) -> object with submit(ids, max_new) / run() (tests)

    def _batcher(self, top_p: float, temperature: float):
        if self._batcher_factory is None:
            from .batching import ContinuousBatcher
        # client-supplied floats key device-side state: quantise them (1e-3 is far below any audible difference in sampling) and keep
        # at most ``max_batchers`` configurations alive, evicting the least recently used IDLE one (its graph and buffers are freed).
        # The degenerate check comes AFTER the quantisation (a temperature in (0, 0.0005) used to round to a key of 0.0,
        # which seedmi_sample_token_bf16 rejects only once a KV slot had been taken): degenerate requests are greedy
        # (temperature -> 0 limit; top_p -> 0 keeps the single most likely token)
        top_p = 0.0 if top_p is None else round(float(top_p), 3)
        temperature = 0.0 if temperature is None else round(float(temperature), 3)
        if not (temperature > 0.0) or not (top_p > 0.0):
            top_p, temperature = 0.0, 1.0
        key = (top_p, temperature)
        if key in self._batchers:
            self._batchers.move_to_end(key)
            return self._batchers[key]
        while len(self._batchers) >= self.max_batchers:
            victim = next((k for k, cb in self._batchers.items() if getattr(cb, "idle", True)), None)
            if victim is None:
                break                               # every configuration is mid-request: exceed the cap rather than drop work
            del self._batchers[victim]
        if self._batcher_factory is not None:
            self._batchers[key] = self._batcher_factory(*key)
        else:
            self._batchers[key] = ContinuousBatcher(self.llm, chunk=self.decode_chunk, top_p=key[0], temperature=key[1],
                                                    eos_token_id=self.tok.eos_token_id)
        return self._batchers[key]

    # ---- seed_llama_flask.py:96-147: request fields, mixed raw / pre-tokenised images
    def _image_ids(self, image_list) -> torch.Tensor:
        tensors, tensor_idx, ids_list, ids_idx = [], [], [], []
        for idx, item in enumerate(image_list):
            if isinstance(item, str):
                tensors.append(self.preprocess(decode_image(item)))
                tensor_idx.append(idx)
            else:
                ids_list.append(item)
                ids_idx.append(idx)
        if tensors:
            ids_1 = self.encode_images(torch.stack(tensors, dim=0)).cpu()
            num_image_ids = ids_1.shape[-1]
        else:
            num_image_ids = len(ids_list[-1])
        images_ids = torch.zeros((len(image_list), num_image_ids), dtype=torch.long)
        if tensor_idx:
            images_ids[tensor_idx, :] = ids_1
        if ids_idx:
            images_ids[ids_idx, :] = torch.tensor(ids_list, dtype=torch.long)
        return images_ids

    def build_prompt(self, text_list: List[str], images_ids: Optional[torch.Tensor], force_boi: bool) -> List[int]:
        """:149-164 by id arithmetic: BOS, text_0, <img> 32000+ids_0 </img>, text_1, ..., text_n [, <img>]."""
        ids = [self.tok.bos_token_id]
        for i, text in enumerate(text_list):
            ids += self.tok.encode(text) if text else []
            if images_ids is not None and i < images_ids.shape[0]:
                ids += [self.boi_token_id] + [self.image_id_shift + int(c) for c in images_ids[i].view(-1).tolist()] + \
                       [self.eoi_token_id]
        if force_boi:
            ids.append(self.boi_token_id)
        return ids

    def _parse_request(self, request_info: dict):
        text_list = request_info['text'].split(IMG_FLAG)
        image_list = request_info['images']
        temperature = request_info.get('temperature', 0.7)
        num_beams = request_info.get('num_beams', 1)
        max_new_tokens = request_info.get('max_new_tokens', 256)
        top_p = request_info.get('top_p', 0.5)
        force_boi = request_info.get('force_boi', False)
        assert len(text_list) == len(image_list) + 1
        if num_beams != 1:
            raise ValueError("num_beams > 1 is not supported by the on-device sampler (the reference demo always sends 1)")
        images_ids = self._image_ids(image_list) if len(image_list) > 0 else None
        images_ids_list = images_ids.tolist() if images_ids is not None else []
        input_ids = self.build_prompt(text_list, images_ids, force_boi)
        return dict(input_ids=input_ids, images_ids_list=images_ids_list, force_boi=force_boi, top_p=top_p, temperature=temperature,
                    max_new_tokens=max_new_tokens)

    def handle(self, request_info: dict) -> dict:
        """One request, the Flask handler's contract: a malformed body fails the reference's own assert (seed_llama_flask.py:107)."""
        return self.handle_many([request_info], raise_asserts=True)[0]

    def handle_many(self, requests: List[dict], raise_asserts: bool = False) -> List[dict]:
        """Serve several /generate requests concurrently: requests with the same sampling configuration share one decode loop.
        A request that cannot be served (malformed, prompt beyond the context) gets an ``error_msg`` reply; its neighbours are served."""
        # every request is parsed and checked BEFORE anything is queued: a bad request gets the reference-style error_msg reply and
        # leaves nothing behind in a decode loop (a request queued before a later one raised would be decoded and discarded next call)
        parsed, errors = [], {}
        for i, r in enumerate(requests):
            try:
                q = self._parse_request(r)
                tmax = getattr(self.llm, "tmax", None)
                if tmax is not None and len(q['input_ids']) > tmax:
                    raise ValueError(f"prompt of {len(q['input_ids'])} tokens exceeds the context of {tmax}")
                if int(q['max_new_tokens']) < 1:
                    raise ValueError("max_new_tokens must be at least 1")
            except Exception as e:        # incl. OSError / PIL.UnidentifiedImageError of a bad image payload, RuntimeError of the encoder
                if raise_asserts and isinstance(e, AssertionError):
                    raise
                q = None
                errors[i] = f'{type(e).__name__}: {e}' if str(e) else type(e).__name__
            parsed.append(q)
        tickets = {}
        try:
            for i, q in enumerate(parsed):
                if q is not None:
                    cb = self._batcher(q['top_p'], q['temperature'])
                    tickets[i] = (cb, cb.submit(q['input_ids'], q['max_new_tokens']))
        except Exception:
            for cb, rid in tickets.values():          # nothing of a failed call stays queued
                if hasattr(cb, "cancel"):
                    cb.cancel(rid)
            raise
        results = {}
        batchers = {id(cb): cb for cb, _ in tickets.values()}
        try:
            for cb in batchers.values():
                results[id(cb)] = cb.run()
        except Exception:
            # a decode loop failed midway: nothing of this call may stay behind in the batchers that did not finish - not the requests
            # still waiting, not the ones already prefilled into slots, not finished results run() never handed out (run() raised before
            # swapping `done` out, or decode_status raised at its end) - or the next call would decode orphaned slots and return stale ids.
            # Such a batcher is aborted and dropped from the cache: the next request of that sampling configuration builds a clean one.
            for i, (cb, rid) in tickets.items():
                if id(cb) not in results and hasattr(cb, "cancel"):
                    cb.cancel(rid)
            for cb in batchers.values():
                if id(cb) not in results:
                    if hasattr(cb, "abort"):
                        cb.abort()
                    for k in [k for k, v in self._batchers.items() if v is cb]:
                        del self._batchers[k]
            raise
        out = []
        for i, q in enumerate(parsed):
            if q is None:
                out.append({'text': '', 'images': [], 'images_ids': [], 'error_msg': [errors[i]]})
            else:
                cb, rid = tickets[i]
                out.append(self._finish(q, torch.tensor(results[id(cb)][rid], dtype=torch.int64)))
        return out

    def _finish(self, q: dict, generate_ids: torch.Tensor) -> dict:
        images_ids_list = q['images_ids_list']
        if q['force_boi']:                                              # :177-178 the forced <img> counts as generated
            generate_ids = torch.cat((torch.tensor([self.boi_token_id], dtype=torch.int64), generate_ids))

        boi_indices = torch.where(generate_ids == self.boi_token_id)[0].tolist()
        eoi_indices = torch.where(generate_ids == self.eoi_token_id)[0].tolist()
        generated_image_base64_list = []
        text_mask = torch.ones_like(generate_ids, dtype=torch.bool)
        error_msg = []
        if len(boi_indices) != len(eoi_indices):
            error_msg.append(
                f'Num of BOI (begain of image) tokens: {len(boi_indices)} is not equal to EOI(end of image tokens): {len(eoi_indices)}, some image Some images will fail to decode.'
            )
        num_images = min(len(boi_indices), len(eoi_indices))
        for idx in range(num_images):
            boi_index, eoi_index = boi_indices[idx], eoi_indices[idx]
            image_ids = generate_ids[boi_index + 1:eoi_index].unsqueeze(0) - self.image_id_shift
            if image_ids.shape[-1] != NUM_IMG_TOKNES:
                error_msg.append(f'Len(image_ids) {image_ids.shape[-1]} is not equal to {NUM_IMG_TOKNES}')
                image_base64 = ''
            elif (image_ids < 0).any() or (image_ids >= NUM_IMG_CODES).any():
                error_msg.append(f'Some image_id out of range: [0, {NUM_IMG_CODES})')
                image_base64 = ''
            else:
                image_base64 = ''
                if self.codebook_entry is not None and self.image_renderer is not None:
                    embeds = self.codebook_entry(image_ids.to(self.device))
                    image_base64 = encode_image(self.image_renderer(embeds))
            generated_image_base64_list.append(image_base64)
            text_mask[boi_index + 1:eoi_index] = False
            images_ids_list.append(image_ids.view(-1).tolist())
        generate_ids = generate_ids[text_mask]

        generate_text = self.tok.decode(generate_ids.tolist())
        generate_text = generate_text.replace(BOI_TOKEN + ' ' + EOI_TOKEN + ' ', IMG_FLAG)
        generate_text = generate_text.replace(self.tok.eos_token, '')
        return {'text': generate_text, 'images': generated_image_base64_list, 'images_ids': images_ids_list,
                'error_msg': error_msg}


def create_app(service: GenerateService):
    """POST/GET /generate with the reference's JSON body (seed_llama_flask.py:93-96)."""
    from fastapi import FastAPI, Request
    app = FastAPI()

    async def generate(request: Request):
        return service.handle(await request.json())

    app.add_api_route('/generate', generate, methods=['GET', 'POST'])
    return app
