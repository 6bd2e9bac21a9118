"""The /generate contract of gradio_demo/seed_llama_flask.py:93-226 on seed_amd/serve.py, with stub engines on CPU
(request parsing, mixed raw / pre-tokenised images, id-arithmetic prompt, span parsing, error_msg conventions) and, on the
GPU, end to end through the real engines at TINY size."""
import base64
import io

import numpy as np
import pytest
import torch

from seed_amd import serve

BOI, EOI, SHIFT = 32000 + 8192, 32000 + 8193, 32000


class CharTokenizer:
    """One id per character (3 + ord): enough to check where text and image spans land."""
    bos_token_id, eos_token_id, eos_token = 1, 2, '</s>'

    def encode(self, text):
        return [3 + ord(c) for c in text]

    def decode(self, ids):
        out = []
        for i in ids:
            out.append({BOI: '<img> ', EOI: '</img> ', 2: '</s>', 1: '<s>'}.get(i, chr(i - 3) if 3 <= i < 3 + 0x110000 else '?'))
        return ''.join(out)


class ScriptedLLM:
    """Stands in for the decode loop (seed_amd.batching.ContinuousBatcher's submit / run protocol): replays a fixed token script and
    stops at EOS or the budget, like the real loop."""

    def __init__(self, script):
        self.script, self.calls = script, []

    def batcher(self, top_p, temperature):
        llm = self

        class B:
            def __init__(self):
                self.q = []

            def submit(self, ids, max_new):
                llm.calls.append(dict(prompt=torch.tensor([list(ids)]), n_new=max_new, top_p=top_p, temperature=temperature))
                self.q.append(max_new)
                return len(self.q) - 1

            def run(self):
                out = {}
                for rid, n in enumerate(self.q):
                    toks = (list(llm.script) + [2] * n)[:n]
                    if 2 in toks:
                        toks = toks[:toks.index(2) + 1]
                    out[rid] = toks
                self.q = []
                return out
        return B()


def _png_b64(h=20, w=30):
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(np.random.RandomState(0).randint(0, 255, (h, w, 3)).astype(np.uint8)).save(buf, format='PNG')
    return base64.b64encode(buf.getvalue()).decode()


def _service(script, **kw):
    enc = lambda batch: torch.arange(32).repeat(batch.shape[0], 1) + 100          # "tokenizer": ids 100..131 per image
    pre = lambda pil: torch.zeros(3, 224, 224)
    llm = ScriptedLLM(script)
    return serve.GenerateService(CharTokenizer(), enc, llm, pre, device="cpu", batcher_factory=llm.batcher, **kw)


def test_prompt_is_spliced_by_id_arithmetic_and_defaults_match_the_reference():
    svc = _service([3 + ord('o'), 3 + ord('k'), 2])
    pre_ids = list(range(7000, 7032))
    out = svc.handle({'text': 'a<image>b<image>c', 'images': [_png_b64(), pre_ids]})
    call = svc.llm.calls[0]
    want = [1, 3 + ord('a'), BOI] + [SHIFT + 100 + i for i in range(32)] + [EOI, 3 + ord('b'), BOI] + \
           [SHIFT + c for c in pre_ids] + [EOI, 3 + ord('c')]
    assert call['prompt'][0].tolist() == want
    assert (call['n_new'], call['top_p'], call['temperature']) == (256, 0.5, 0.7)          # :98-102 defaults
    assert out['text'] == 'ok' and out['images'] == [] and out['error_msg'] == []
    assert out['images_ids'] == [[100 + i for i in range(32)], pre_ids]


def test_text_and_image_count_must_match():
    with pytest.raises(AssertionError):
        _service([2]).handle({'text': 'no flag here', 'images': [list(range(32))]})


def test_generated_image_span_is_parsed_and_masked_out_of_the_text():
    codes = [5 * i for i in range(32)]
    script = [3 + ord('x'), BOI] + [SHIFT + c for c in codes] + [EOI, 3 + ord('y'), 2, 3 + ord('z')]
    out = _service(script).handle({'text': 'hi', 'images': [], 'max_new_tokens': 64})
    assert out['images_ids'] == [codes]
    assert out['images'] == ['']                      # no renderer attached: the slot stays empty, as when decoding fails
    assert out['text'] == 'x' + serve.IMG_FLAG + 'y'  # '<img> </img> ' -> '<image>', EOS dropped, nothing after EOS
    assert out['error_msg'] == []


def test_error_msg_conventions():
    short = [BOI] + [SHIFT + 1] * 5 + [EOI]
    out = _service(short).handle({'text': 'q', 'images': []})
    assert out['error_msg'] == ['Len(image_ids) 5 is not equal to 32'] and out['images'] == ['']
    bad = [BOI] + [SHIFT - 7] + [SHIFT + 1] * 31 + [EOI]
    out = _service(bad).handle({'text': 'q', 'images': []})
    assert out['error_msg'] == ['Some image_id out of range: [0, 8192)']
    unbalanced = [BOI] + [SHIFT + 1] * 32
    out = _service(unbalanced).handle({'text': 'q', 'images': []})
    assert out['error_msg'][0].startswith('Num of BOI (begain of image) tokens: 1 is not equal to EOI(end of image tokens): 0')


def test_force_boi_counts_the_forced_token_as_generated_and_renderer_hook():
    codes = list(range(32))
    script = [SHIFT + c for c in codes] + [EOI, 2]
    from PIL import Image
    svc = _service(script, codebook_entry=lambda ids: torch.zeros(ids.shape[0], 1024),
                   image_renderer=lambda emb: Image.new('RGB', (8, 8), (255, 0, 0)))
    out = svc.handle({'text': 'draw', 'images': [], 'force_boi': True})
    assert svc.llm.calls[0]['prompt'][0, -1].item() == BOI
    assert out['images_ids'] == [codes] and out['error_msg'] == []
    img = serve.decode_image(out['images'][0])
    assert img.size == (8, 8)


def test_bad_requests_get_error_replies_and_leave_nothing_queued():
    """every request is validated before any is queued - a prompt beyond the context or a malformed body in a batch of
    requests yields a reference-style error_msg reply, the other requests are served, and no decode loop keeps a stale ticket."""
    svc = _service([3 + ord('o'), 3 + ord('k'), 2])
    svc.llm.tmax = 40
    good = {'text': 'hi', 'images': []}
    too_long = {'text': 'x' * 100, 'images': []}
    malformed = {'text': 'no flag here', 'images': [list(range(32))]}
    outs = svc.handle_many([good, too_long, malformed, good])
    assert [o['text'] for o in outs] == ['ok', '', '', 'ok']
    assert 'exceeds the context' in outs[1]['error_msg'][0] and outs[2]['error_msg'][0].startswith('AssertionError')
    assert len(svc.llm.calls) == 2                                   # only the two good requests reached a decode loop
    assert all(not b.q for b in svc._batchers.values())              # and nothing is left queued


def test_sampling_configurations_are_bounded_lru():
    """one decode loop (logits buffer + workspace + captured graph) per sampling configuration, keyed by client floats:
    quantised, and at most ``max_batchers`` kept (least recently used idle one evicted)."""
    svc = _service([2])
    svc.max_batchers = 3
    for k in range(10):
        svc.handle({'text': 'q', 'images': [], 'temperature': 0.5 + 0.01 * k, 'top_p': 0.9})
        assert len(svc._batchers) <= 3
    a = svc._batcher(0.9, 0.70004)
    assert svc._batcher(0.9, 0.70001) is a                            # quantised to 1e-3
    svc._batcher(0.9, 0.1); svc._batcher(0.9, 0.2)
    assert svc._batcher(0.9, 0.7) is a and len(svc._batchers) == 3    # recently used entries survive


@pytest.mark.gpu
def test_generate_end_to_end_on_device_engines():
    """Raw image -> device pre-processing -> tokenizer engine -> spliced prompt -> sampled graph decode -> parsed reply,
    through the real engines at test size (TINY tokenizer, LLAMA_TINY with the 40194-wide SEED vocabulary)."""
    import dataclasses
    from seed_amd import config as C
    from seed_amd.llama_engine import LlamaEngine
    from seed_amd.preprocess import DevicePreprocessor, BILINEAR
    from seed_amd.tokenizer_engine import TokenizerEngine
    from seed_amd.weights import make_llama_state_dict, make_tokenizer_state_dict
    tcfg = C.TINY
    teng = TokenizerEngine(make_tokenizer_state_dict(tcfg, seed=0), tcfg)
    lcfg = dataclasses.replace(C.LLAMA_TINY, vocab=40194)
    leng = LlamaEngine(make_llama_state_dict(lcfg, seed=1), lcfg, device="cuda", batch_cap=4, tmax=256)
    pre = DevicePreprocessor(tcfg.img_size, interpolation=BILINEAR, keep_ratio=False)
    svc = serve.GenerateService(CharTokenizer(), teng.encode, leng, pre)
    out = svc.handle({'text': 'look<image>what is it?', 'images': [_png_b64(40, 50)], 'max_new_tokens': 24, 'top_p': 0.5})
    assert set(out) == {'text', 'images', 'images_ids', 'error_msg'}
    assert len(out['images_ids']) >= 1 and len(out['images_ids'][0]) == 32
    assert all(0 <= c < tcfg.n_embed for c in out['images_ids'][0])
    assert isinstance(out['text'], str)
    # several requests at once share the decode loop; greedy replies (top_p 0 / temperature 0) do not depend on the company they keep
    reqs = [{'text': 'a' * (3 + i), 'images': [], 'max_new_tokens': 6 + i, 'top_p': 0.0} for i in range(3)]
    together = svc.handle_many(reqs)
    alone = [svc.handle(r) for r in reqs]
    assert [o['text'] for o in together] == [o['text'] for o in alone]
    assert svc.handle({'text': 'aaa', 'images': [], 'max_new_tokens': 6, 'temperature': 0.0})['text'] == alone[0]['text']


def test_tiny_temperature_is_greedy_and_a_bad_image_payload_is_an_error_reply():
    """a temperature in (0, 0.0005) rounds to 0.0 - the key must then be the greedy one (the device sampler rejects
    temperature 0 only after a KV slot was taken); an undecodable image is an error_msg reply for THAT request, its neighbour is served."""
    svc = _service([3 + ord('o'), 2])
    svc.handle({'text': 'a', 'images': [], 'temperature': 0.0004, 'top_p': 0.5})
    assert (svc.llm.calls[-1]['top_p'], svc.llm.calls[-1]['temperature']) == (0.0, 1.0)
    svc.handle({'text': 'a', 'images': [], 'temperature': 0.7, 'top_p': 0.0004})
    assert (svc.llm.calls[-1]['top_p'], svc.llm.calls[-1]['temperature']) == (0.0, 1.0)
    svc.handle({'text': 'a', 'images': [], 'temperature': 0.7004, 'top_p': 0.5})
    assert (svc.llm.calls[-1]['top_p'], svc.llm.calls[-1]['temperature']) == (0.5, 0.7)
    bad = base64.b64encode(b'not an image at all').decode()
    outs = svc.handle_many([{'text': 'x<image>y', 'images': [bad]}, {'text': 'fine', 'images': []}])
    assert outs[0]['error_msg'] and outs[0]['text'] == ''
    assert outs[1]['text'] == 'o' and not outs[1]['error_msg']


def test_a_failing_decode_loop_leaves_nothing_queued():
    """if one batcher's run() raises, the tickets of the batchers that have not run yet are cancelled."""
    class Boom:
        def __init__(self, fail):
            self.fail, self.q, self.cancelled = fail, [], []

        def submit(self, ids, max_new):
            self.q.append(max_new)
            return len(self.q) - 1

        def cancel(self, rid):
            self.cancelled.append(rid)
            return True

        def run(self):
            if self.fail:
                raise RuntimeError("decode loop died")
            return {i: [2] for i in range(len(self.q))}
    made = []

    def factory(top_p, temperature):
        made.append(Boom(fail=len(made) == 0))
        return made[-1]
    svc = _service([2])
    svc._batcher_factory = factory
    with pytest.raises(RuntimeError, match="decode loop died"):
        svc.handle_many([{'text': 'a', 'images': [], 'top_p': 0.5}, {'text': 'b', 'images': [], 'top_p': 0.9}])
    assert made[0].cancelled == [0] and made[1].cancelled == [0]


def test_a_failing_decode_loop_aborts_and_drops_its_batcher():
    """requests of a failed call that were already prefilled into slots (or finished but never handed out) must not survive in
    the batcher: it is aborted and dropped from the cache, so the next call of that sampling configuration starts from a clean one."""
    class Boom:
        def __init__(self, fail):
            self.fail, self.q, self.aborted = fail, [], 0
            self.active, self.done = {}, {}

        def submit(self, ids, max_new):
            self.q.append(max_new)
            return len(self.q) - 1

        def cancel(self, rid):
            return False                                   # (already prefilled into a slot: not cancellable)

        def abort(self):
            self.aborted += 1
            self.active, self.done = {}, {}

        def run(self):
            if self.fail:
                self.active = {0: {"id": 0}}               # what a loop that died midway leaves behind
                self.done = {0: [7, 7, 7]}
                raise RuntimeError("decode loop died")
            return {i: [2] for i in range(len(self.q))}
    made = []

    def factory(top_p, temperature):
        made.append(Boom(fail=len(made) == 0))
        return made[-1]
    svc = _service([2])
    svc._batcher_factory = factory
    with pytest.raises(RuntimeError, match="decode loop died"):
        svc.handle_many([{'text': 'a', 'images': [], 'top_p': 0.5}])
    assert made[0].aborted == 1 and not made[0].active and not made[0].done
    assert all(cb is not made[0] for cb in svc._batchers.values())
    out = svc.handle_many([{'text': 'a', 'images': [], 'top_p': 0.5}])      # same configuration: a fresh batcher, no stale ids
    assert len(made) == 2 and made[1].aborted == 0 and out[0]['text'] == '' and not out[0]['error_msg']      # ([2] = EOS only)
