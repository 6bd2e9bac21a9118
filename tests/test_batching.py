"""Continuous-batching decode loop (seed_amd/batching.py) on a real MI355X: more requests than slots, different prompt
lengths and budgets, EOS in the middle of a chunk - every request's greedy tokens must equal the same request decoded alone (rows
of the decode step are independent in every kernel), and the per-slot step itself must equal the shared-length step."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from seed_amd import config as C  # noqa: E402
from seed_amd.batching import ContinuousBatcher  # noqa: E402
from seed_amd.llama_engine import LlamaEngine  # noqa: E402
from seed_amd.weights import make_llama_state_dict  # noqa: E402


def _alone(eng, prompt, n_new, eos=None):
    toks, _ = eng.greedy_decode(prompt.view(1, -1).cuda(), n_new)
    out = toks[0].cpu().tolist()
    if eos is not None and eos in out:
        out = out[:out.index(eos) + 1]
    return out


def test_continuous_batching_equals_requests_run_alone():
    cfg = C.LLAMA_TINY
    sd = make_llama_state_dict(cfg, seed=9, norm_jitter=0.05)
    eng = LlamaEngine(sd, cfg, device="cuda", batch_cap=4, tmax=96)
    g = torch.Generator().manual_seed(21)
    reqs = [(torch.randint(3, cfg.vocab, (T0,), generator=g), n) for T0, n in
            ((5, 7), (17, 20), (9, 1), (30, 12), (3, 33), (12, 9), (25, 16), (8, 5), (40, 11))]
    want = [_alone(eng, p, n) for p, n in reqs]
    # an "EOS" that really occurs: the 4th token request 1 generates on its own
    eos = want[1][3]
    want_eos = [_alone(eng, p, n, eos) for p, n in reqs]
    for slots, chunk, eos_id, expect in ((3, 4, None, want), (4, 8, None, want), (2, 5, eos, want_eos)):
        cb = ContinuousBatcher(eng, slots=slots, chunk=chunk, eos_token_id=eos_id)
        ids = [cb.submit(p, n) for p, n in reqs]
        out = cb.run()
        assert sorted(out) == ids
        for i, rid in enumerate(ids):
            assert out[rid] == expect[i], (slots, chunk, i, out[rid], expect[i])
    # requests submitted while others are in flight join at the next chunk boundary
    cb = ContinuousBatcher(eng, slots=2, chunk=3)
    a = cb.submit(*reqs[1])
    first = cb.run()
    b = cb.submit(*reqs[3])
    c = cb.submit(*reqs[4])
    rest = cb.run()
    assert first[a] == want[1] and rest[b] == want[3] and rest[c] == want[4]


def test_sampled_continuous_batching_is_reproducible_and_in_the_nucleus():
    from oracle import sample_oracle as S
    cfg = C.LLAMA_TINY
    sd = make_llama_state_dict(cfg, seed=9, norm_jitter=0.05)
    eng = LlamaEngine(sd, cfg, device="cuda", batch_cap=3, tmax=64)
    g = torch.Generator().manual_seed(5)
    reqs = [(torch.randint(3, cfg.vocab, (T0,), generator=g), n) for T0, n in ((6, 9), (11, 6), (4, 12), (9, 4))]

    def run(seed):
        cb = ContinuousBatcher(eng, slots=3, chunk=4, top_p=0.5, temperature=1.0,
                               generator=torch.Generator(device="cuda").manual_seed(seed))
        ids = [cb.submit(p, n) for p, n in reqs]
        out = cb.run()
        return [out[i] for i in ids]
    a, b, c = run(1), run(1), run(2)
    assert a == b and a != c
    assert all(len(t) == n for t, (_, n) in zip(a, reqs))
    # every sampled token lies in the top-p nucleus of the logits it was drawn from (teacher-forced through the plain engine)
    for toks, (p, n) in zip(a, reqs):
        eng.reset()
        lg = eng.forward(p.view(1, -1).cuda(), last_only=True)
        for t in toks:
            order, k, _ = S.top_p_keep(lg[0, 0].float().cpu().numpy(), 1.0, 0.5)
            assert t in set(order[:k + 1].tolist())
            lg = eng.forward(torch.tensor([[t]], device="cuda"), last_only=True)


def test_cache_resize_invalidates_baked_addresses():
    """LlamaEngine.resize_cache REPLACES the K/V cache tensors.  A batcher that served requests before the resize (captured
    step + per-slot weight structs hold the old addresses) must re-derive them - same tokens as before - and a decode graph captured
    before the resize must refuse to replay instead of writing into the freed cache."""
    from seed_amd import lib as L
    cfg = C.LLAMA_TINY
    sd = make_llama_state_dict(cfg, seed=9, norm_jitter=0.05)
    eng = LlamaEngine(sd, cfg, device="cuda", batch_cap=3, tmax=64)
    g = torch.Generator().manual_seed(5)
    reqs = [(torch.randint(3, cfg.vocab, (T0,), generator=g), n) for T0, n in ((5, 9), (11, 14), (7, 6), (20, 12))]
    want = [_alone(eng, p, n) for p, n in reqs]
    cb = ContinuousBatcher(eng, slots=3, chunk=4)
    ids = [cb.submit(p, n) for p, n in reqs[:2]]
    out = cb.run()
    assert [out[i] for i in ids] == want[:2]
    gen0 = eng.cache_generation
    first = eng.forward(reqs[0][0].view(1, -1).cuda(), last_only=True)[:, 0].float().argmax(-1, keepdim=True)
    replay, _ = eng.capture_decode_graph(first, 4)
    replay(1)
    eng.resize_cache(batch_cap=5, tmax=96)                    # what LlamaForCausalLM does when a larger batch arrives
    assert eng.cache_generation == gen0 + 1
    junk = [torch.empty(3, cfg.heads, 64, cfg.head_dim, dtype=torch.bfloat16, device="cuda").fill_(float("nan")) for _ in range(8)]
    with pytest.raises(L.SeedmiError, match="resized"):
        replay(1)
    ids = [cb.submit(p, n) for p, n in reqs]
    out = cb.run()                                            # slot structs and the captured step are rebuilt for the new cache
    assert [out[i] for i in ids] == want
    assert cb._generation == eng.cache_generation
    del junk
    eng.resize_cache(batch_cap=2)                             # fewer rows than this batcher's slots: refused, not corrupted
    cb.submit(*reqs[0])
    with pytest.raises(L.SeedmiError, match="slots"):
        cb.run()
