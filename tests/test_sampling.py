"""Next-token selection: greedy / temperature + top-p on the device.

CPU: the oracle's nucleus rule against the installed transformers' TemperatureLogitsWarper + TopPLogitsWarper (the
third-party code the reference's generate() runs).  GPU: seedmi_sample_token_bf16 through the C ABI against the oracle:
identical token wherever the draw is further than 1e-5 (relative mass) from a decision boundary — the device sums its fp32
weights in a different order than the oracle's float64 cumsum — plus exact greedy parity with first-index tie-break.
"""
import numpy as np
import pytest
import torch

from oracle import sample_oracle as S


def _logits(B, V, seed, scale=3.0, ties=False):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, V, generator=g) * scale
    if ties:
        x = (x * 2).round() / 2                      # coarse grid: many exact ties, also at the top
    return x.bfloat16()


@pytest.mark.parametrize("top_p,temperature", [(0.5, 1.0), (0.9, 0.7), (0.1, 1.3), (0.999, 1.0)])
def test_oracle_nucleus_matches_transformers_warpers(top_p, temperature):
    from transformers.generation.logits_process import TemperatureLogitsWarper, TopPLogitsWarper
    x = torch.randn(6, 1000, generator=torch.Generator().manual_seed(1)) * 3.0     # fp32 gaussian logits: no ties
    scores = TopPLogitsWarper(top_p=top_p)(None, TemperatureLogitsWarper(temperature)(None, x.clone()))
    hf_keep = torch.isfinite(scores)
    for b in range(x.shape[0]):
        order, n, margin = S.top_p_keep(x[b].numpy(), temperature, top_p)
        mine = np.zeros(x.shape[1], dtype=bool)
        mine[order[:n]] = True
        if margin > 1e-6:                            # HF sums in fp32; skip rows whose boundary test is within rounding
            assert np.array_equal(mine, hf_keep[b].numpy()), (b, n, int(hf_keep[b].sum()))


def test_oracle_draw_follows_the_renormalised_distribution():
    x = _logits(1, 50, 3, scale=1.0)[0].float().numpy()
    order, n, _ = S.top_p_keep(x, 1.0, 0.8)
    w = S.weights(x, 1.0)
    pk = w[order[:n]] / w[order[:n]].sum()
    us = (np.arange(20000) + 0.5) / 20000
    counts = np.zeros(50)
    for u in us:
        counts[S.sample_token(x, 1.0, 0.8, float(u))[0]] += 1
    assert counts[order[n:]].sum() == 0
    assert np.abs(counts[order[:n]] / us.size - pk).max() < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("V,B,ties", [(40194, 32, False), (40194, 8, True), (1154, 5, False), (300, 3, True), (49000, 2, False)])
@pytest.mark.parametrize("top_p,temperature", [(0.5, 1.0), (0.9, 0.7), (1.0, 1.0), (0.02, 1.0)])
def test_device_sampler_matches_oracle(V, B, ties, top_p, temperature):
    from seed_amd import lib as L
    lib = L.load()
    ldl = (V + 63) // 64 * 64
    x = torch.full((B, ldl), 77.0, dtype=torch.bfloat16)          # padding columns must be ignored even if they are huge
    x[:, :V] = _logits(B, V, V + B, ties=ties)
    xd = x.cuda()
    steps = 6
    u = torch.rand(steps, B, generator=torch.Generator().manual_seed(5))
    ud = u.cuda()
    tok = torch.empty(B, dtype=torch.int64, device="cuda")
    hist = torch.full((B, steps), -1, dtype=torch.int64, device="cuda")
    step_dev = torch.zeros(1, dtype=torch.int32, device="cuda")
    checked = 0
    for s in range(steps):
        step_dev.fill_(s + 3)                                     # step = *step_dev + offset
        L.check(lib.seedmi_sample_token_bf16(L.ptr(xd), ldl, B, V, temperature, top_p, L.ptr(ud), L.ptr(step_dev), -3, L.ptr(tok),
                                             L.ptr(hist), steps, steps, L.stream_ptr()), "sample")
        torch.cuda.synchronize()
        for b in range(B):
            want, margin = S.sample_token(x[b, :V].float().numpy(), temperature, top_p, float(u[s, b]))
            got = int(tok[b])
            assert 0 <= got < V
            assert int(hist[b, s]) == got
            if margin > 1e-5:
                assert got == want, (s, b, got, want, margin)
                checked += 1
            else:                                                  # on a boundary: must still be a kept token
                order, n, _ = S.top_p_keep(x[b, :V].float().numpy(), temperature, top_p)
                assert got in set(order[:n + 1].tolist())
    assert checked >= 0.5 * steps * B      # (with top_p = 1 and 40k+ tokens many draws sit within 1e-5 of a boundary)


@pytest.mark.gpu
def test_device_greedy_is_first_index_argmax():
    from seed_amd import lib as L
    lib = L.load()
    V, B = 40194, 16
    x = _logits(B, V, 9, ties=True)
    x[3, 100] = x[3, 20000] = x[3].max() + 1                       # forced tie at the top: the lower id must win
    xd = x.cuda()
    tok = torch.empty(B, dtype=torch.int64, device="cuda")
    L.check(lib.seedmi_sample_token_bf16(L.ptr(xd), V, B, V, 1.0, 0.0, None, None, 0, L.ptr(tok), None, 0, 0, L.stream_ptr()), "greedy")
    torch.cuda.synchronize()
    want = [S.greedy_token(x[b].float().numpy()) for b in range(B)]
    assert tok.cpu().tolist() == want
    assert want[3] == 100
