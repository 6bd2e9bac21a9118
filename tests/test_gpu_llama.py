"""LlamaForCausalLM.forward parity on the GPU: HIP path (C ABI) vs the CPU oracle and the reference-module golden.

Tolerance (stated, BASELINE north_star "LLaMA logits within a stated fp tolerance"): bf16 logits within
atol = 2e-2 * max|logit| and the normalised error below 2e-2 of the fp32 oracle; greedy tokens equal to the fp32
oracle's wherever its top-2 logit gap exceeds 1e-2 * max|logit|.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import seed_oracle as O  # noqa: E402
from seed_amd import config as C  # noqa: E402
from seed_amd.llama_engine import LlamaEngine  # noqa: E402
from seed_amd.weights import make_llama_state_dict  # noqa: E402


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm()).item()


def _check_logits(got, ref32, ref16, tag):
    got = got.float().cpu()
    e, e16 = _rel(got, ref32), _rel(ref16, ref32)
    mx = ref32.abs().max().item()
    worst = (got - ref32).abs().max().item()
    print(f"[{tag}] logits rel vs fp32 oracle: hip {e:.3e} / bf16-oracle {e16:.3e}; max abs err {worst:.3e} (max|logit| {mx:.3f})")
    assert e < max(1.5 * e16, 2e-2), (e, e16)
    assert worst <= 2e-2 * mx + 4 * (ref16 - ref32).abs().max().item()


@pytest.mark.parametrize("B,T", [(2, 12), (8, 12)])     # M = 24 -> weight-streaming GEMM, M = 96 -> 128x128 MFMA GEMM
def test_llama_prefill_and_decode(golden_dir, B, T):
    cfg = C.LLAMA_TINY
    g = np.load(os.path.join(golden_dir, "llama_tiny.npz"))
    sd = make_llama_state_dict(cfg, seed=int(g["seed_w"]), norm_jitter=float(g["norm_jitter"]))
    if B == 2:
        ids = torch.from_numpy(g["input_ids"])
    else:
        ids = torch.randint(3, cfg.vocab, (B, T), generator=torch.Generator().manual_seed(6))
    eng = LlamaEngine(sd, cfg, device="cuda", batch_cap=B, tmax=64)
    logits = eng.forward(ids.cuda())
    torch.cuda.synchronize()
    assert tuple(logits.shape) == (B, T, cfg.vocab)
    l32, past32 = O.llama_forward(sd, cfg, ids, mode="fp32")
    l16, _ = O.llama_forward(sd, cfg, ids, mode="bf16")
    _check_logits(logits, l32, l16, f"prefill B{B}")
    # KV cache layout = the reference's past_key_values: [B,H,T,hd], keys post-RoPE (llama_xformer.py:236-239)
    assert _rel(eng.k_cache[0][:B, :, :T].float(), past32[0][0]) < 1e-2
    assert _rel(eng.v_cache[1][:B, :, :T].float(), past32[1][1]) < 2e-2
    if B == 2:
        assert _rel(logits.float(), torch.from_numpy(g["prefill_logits_fp32"])) < 2e-2      # reference module golden
    # greedy decode: prefill (last position only) + n_new-1 cached single-token steps
    n_new = 6
    toks, steps = eng.greedy_decode(ids.cuda(), n_new)
    torch.cuda.synchronize()
    t32, s32 = O.llama_greedy_decode(sd, cfg, ids, n_new, mode="fp32")
    # teacher-forced comparison of the decode-step logits: feed the oracle's tokens through the engine
    eng.reset()
    lg = eng.forward(ids.cuda(), last_only=True)
    _check_logits(lg[:, 0], s32[:, 0], s32[:, 0], f"prefill-last B{B}")
    for i in range(1, n_new):
        lg = eng.forward(t32[:, i - 1:i].cuda(), last_only=True)
        _check_logits(lg[:, 0], s32[:, i], s32[:, i], f"decode step {i} B{B}")
    top2 = s32.topk(2, dim=-1).values
    confident = (top2[..., 0] - top2[..., 1]) > 1e-2 * s32.abs().amax(-1)
    same = toks.cpu() == t32
    # rows stay comparable until their first divergence; up to there every confident step must agree
    alive = torch.ones(B, dtype=torch.bool)
    for i in range(n_new):
        assert (same[:, i] | ~confident[:, i] | ~alive).all(), f"greedy token differs at confident step {i}"
        alive &= same[:, i]
    print(f"[greedy B{B}] token agreement {same.float().mean().item():.3f}")


@pytest.mark.parametrize("B,T", [(2, 12), (8, 12)])
def test_llama_fp16_prefill_and_decode(B, T):
    """VERDICT r4 item 5 on the LLM side: the reference's shipped model dtype (`torch_dtype: fp16`, configs/llm/seed_llama_8b.yaml:4) through
    libseedmi_f16.so - the same kernels with fp16 as the 16-bit element - against the oracle in fp32 and in its "fp16" choreography
    (llama_xformer.py:105-113 RMSNorm island, :147-150 RoPE tables cast on read, :718 logits in the model dtype), prefill at both GEMM
    paths, the KV cache, cached decode steps and the hipGraph-replayed greedy loop."""
    cfg = C.LLAMA_TINY
    sd = make_llama_state_dict(cfg, seed=3, norm_jitter=0.05)
    ids = torch.randint(3, cfg.vocab, (B, T), generator=torch.Generator().manual_seed(6 + B))
    eng = LlamaEngine(sd, cfg, device="cuda", batch_cap=B, tmax=64, dtype=torch.float16)
    assert eng.lib.seedmi_compute_dtype() == 1
    logits = eng.forward(ids.cuda())
    torch.cuda.synchronize()
    assert logits.dtype == torch.float16 and tuple(logits.shape) == (B, T, cfg.vocab)
    l32, past32 = O.llama_forward(sd, cfg, ids, mode="fp32")
    l16, _ = O.llama_forward(sd, cfg, ids, mode="fp16")
    _check_logits(logits, l32, l16, f"fp16 prefill B{B}")
    e16, eb = _rel(logits.float(), l32), _rel(O.llama_forward(sd, cfg, ids, mode="bf16")[0], l32)
    print(f"[fp16 prefill B{B}] rel vs fp32: fp16 build {e16:.3e}, a bf16 pipeline {eb:.3e}")
    assert e16 < 0.5 * eb                                   # 11 significand bits against 8: the fp16 build must be the closer one
    assert _rel(eng.k_cache[0][:B, :, :T].float(), past32[0][0]) < 2e-3
    assert _rel(eng.v_cache[1][:B, :, :T].float(), past32[1][1]) < 4e-3
    n_new = 6
    t32, s32 = O.llama_greedy_decode(sd, cfg, ids, n_new, mode="fp32")
    eng.reset()
    lg = eng.forward(ids.cuda(), last_only=True)
    _check_logits(lg[:, 0], s32[:, 0], s32[:, 0], f"fp16 prefill-last B{B}")
    for i in range(1, n_new):
        lg = eng.forward(t32[:, i - 1:i].cuda(), last_only=True)
        _check_logits(lg[:, 0], s32[:, i], s32[:, i], f"fp16 decode step {i} B{B}")
    toks, _ = eng.greedy_decode(ids.cuda(), n_new)
    graphed = eng.greedy_decode_graph(ids.cuda(), n_new)
    torch.cuda.synchronize()
    assert torch.equal(toks, graphed), "fp16: hipGraph replay and the eager loop disagree"
    top2 = s32.topk(2, dim=-1).values
    confident = (top2[..., 0] - top2[..., 1]) > 1e-2 * s32.abs().amax(-1)
    same, alive = toks.cpu() == t32, torch.ones(B, dtype=torch.bool)
    for i in range(n_new):
        assert (same[:, i] | ~confident[:, i] | ~alive).all(), f"fp16: greedy token differs at confident step {i}"
        alive &= same[:, i]


def test_graph_decode_equals_eager_decode():
    """The hipGraph-replayed decode loop (device-resident cache length) produces exactly the eager loop's tokens."""
    cfg = C.LLAMA_TINY
    sd = make_llama_state_dict(cfg, seed=9, norm_jitter=0.05)
    ids = torch.randint(3, cfg.vocab, (3, 10), generator=torch.Generator().manual_seed(4)).cuda()
    eng = LlamaEngine(sd, cfg, device="cuda", batch_cap=3, tmax=64)
    eager, _ = eng.greedy_decode(ids, 9)
    graphed = eng.greedy_decode_graph(ids, 9)
    torch.cuda.synchronize()
    assert torch.equal(eager, graphed)
    assert eng.past_len == 10 + 8


def test_sampled_graph_decode_is_reproducible_and_stays_in_the_nucleus():
    """generate(do_sample=True, top_p=0.5) on the device: the captured step (forward + seedmi_sample_token_bf16) replayed from
    a hipGraph gives exactly the tokens of an eager loop fed the same uniforms, and every drawn token lies in the top-p
    nucleus of the logits it was drawn from (checked with the sampling oracle)."""
    from oracle import sample_oracle as S
    cfg = C.LLAMA_TINY
    sd = make_llama_state_dict(cfg, seed=9, norm_jitter=0.05)
    B, T0, n_new, top_p = 3, 10, 9, 0.5
    ids = torch.randint(3, cfg.vocab, (B, T0), generator=torch.Generator().manual_seed(4)).cuda()
    eng = LlamaEngine(sd, cfg, device="cuda", batch_cap=B, tmax=64)
    g = torch.Generator(device="cuda").manual_seed(123)
    graphed = eng.sample_decode_graph(ids, n_new, top_p=top_p, temperature=1.0, generator=g).clone()
    torch.cuda.synchronize()
    assert eng.past_len == T0 + n_new - 1
    # eager replay with the same uniforms
    u = torch.rand(n_new, B, dtype=torch.float32, device="cuda", generator=torch.Generator(device="cuda").manual_seed(123))
    eng.reset()
    logits = eng.forward(ids, last_only=True)
    tok = torch.empty(B, dtype=torch.int64, device="cuda")
    step = torch.zeros(1, dtype=torch.int32, device="cuda")
    eager = []
    for s in range(n_new):
        step.fill_(s)
        eng.select_token(logits[:, 0], tok, top_p, 1.0, u, step, 0, None)
        torch.cuda.synchronize()
        row_logits = logits[:, 0].float().cpu().numpy()
        for b in range(B):
            order, n, _ = S.top_p_keep(row_logits[b], 1.0, top_p)
            assert int(tok[b]) in set(order[:n + 1].tolist()), (s, b)
        eager.append(tok.clone())
        if s + 1 < n_new:
            logits = eng.forward(tok.view(B, 1), last_only=True)
    assert torch.equal(torch.stack(eager, dim=1), graphed)
    # a different seed gives a different continuation; top_p = 0 is the greedy path
    other = eng.sample_decode_graph(ids, n_new, top_p=top_p, generator=torch.Generator(device="cuda").manual_seed(7))
    assert not torch.equal(other, graphed)
    assert torch.equal(eng.sample_decode_graph(ids, n_new, top_p=0.0), eng.greedy_decode(ids, n_new)[0])


@pytest.mark.parametrize("fold_norm", [True, False], ids=["folded_norm", "explicit_norm"])
def test_llama8b_width_parity_with_oracle(fold_norm):
    """BASELINE.json config 3 at its real WIDTH against the oracle: SEED-LLaMA-8B dims (hidden 4096, FFN 11008,
    32 heads x 128, vocab 40194) with 2 of the 32 layers - every kernel shape of the timed decode path (K = 4096 / 11008 folded-norm
    weight-streaming GEMMs, fused RoPE + append + attention, lm_head on the last position) - B = 32, prompt T0 = 59 with a 32-code
    image span.  Prefill-last logits and teacher-forced decode steps 1, 2, 8 within 2e-2 * max|logit| / 2e-2 normalised of the fp32
    oracle (and no further from it than 1.5x the bf16 oracle), greedy tokens equal on confident rows, for the eager loop AND the
    hipGraph replay, with the RMSNorm folded into the GEMMs (what bench.py times) and with explicit norm launches."""
    from dataclasses import replace
    cfg = replace(C.LLAMA_8B, layers=2)
    sd = make_llama_state_dict(cfg, seed=0, dtype=torch.bfloat16, norm_jitter=0.05)      # bf16-representable: oracle and engine share weights
    B, T0, n_new = 32, 59, 9
    g = torch.Generator().manual_seed(99)
    prompt = torch.randint(3, 32000, (B, T0), generator=g)
    prompt[:, 0] = 1
    prompt[:, 9] = 32000 + 8192                                                             # <img>
    prompt[:, 10:42] = 32000 + torch.randint(0, 8192, (B, 32), generator=g)
    prompt[:, 42] = 32000 + 8193                                                            # </img>
    eng = LlamaEngine(sd, cfg, device="cuda", batch_cap=B, tmax=96, fold_norm=fold_norm)
    # oracle: prefill, then greedy steps on its own tokens (fp32) and the same tokens teacher-forced through the bf16 oracle
    l32, p32 = O.llama_forward(sd, cfg, prompt, mode="fp32")
    l16, p16 = O.llama_forward(sd, cfg, prompt, mode="bf16")
    s32, s16 = [l32[:, -1]], [l16[:, -1]]
    toks = [l32[:, -1].argmax(-1, keepdim=True)]
    for i in range(1, n_new):
        a, p32 = O.llama_forward(sd, cfg, toks[-1], past=p32, mode="fp32")
        b, p16 = O.llama_forward(sd, cfg, toks[-1], past=p16, mode="bf16")
        s32.append(a[:, -1]); s16.append(b[:, -1])
        toks.append(a[:, -1].argmax(-1, keepdim=True))
    t32 = torch.cat(toks, dim=1)
    s32, s16 = torch.stack(s32, 1), torch.stack(s16, 1)
    # engine, teacher-forced
    lg = eng.forward(prompt.cuda(), last_only=True)
    _check_logits(lg[:, 0], s32[:, 0], s16[:, 0], f"8B-width prefill-last fold={fold_norm}")
    for i in range(1, n_new):
        lg = eng.forward(t32[:, i - 1:i].cuda(), last_only=True)
        if i in (1, 2, 8):
            _check_logits(lg[:, 0], s32[:, i], s16[:, i], f"8B-width decode step {i} fold={fold_norm}")
    torch.cuda.synchronize()
    # post-RoPE keys in the reference's cache layout: no further from the fp32 oracle than 1.5x the bf16 oracle's own distance
    ek, ek16 = _rel(eng.k_cache[1][:B, :, :T0].float(), p32[1][0][:, :, :T0]), _rel(p16[1][0][:, :, :T0], p32[1][0][:, :, :T0])
    print(f"[8B-width layer-1 keys fold={fold_norm}] rel vs fp32 oracle: hip {ek:.3e} / bf16-oracle {ek16:.3e}")
    assert ek < max(1.5 * ek16, 5e-3), (ek, ek16)
    # free-running greedy: eager loop and hipGraph replay vs the fp32 oracle on confident rows.  Confident = the fp32 top-2 gap exceeds both
    # SURVEY 8d's margin (1e-2 * max|logit|) and 3x the bf16 ORACLE's own largest deviation on that row: at this width the two bf16
    # pipelines sit 0.11-0.12 (max abs) from fp32, above the 0.07 the first margin alone allows - a row inside that band flips with the
    # order of an fp32 summation (it did when the split-K decode GEMM changed gate/up's), which is noise, not a defect
    top2 = s32.topk(2, dim=-1).values
    gap = top2[..., 0] - top2[..., 1]
    confident = (gap > 1e-2 * s32.abs().amax(-1)) & (gap > 3.0 * (s16 - s32).abs().amax(-1))
    eager, _ = eng.greedy_decode(prompt.cuda(), n_new)
    graphed = eng.greedy_decode_graph(prompt.cuda(), n_new).clone()
    torch.cuda.synchronize()
    assert torch.equal(eager, graphed), "graph replay differs from the eager loop"
    same = eager.cpu() == t32
    alive = torch.ones(B, dtype=torch.bool)
    for i in range(n_new):
        assert (same[:, i] | ~confident[:, i] | ~alive).all(), f"greedy token differs at confident step {i}"
        alive &= same[:, i]
    print(f"[8B-width greedy fold={fold_norm}] token agreement {same.float().mean().item():.3f}, confident {confident.float().mean().item():.3f}")


def _config3_prompt(B, T0=59):
    g = torch.Generator().manual_seed(99)
    prompt = torch.randint(3, 32000, (B, T0), generator=g)
    prompt[:, 0] = 1
    prompt[:, 9] = 32000 + 8192                                                             # <img>
    prompt[:, 10:42] = 32000 + torch.randint(0, 8192, (B, 32), generator=g)
    prompt[:, 42] = 32000 + 8193                                                            # </img>
    return prompt


def _full_depth_config3(sd, report_name, tag):
    """BASELINE.json config 3 at FULL depth against the oracle; returns the report and the boolean maps the callers assert on.

    The engine decodes freely (graph replay); its own tokens are then teacher-forced through the eager path (logits kept at the probed
    steps) and through the oracle as ONE causal forward over prompt + tokens: position T0 - 1 + i of that forward is decode step i with
    exactly the engine's context (cached decoding and a causal forward are the same function; the oracle's rounding points are row-wise)."""
    import gc
    import json
    import time
    cfg = C.LLAMA_8B
    B, T0, n_steps = 4, 59, 128
    probes = (0, 1, 2, 64, 128)
    prompt = _config3_prompt(B, T0)
    eng = LlamaEngine(sd, cfg, device="cuda", batch_cap=B, tmax=T0 + n_steps + 5, fold_norm=True)
    graphed = eng.greedy_decode_graph(prompt.cuda(), n_steps + 1).clone()                   # tokens t_1 .. t_129 (128 decode steps)
    torch.cuda.synchronize()
    eng.reset()
    kept = {}
    lg = eng.forward(prompt.cuda(), last_only=True)
    kept[0] = lg[:, 0].float().cpu()
    eager = [lg[:, 0].float().argmax(-1)]
    for i in range(1, n_steps + 1):
        lg = eng.forward(graphed[:, i - 1:i], last_only=True)
        if i in probes:
            kept[i] = lg[:, 0].float().cpu()
        eager.append(lg[:, 0].float().argmax(-1))
    torch.cuda.synchronize()
    eager = torch.stack(eager, dim=1)
    assert torch.equal(eager, graphed), "teacher-forced eager path and hipGraph replay disagree"
    sd_cpu = {k: v.cpu() for k, v in sd.items()}
    del eng
    gc.collect()
    torch.cuda.empty_cache()
    seq = torch.cat([prompt, graphed[:, :n_steps].cpu()], dim=1)                            # [B, T0 + 128]
    t0 = time.time()
    l32, _ = O.llama_forward(sd_cpu, cfg, seq, mode="fp32")
    l16, _ = O.llama_forward(sd_cpu, cfg, seq, mode="bf16")
    oracle_s = time.time() - t0
    report = {"weights": tag, "B": B, "T0": T0, "layers": cfg.layers, "oracle_seconds": round(oracle_s, 1), "steps": {}}
    for i in probes:
        pos = T0 - 1 + i
        r32, r16 = l32[:, pos], l16[:, pos]
        what = "prefill-last" if i == 0 else f"decode step {i}"
        _check_logits(kept[i], r32, r16, f"8B full depth [{tag}] {what}")
        # SURVEY 8d as written: atol 2e-2 * max|logit| + rtol 2e-2 against the bf16 oracle.  REPORTED un-widened (what fraction of the
        # logits meets it and by how much the worst one misses), ASSERTED widened by the bf16 oracle's own distance from fp32: two bf16
        # pipelines that round in different places are each that far from the truth, and at 32 layers that distance alone exceeds 8d's 2 %
        d16 = (kept[i] - r16).abs()
        tol0 = 2e-2 * r16.abs().max() + 2e-2 * r16.abs()
        tol = tol0 + 2 * (r16 - r32).abs().max()
        assert (d16 <= tol).all(), (what, d16.max().item(), tol.min().item())
        report["steps"][what] = {"rel_vs_fp32": _rel(kept[i], r32), "rel_bf16_oracle_vs_fp32": _rel(r16, r32), "rel_vs_bf16_oracle": _rel(kept[i], r16),
                                 "max_abs_vs_bf16_oracle": d16.max().item(), "max_abs_logit": r32.abs().max().item(),
                                 "survey_8d_unwidened": {"fraction_of_logits_within": (d16 <= tol0).float().mean().item(),
                                                         "worst_error_over_tolerance": (d16 / tol0).max().item(),
                                                         "bf16_oracle_vs_fp32_worst_over_tolerance": ((r16 - r32).abs() / tol0).max().item()}}
        print(f"[8B full depth {tag}] {what}: un-widened 8d tolerance met by {(d16 <= tol0).float().mean().item():.4f} of the logits, worst "
              f"{(d16 / tol0).max().item():.2f}x the tolerance (the bf16 oracle vs fp32: {((r16 - r32).abs() / tol0).max().item():.2f}x)")
    # greedy ids: the engine's token after context i vs the fp32 oracle's argmax at the same context.  SURVEY 8d's margin (top-2 gap >
    # 1e-2 * max|logit|) was chosen for shallow stacks: after 32 layers two bf16 pipelines differ from fp32 by more than that (the bf16
    # ORACLE's own argmax flips on such rows, counted below), so a row is "confident" when its fp32 top-2 gap exceeds 3x the bf16 oracle's
    # own largest logit deviation on that row - a pipeline within 1.5x the oracle's deviation (asserted above at the probes) cannot flip it.
    s32, s16 = l32[:, T0 - 1:T0 + n_steps], l16[:, T0 - 1:T0 + n_steps]                     # [B, 129, V]
    top2 = s32.topk(2, dim=-1).values
    gap = top2[..., 0] - top2[..., 1]
    dev16 = (s16 - s32).abs().amax(-1)
    confident = gap > 3.0 * dev16
    loose = gap > 1e-2 * s32.abs().amax(-1)
    same = graphed.cpu() == s32.argmax(-1)
    same16 = s16.argmax(-1) == s32.argmax(-1)
    report["greedy_agreement"] = same.float().mean().item()
    report["greedy_agreement_bf16_oracle"] = same16.float().mean().item()
    report["confident_fraction"] = confident.float().mean().item()
    report["survey_margin_rows"] = {"fraction": loose.float().mean().item(), "hip_differs": int((~same & loose).sum()),
                                    "bf16_oracle_differs": int((~same16 & loose).sum())}
    print(f"[8B full depth {tag}] greedy agreement with the fp32 oracle: hip {report['greedy_agreement']:.3f} / bf16 oracle "
          f"{report['greedy_agreement_bf16_oracle']:.3f} over {same.numel()} positions; confident {report['confident_fraction']:.3f}; "
          f"rows above the 1e-2 margin: hip differs on {report['survey_margin_rows']['hip_differs']}, the bf16 oracle itself on "
          f"{report['survey_margin_rows']['bf16_oracle_differs']}; oracle {oracle_s:.1f} s")
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    json.dump(report, open(os.path.join(out, report_name), "w"), indent=1)
    oracle_argmax = s32.argmax(-1)
    del sd_cpu, l32, l16
    gc.collect()
    return report, same, same16, confident, seq, oracle_argmax


def test_llama8b_full_depth_config3_parity():
    """BASELINE.json config 3 as SURVEY 8d specifies it, at FULL depth: SEED-LLaMA-8B (32 layers, hidden 4096, FFN 11008, vocab 40194),
    the config-3 prompt (BOS + 8 text + <img> + 32 image codes + </img> + 16 text = T0 59), B = 4 (oracle cost; the kernels' shapes do not
    depend on B below 64), 128 greedy decode steps through the folded-norm hipGraph path that bench.py times.  Logits at the prefill's
    last position and at decode steps {1, 2, 64, 128} against the oracle in fp32 and in the bf16 choreography, greedy ids against the fp32
    oracle's argmax on confident rows at EVERY one of the 129 positions (llama_xformer.py:280-332, 496-627, 661-743).

    Weights: the reference's own initialiser, N(0, 0.02) everywhere (llama_xformer.py:363-372).  Its logits are nearly flat (top-2 gap
    ~0.1 under a bf16 noise of ~0.3), so only ~1 % of the positions are confident: this run is the NOISE-FLOOR report (logit distances,
    how often the bf16 oracle itself flips); the token evidence is test_llama8b_full_depth_peaked_logits_greedy_parity below."""
    sd = make_llama_state_dict(C.LLAMA_8B, seed=0, device="cuda", dtype=torch.bfloat16, norm_jitter=0.05)
    report, same, same16, confident, _, _ = _full_depth_config3(sd, "llama8b_depth_parity.json", "N(0, 0.02) weights")
    assert (same | ~confident).all(), f"{(~same & confident).sum().item()} greedy ids differ from the fp32 oracle on confident rows"
    # and the HIP path is no worse at picking the fp32 oracle's token than the oracle's own bf16 choreography (3 flips of slack)
    assert (~same).sum() <= (~same16).sum() * 1.25 + 3, ((~same).sum().item(), (~same16).sum().item())


def test_llama8b_full_depth_peaked_logits_greedy_parity():
    """config 3 at full depth on a second synthetic state dict whose logits are PEAKED, so that greedy-token parity
    is asserted on MOST of the 516 positions instead of ~1 % of them.  seed_amd.weights.make_llama_successor_state_dict: the same
    N(0, 0.02) 32-layer body, embed_tokens ~ N(0, 1.2) and lm_head = a permutation of the embedding rows: the fp32 model emits
    successor[last token] (a walk through the vocabulary, a different token at every step) with a top-2 gap several times the bf16
    deviation, while attention and the MLPs still carry ~99 % of the residual stream's variance.  Asserted: at least half of the 516
    positions are confident (fp32 top-2 gap > 3x the bf16 oracle's own largest deviation on that row), the hipGraph-replayed greedy ids
    equal the fp32 oracle's argmax on EVERY confident position, and the fp32 oracle's argmax there is the successor the weights were
    built for (the test knows what the right token is without trusting either pipeline)."""
    from seed_amd.weights import make_llama_successor_state_dict
    sd, successor = make_llama_successor_state_dict(C.LLAMA_8B, seed=0, device="cuda", dtype=torch.bfloat16, norm_jitter=0.05)
    report, same, same16, confident, seq, oracle_argmax = _full_depth_config3(sd, "llama8b_depth_parity_peaked.json",
                                                                              "successor weights, embed_std 1.2")
    frac = confident.float().mean().item()
    assert frac >= 0.5, f"only {frac:.3f} of the positions are confident: the peaked state dict is not peaked enough"
    assert (same | ~confident).all(), f"{(~same & confident).sum().item()} greedy ids differ from the fp32 oracle on confident rows"
    # what the fp32 oracle picks on confident rows is the successor of the last context token
    last = seq[:, 59 - 1:59 + 128]                                                          # the token fed at each of the 129 positions
    hit = oracle_argmax == successor.cpu()[last]
    assert (hit | ~confident).float().mean().item() > 0.98, (hit & confident).float().mean().item()
    assert (~same).sum() <= (~same16).sum() * 1.25 + 3, ((~same).sum().item(), (~same16).sum().item())
    print(f"[8B full depth peaked] confident {frac:.3f}, greedy agreement {same.float().mean().item():.3f}, fp32 oracle emits the successor on "
          f"{hit.float().mean().item():.3f} of the positions")


def test_llama14b_width_parity_with_oracle():
    """VERDICT r3 item 1a / BASELINE.json config 5 at its real WIDTH against the oracle: SEED-LLaMA-14B dims (LLaMA-2-13B body: hidden
    5120, 40 heads x 128, FFN 13824, rms eps 1e-5, vocab 40194) with 2 of the 40 layers, B = 8 sequences of T = 649 built exactly as
    bench.py's config-5 leg builds them (4 x (128 text ids, <img>, 32 image codes, </img>) after BOS;
    gradio_demo/seed_llama_flask.py:144-150 is the interleaving, llama_xformer.py:496-627 the forward).  One prefill with the KV cache
    written, logits of ALL 649 positions (the reference's API, llama_xformer.py:718) against O.llama_forward in fp32 and bf16 under the
    same contract as the 8B tests; hidden 5120 / FFN 13824 hit tile counts of the 256x256 GEMM, the SwiGLU epilogue and the tiled
    causal attention (T = 649: 5.07 query tiles) that no 8B test does.  The cache rows are compared too (they feed the decode that follows)."""
    from dataclasses import replace
    cfg = replace(C.LLAMA_14B, layers=2)
    sd = make_llama_state_dict(cfg, seed=0, dtype=torch.bfloat16, norm_jitter=0.05)
    B, T = 8, 649
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(3, 32000, (B, T), generator=g)
    ids[:, 0] = 1
    for r in range(4):
        s0 = 1 + r * (128 + 34) + 128
        ids[:, s0] = 32000 + 8192
        ids[:, s0 + 1:s0 + 33] = 32000 + torch.randint(0, 8192, (B, 32), generator=g)
        ids[:, s0 + 33] = 32000 + 8193
    eng = LlamaEngine(sd, cfg, device="cuda", batch_cap=B, tmax=704, decode_packed=False)      # as bench.py's leg constructs it
    logits = eng.forward(ids.cuda())
    torch.cuda.synchronize()
    assert tuple(logits.shape) == (B, T, cfg.vocab)
    last = eng.forward(ids.cuda(), past_len=0, last_only=True)                              # the leg bench.py times: last position only
    torch.cuda.synchronize()
    # (the last-position lm_head runs on the weight-streaming kernel, M = 8: another fp32 summation order than the MFMA tile's)
    assert _rel(last[:, 0].float(), logits[:, -1].float()) < 5e-3, "last_only prefill differs from the all-positions prefill"
    l32, p32 = O.llama_forward(sd, cfg, ids, mode="fp32")
    l16, p16 = O.llama_forward(sd, cfg, ids, mode="bf16")
    _check_logits(logits, l32, l16, "14B-width prefill, all 649 positions")
    for pos in (0, 128, 161, 162, 648):                      # BOS, <img>, last image code, </img>, last position
        _check_logits(logits[:, pos], l32[:, pos], l16[:, pos], f"14B-width prefill position {pos}")
    for layer in (0, 1):
        ek, ek16 = _rel(eng.k_cache[layer][:B, :, :T].float(), p32[layer][0]), _rel(p16[layer][0], p32[layer][0])
        ev, ev16 = _rel(eng.v_cache[layer][:B, :, :T].float(), p32[layer][1]), _rel(p16[layer][1], p32[layer][1])
        print(f"[14B-width layer {layer}] keys rel vs fp32 oracle: hip {ek:.3e} / bf16-oracle {ek16:.3e}; values {ev:.3e} / {ev16:.3e}")
        assert ek < max(1.5 * ek16, 5e-3) and ev < max(1.5 * ev16, 5e-3), (layer, ek, ek16, ev, ev16)
    # greedy next token of every position vs the fp32 oracle on confident rows (same definition as the 8B width test)
    top2 = l32.topk(2, dim=-1).values
    gap = top2[..., 0] - top2[..., 1]
    confident = (gap > 1e-2 * l32.abs().amax(-1)) & (gap > 3.0 * (l16 - l32).abs().amax(-1))
    same = logits.float().cpu().argmax(-1) == l32.argmax(-1)
    assert (same | ~confident).all(), f"{(~same & confident).sum().item()} argmax ids differ on confident positions"
    print(f"[14B-width] argmax agreement {same.float().mean().item():.3f}, confident {confident.float().mean().item():.3f}")
    # one cached decode step on top of the prefill (M = 8: the weight-streaming path at K = 5120 / 13824 on row-major weights)
    tok = l32[:, -1].argmax(-1, keepdim=True)
    a32, _ = O.llama_forward(sd, cfg, tok, past=p32, mode="fp32")
    a16, _ = O.llama_forward(sd, cfg, tok, past=p16, mode="bf16")
    step = eng.forward(tok.cuda(), last_only=True)
    torch.cuda.synchronize()
    _check_logits(step[:, 0], a32[:, -1], a16[:, -1], "14B-width decode step 1")


def test_llama14b_full_depth_parity():
    """VERDICT r4 item 4 / BASELINE.json config 5 at FULL depth: SEED-LLaMA-14B (LLaMA-2-13B body: 40 layers, hidden 5120, 40 heads x 128,
    FFN 13824, rms eps 1e-5, vocab 40194), B = 2 sequences of T = 649 built exactly as bench.py's config-5 leg builds them, on the PEAKED
    ("successor") weights, so that token parity is asserted on most positions.  One prefill with the KV cache written - logits at
    positions {0, 128, 161, 162, 648} (BOS, <img>, last image code, </img>, last) - then 16 greedy decode steps through the folded-norm
    hipGraph path (the config's decode side), teacher-forced through the eager path for their logits.  Oracle: O.llama_forward as ONE
    causal forward over prompt + the engine's tokens, fp32 and bf16 (llama_xformer.py:496-627, 661-743).  Asserted under _check_logits
    (widened by the bf16 oracle's own distance, as at 8B); SURVEY 8d's un-widened tolerance is REPORTED beside it
    (profiles/r05_llama14b_depth_parity.json)."""
    import gc
    import json
    import time
    from seed_amd.weights import make_llama_successor_state_dict
    cfg = C.LLAMA_14B
    B, T, n_steps = 2, 649, 16
    # embed_std 2.0: measured with the oracle at these dims / 40 layers (fp32 vs bf16, 40 positions): 0.95 of the positions confident
    # (1.2, the 8B value: 0.10 - the 40-layer, 5120-wide body adds more to the stream; 3.0: 1.00)
    sd, successor = make_llama_successor_state_dict(cfg, seed=0, device="cuda", dtype=torch.bfloat16, norm_jitter=0.05, embed_std=2.0)
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(3, 32000, (B, T), generator=g)
    ids[:, 0] = 1
    for r in range(4):
        s0 = 1 + r * (128 + 34) + 128
        ids[:, s0] = 32000 + 8192
        ids[:, s0 + 1:s0 + 33] = 32000 + torch.randint(0, 8192, (B, 32), generator=g)
        ids[:, s0 + 33] = 32000 + 8193
    probes = (0, 128, 161, 162, 648)
    eng = LlamaEngine(sd, cfg, device="cuda", batch_cap=B, tmax=704, fold_norm=True)
    allpos = eng.forward(ids.cuda())                                                        # the reference's API: logits of every position
    torch.cuda.synchronize()
    kept_prefill = {p_: allpos[:, p_].float().cpu() for p_ in probes}
    hip_argmax_prefill = allpos.float().argmax(-1).cpu()
    del allpos
    graphed = eng.greedy_decode_graph(ids.cuda(), n_steps + 1).clone()                      # t_1 .. t_17: 16 decode steps behind the prefill
    torch.cuda.synchronize()
    eng.reset()
    lg = eng.forward(ids.cuda(), last_only=True)
    eager = [lg[:, 0].float().argmax(-1)]
    kept_step = {}
    for i in range(1, n_steps + 1):
        lg = eng.forward(graphed[:, i - 1:i], last_only=True)
        kept_step[i] = lg[:, 0].float().cpu()
        eager.append(lg[:, 0].float().argmax(-1))
    torch.cuda.synchronize()
    assert torch.equal(torch.stack(eager, dim=1), graphed), "teacher-forced eager path and hipGraph replay disagree"
    eng.decode_status(B)
    sd_cpu = {k: v.cpu() for k, v in sd.items()}
    del eng, sd
    gc.collect()
    torch.cuda.empty_cache()
    seq = torch.cat([ids, graphed[:, :n_steps].cpu()], dim=1)                               # [B, 649 + 16]
    nthreads = torch.get_num_threads()
    torch.set_num_threads(min(32, os.cpu_count() or 1))                                     # (torch's CPU GEMMs lose to oversubscription beyond ~32)
    try:
        t0 = time.time()
        l32, _ = O.llama_forward(sd_cpu, cfg, seq, mode="fp32")
        l16, _ = O.llama_forward(sd_cpu, cfg, seq, mode="bf16")
        oracle_s = time.time() - t0
    finally:
        torch.set_num_threads(nthreads)
    report = {"model": "SEED-LLaMA-14B dims, 40 layers, successor weights (embed_std 2.0)", "B": B, "T": T, "decode_steps": n_steps,
              "oracle_seconds": round(oracle_s, 1), "points": {}}

    def check(got, pos, what):
        r32, r16 = l32[:, pos], l16[:, pos]
        _check_logits(got, r32, r16, f"14B full depth {what}")
        d16 = (got - r16).abs()
        tol0 = 2e-2 * r16.abs().max() + 2e-2 * r16.abs()
        tol = tol0 + 2 * (r16 - r32).abs().max()
        assert (d16 <= tol).all(), (what, d16.max().item(), tol.min().item())
        report["points"][what] = {"rel_vs_fp32": _rel(got, r32), "rel_bf16_oracle_vs_fp32": _rel(r16, r32), "rel_vs_bf16_oracle": _rel(got, r16),
                                  "survey_8d_unwidened": {"fraction_of_logits_within": (d16 <= tol0).float().mean().item(),
                                                          "worst_error_over_tolerance": (d16 / tol0).max().item(),
                                                          "bf16_oracle_vs_fp32_worst_over_tolerance": ((r16 - r32).abs() / tol0).max().item()}}
    for p_ in probes:
        check(kept_prefill[p_], p_, f"prefill position {p_}")
    for i in (1, 2, 8, 16):
        check(kept_step[i], T - 1 + i, f"decode step {i}")
    # token evidence: every prefill position (next-token argmax) and every decode step vs the fp32 oracle, on confident rows
    top2 = l32.topk(2, dim=-1).values
    gap = top2[..., 0] - top2[..., 1]
    confident = gap > 3.0 * (l16 - l32).abs().amax(-1)
    hip_ids = torch.cat([hip_argmax_prefill, graphed[:, 1:n_steps + 1].cpu()], dim=1)       # argmax after each of the 665 contexts
    same = hip_ids == l32.argmax(-1)
    same16 = l16.argmax(-1) == l32.argmax(-1)
    frac = confident.float().mean().item()
    hit = l32.argmax(-1) == successor.cpu()[seq]
    report.update({"confident_fraction": frac, "greedy_agreement": same.float().mean().item(),
                   "greedy_agreement_bf16_oracle": same16.float().mean().item(),
                   "hip_differs_on_confident": int((~same & confident).sum()),
                   "fp32_oracle_emits_the_successor": hit.float().mean().item()})
    print(f"[14B full depth] confident {frac:.3f}, argmax agreement with the fp32 oracle: hip {report['greedy_agreement']:.3f} / bf16 oracle "
          f"{report['greedy_agreement_bf16_oracle']:.3f} over {same.numel()} positions; oracle {oracle_s:.1f} s")
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    json.dump(report, open(os.path.join(out, "llama14b_depth_parity.json"), "w"), indent=1)
    assert frac >= 0.5, f"only {frac:.3f} of the positions are confident: the peaked state dict is not peaked enough at 14B depth"
    assert (same | ~confident).all(), f"{(~same & confident).sum().item()} argmax ids differ from the fp32 oracle on confident rows"
    assert (hit | ~confident).float().mean().item() > 0.98
    assert (~same).sum() <= (~same16).sum() * 1.25 + 3, ((~same).sum().item(), (~same16).sum().item())


def test_persistent_decode_layers_are_bit_identical():
    """Devtools build, seedmi_set_option("decode_persistent", 1): all decoder layers of a decode step in one persistent launch (grid barriers instead of
    kernel boundaries, llama_xformer.py:280-332 per layer).  Same tile functions as the per-phase launches, so logits, the KV cache and
    the greedy tokens of the eager loop and of the hipGraph replay must be EQUAL, at 8B width (2 layers, B = 32)."""
    from dataclasses import replace
    from seed_amd import lib as L
    lib = L.load()
    if lib.seedmi_set_option(b"decode_persistent", 0) != 0:
        pytest.skip("the persistent decode kernel lives in the devtools build (SEEDMI_LIB_PATH=seed_amd/libseedmi_dev.so): measured slower")
    cfg = replace(C.LLAMA_8B, layers=2)
    sd = make_llama_state_dict(cfg, seed=3, dtype=torch.bfloat16, norm_jitter=0.05)
    B, T0, n_new = 32, 21, 7
    prompt = torch.randint(3, 32000, (B, T0), generator=torch.Generator().manual_seed(5)).cuda()
    eng = LlamaEngine(sd, cfg, device="cuda", batch_cap=B, tmax=64)
    out = {}
    try:
        for mode in (0, 1):
            L.check(lib.seedmi_set_option(b"decode_persistent", mode), "opt")
            eng.reset()
            lg = eng.forward(prompt, last_only=True)
            steps = [lg[:, 0].clone()]
            tok = lg[:, 0].float().argmax(-1, keepdim=True)
            for _ in range(4):
                lg = eng.forward(tok, last_only=True)
                steps.append(lg[:, 0].clone())
                tok = lg[:, 0].float().argmax(-1, keepdim=True)
            torch.cuda.synchronize()
            kc = eng.k_cache[1][:B, :, :T0 + 4].clone()
            vc = eng.v_cache[0][:B, :, :T0 + 4].clone()
            eager, _ = eng.greedy_decode(prompt, n_new)
            graphed = eng.greedy_decode_graph(prompt, n_new).clone()
            torch.cuda.synchronize()
            out[mode] = (torch.stack(steps), kc, vc, eager.clone(), graphed)
    finally:
        lib.seedmi_set_option(b"decode_persistent", 0)
    for a, b, what in zip(out[0], out[1], ("logits", "k cache", "v cache", "eager tokens", "graph tokens")):
        assert torch.equal(a, b), what
    assert torch.equal(out[1][3], out[1][4])


def test_replay_beyond_the_captured_steps_is_refused():
    """replay(k) past the capture's capacity would index the cache / uniforms / out by an unchecked device counter."""
    from seed_amd import lib as L
    cfg = C.LLAMA_TINY
    sd = make_llama_state_dict(cfg, seed=9, norm_jitter=0.05)
    eng = LlamaEngine(sd, cfg, device="cuda", batch_cap=2, tmax=32)
    ids = torch.randint(3, cfg.vocab, (2, 5), generator=torch.Generator().manual_seed(4)).cuda()
    eng.reset()
    first = eng.forward(ids, last_only=True)[:, 0].float().argmax(-1, keepdim=True)
    replay, out = eng.capture_decode_graph(first, 4)
    replay(3)
    with pytest.raises(L.SeedmiError):
        replay(1)
    # positions beyond the RoPE table are clamped by the kernels instead of read out of bounds (reference: device assert)
    eng.reset()
    pos = torch.full((2, 5), cfg.max_pos + 1000, dtype=torch.int64, device="cuda")
    lg = eng.forward(ids, position_ids=pos)
    torch.cuda.synchronize()
    assert torch.isfinite(lg.float()).all()


def test_inputs_embeds_and_hidden_states_match_the_oracle():
    """LlamaModel.forward's inputs_embeds / output_hidden_states (llama_xformer.py:502-544, 569-570, 613-617) through
    seedmi_llama_forward_io: prefill (MFMA GEMM path, M > 64) and a cached decode step (folded weight-streaming path)."""
    cfg = C.LLAMA_TINY
    sd = make_llama_state_dict(cfg, seed=3, norm_jitter=0.05)
    ids = torch.randint(3, cfg.vocab, (8, 12), generator=torch.Generator().manual_seed(6))
    eng = LlamaEngine(sd, cfg, device="cuda", batch_cap=8, tmax=64)
    emb = sd["model.embed_tokens.weight"].to(torch.bfloat16)[ids]
    hs = []
    a = eng.forward(ids.cuda(), hidden_states_out=hs)
    eng.reset()
    hs_e = []
    b = eng.forward(None, inputs_embeds=emb.cuda(), hidden_states_out=hs_e)
    torch.cuda.synchronize()
    assert torch.equal(a, b) and all(torch.equal(x, y) for x, y in zip(hs, hs_e))
    ref_h = []
    ref, past = O.llama_forward(sd, cfg, ids, mode="fp32", hidden_out=ref_h)
    assert len(hs) == cfg.layers + 1
    assert torch.equal(hs[0].cpu(), emb)                                                    # embedding output first
    for i, (x, y) in enumerate(zip(hs, ref_h)):
        assert _rel(x.float(), y) < 2e-2, (i, _rel(x.float(), y))
    # decode step with hidden states (T = 1)
    hs1 = []
    tok = ref[:, -1].argmax(-1, keepdim=True)
    eng.forward(tok.cuda(), hidden_states_out=hs1)
    ref_h1 = []
    O.llama_forward(sd, cfg, tok, past=past, mode="fp32", hidden_out=ref_h1)
    torch.cuda.synchronize()
    for i, (x, y) in enumerate(zip(hs1, ref_h1)):
        assert _rel(x.float(), y) < 2e-2, (i, _rel(x.float(), y))


def test_llama8b_full_size_properties():
    """BASELINE.json's decode configuration at full size (SEED-LLaMA-8B dims, batch 32, prompt 59 with a 32-code image span)
    through size-independent properties - the oracle needs minutes per step at this size: the hipGraph-replayed decode
    (folded RMSNorm, fragment-major weights, fused attention, device-side token selection) gives exactly the tokens of the
    eager loop, rows are independent of their batch neighbours (pure map over sequences = the replica-parallel property),
    and prefill logits of a row do not depend on the batch it sits in beyond bf16 accumulation order (bit-identical here,
    because every kernel on the path is batch-row independent)."""
    cfg = C.LLAMA_8B
    sd = make_llama_state_dict(cfg, seed=0, device="cuda", dtype=torch.bfloat16)
    B, T0, n_new = 32, 59, 6
    eng = LlamaEngine(sd, cfg, device="cuda", batch_cap=B, tmax=128)
    del sd
    g = torch.Generator(device="cuda").manual_seed(99)
    prompt = torch.randint(3, 32000, (B, T0), device="cuda", generator=g)
    prompt[:, 0] = 1
    prompt[:, 10:42] = 32000 + torch.randint(0, 8192, (B, 32), device="cuda", generator=g)
    eager, logits = eng.greedy_decode(prompt, n_new)
    graphed = eng.greedy_decode_graph(prompt, n_new).clone()
    torch.cuda.synchronize()
    assert torch.isfinite(logits.float()).all()
    assert int(eager.min()) >= 0 and int(eager.max()) < cfg.vocab
    assert torch.equal(eager, graphed)
    assert eng.past_len == T0 + n_new - 1
    # a 5-row slice run on its own reproduces those rows' tokens
    sub = eng.greedy_decode_graph(prompt[7:12].contiguous(), n_new)
    torch.cuda.synchronize()
    assert torch.equal(sub, graphed[7:12])


def test_split_k_error_word_is_sticky_and_reported():
    """the split-K decode GEMM records a partner that never arrived in the last flag word of its workspace.  The per-step
    clear must leave that word alone (it used to zero it with the hand-off flags), seedmi_llama_decode_status must report it after the
    loop, clear it once reported, and a healthy loop must report nothing."""
    import ctypes as Ct
    from dataclasses import replace
    from seed_amd import lib as L
    cfg = replace(C.LLAMA_8B, layers=1)
    sd = make_llama_state_dict(cfg, seed=1, dtype=torch.bfloat16)
    B = 4
    eng = LlamaEngine(sd, cfg, device="cuda", batch_cap=B, tmax=64)
    prompt = torch.randint(3, 32000, (B, 7), generator=torch.Generator().manual_seed(1)).cuda()
    toks = eng.greedy_decode_graph(prompt, 5)                               # ends with decode_status(): healthy -> no error
    torch.cuda.synchronize()
    ws = eng._ws
    # poison the sticky word the way a timed-out owner would (1 + blockIdx): the split-K area is the FIRST part of the workspace
    # (include/seedmi.h), its flag words come first and the error word is the last of the 1024
    view = ws.view(torch.int32)
    word = 1023
    assert int(view[word]) == 0
    view[word] = 7
    assert eng.lib.seedmi_llama_decode_status(Ct.byref(eng.w), B, L.ptr(ws), ws.numel(), L.stream_ptr()) != 0
    assert "workgroup 6" in eng.lib.seedmi_last_error().decode()
    assert eng.lib.seedmi_llama_decode_status(Ct.byref(eng.w), B, L.ptr(ws), ws.numel(), L.stream_ptr()) == 0     # cleared once reported
    view[word] = 3
    tok = toks[:, -1:].contiguous()
    for _ in range(3):
        eng.forward(tok, last_only=True)                                    # each step clears the hand-off flags, NOT the error word
    torch.cuda.synchronize()
    assert int(view[word]) == 3
    with pytest.raises(L.SeedmiError, match="gave up waiting"):
        eng.decode_status(B)
    eng.decode_status(B)
    # a workspace that never saw seedmi_llama_workspace_init (here: zeroed only) is refused by the status call instead of
    # having its error word's bytes read as a verdict
    raw = torch.zeros_like(ws)
    assert eng.lib.seedmi_llama_decode_status(Ct.byref(eng.w), B, L.ptr(raw), raw.numel(), L.stream_ptr()) == -1
    assert "never initialised" in eng.lib.seedmi_last_error().decode()
    L.check(eng.lib.seedmi_llama_workspace_init(L.ptr(raw), raw.numel(), L.stream_ptr()), "init")
    assert eng.lib.seedmi_llama_decode_status(Ct.byref(eng.w), B, L.ptr(raw), raw.numel(), L.stream_ptr()) == 0


@pytest.mark.parametrize("B", [5, 32])
def test_llama_workspace_needs_only_its_flag_area_initialised(B):
    """workspaces are torch.empty + seedmi_llama_workspace_init since round 5 (which writes the 1022 flag words, the tag and
    the error word only).  Poison every byte with 0xFF (NaN bit patterns in every 16-bit and fp32 region, incl. the padded fragment rows
    beyond the batch) before the init: prefill logits, every decode step's logits, the tokens and the status must be BIT-equal to a run
    on a zeroed-then-initialised workspace - i.e. include/seedmi.h's contract (only the flag area must be initialised) is what the code
    needs.  B = 5 pads 11 of 16 fragment rows; B = 32 takes the split-K hand-off with cut tiles."""
    from dataclasses import replace
    from seed_amd import lib as L
    cfg = replace(C.LLAMA_8B, layers=2)
    sd = make_llama_state_dict(cfg, seed=1, dtype=torch.bfloat16)
    prompt = torch.randint(3, 32000, (B, 9), generator=torch.Generator().manual_seed(2)).cuda()
    runs = []
    for poison in (None, 0xFF):
        eng = LlamaEngine(sd, cfg, device="cuda", batch_cap=B, tmax=64)
        made = []

        def new_workspace(nbytes, eng=eng, poison=poison, made=made):
            ws = torch.zeros(nbytes, dtype=torch.uint8, device="cuda") if poison is None else torch.full((nbytes,), poison, dtype=torch.uint8, device="cuda")
            L.check(eng.lib.seedmi_llama_workspace_init(L.ptr(ws), ws.numel(), L.stream_ptr()), "init")
            made.append(ws)
            return ws
        eng.new_workspace = new_workspace
        toks, steps = eng.greedy_decode(prompt, 6)                 # eager prefill + 5 decode steps, ends with decode_status()
        toks_g = eng.greedy_decode_graph(prompt, 6)                # and the captured step replayed from a hipGraph on the same workspace
        torch.cuda.synchronize()
        assert made, "the engine did not allocate through new_workspace"
        assert torch.isfinite(steps.float()).all()
        runs.append((toks.cpu(), steps.cpu(), toks_g.cpu()))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][2], runs[1][2])
    assert torch.equal(runs[0][1].view(torch.int16), runs[1][1].view(torch.int16)), "logits depend on uninitialised workspace bytes"


def test_padded_attention_mask_warns(tmp_path):
    """VERDICT r3 weak 12 / SURVEY H7: padding masks are not applied (the reference's xformers path ignores them too); a mask that
    contains zeros now says so instead of silently answering for padding.  A mask of ones stays silent."""
    import warnings as W
    from models.llama_xformer import LlamaForCausalLM
    from transformers.models.llama.configuration_llama import LlamaConfig as HFConfig
    cfg = C.LLAMA_TINY
    hf = HFConfig(vocab_size=cfg.vocab, hidden_size=cfg.hidden, intermediate_size=cfg.ffn, num_hidden_layers=cfg.layers,
                  num_attention_heads=cfg.heads, rms_norm_eps=cfg.rms_eps, max_position_embeddings=cfg.max_pos)
    model = LlamaForCausalLM(hf)
    model.load_state_dict(make_llama_state_dict(cfg, seed=2), strict=True)
    model = model.eval().to("cuda")
    ids = torch.randint(3, cfg.vocab, (2, 6), generator=torch.Generator().manual_seed(0)).cuda()
    with W.catch_warnings():
        W.simplefilter("error")
        model(input_ids=ids, attention_mask=torch.ones_like(ids))
    mask = torch.ones_like(ids)
    mask[1, :2] = 0
    with pytest.warns(RuntimeWarning, match="padding masks are not applied"):
        model(input_ids=ids, attention_mask=mask)
