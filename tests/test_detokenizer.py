"""De-tokenizer front half: Blip2QformerQuantizer.get_codebook_entry, ids -> unCLIP image embeds.

CPU: the oracle restatement against vectors from the reference's own modules (tests/golden/detok_*.npz, written by
oracle/make_golden.py).  GPU: seedmi_detokenize through the C ABI against the oracle and the same golden vectors.
Tolerances (floating point, bf16 path): the HIP path must sit as close to the oracle's bf16 choreography as the
reference's own bf16 CPU run does, and both must track the fp32 ground truth to bf16 accumulation noise.
"""
import os

import numpy as np
import pytest
import torch

from oracle import seed_oracle as O
from seed_amd import config as C
from seed_amd.weights import make_detokenizer_state_dict

CASES = [("tiny", C.TINY), ("full", C.SEED2)]


def _rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return ((a - b).norm() / b.norm()).item()


def _case(golden_dir, name, cfg):
    g = np.load(os.path.join(golden_dir, f"detok_{name}.npz"))
    sd = make_detokenizer_state_dict(cfg, seed=int(g["seed_w"]))
    return g, sd, torch.from_numpy(g["ids"])


@pytest.mark.parametrize("name,cfg", CASES)
def test_oracle_fp32_matches_reference_modules(golden_dir, name, cfg):
    g, sd, ids = _case(golden_dir, name, cfg)
    taps = {}
    out = O.get_codebook_entry(sd, ids, cfg, "fp32", taps)
    assert tuple(out.shape) == (ids.shape[0], cfg.image_features_dim)
    assert _rel(out, g["embeds_fp32"]) < 2e-6
    assert _rel(taps["hidden"][:, :4, :64], g["hidden_fp32_slice"]) < 2e-6


@pytest.mark.parametrize("name,cfg", CASES)
def test_oracle_bf16_tracks_reference_bf16(golden_dir, name, cfg):
    g, sd, ids = _case(golden_dir, name, cfg)
    out = O.get_codebook_entry(sd, ids, cfg, "bf16")
    e_ref = _rel(out, g["embeds_bf16"])                # restatement vs the reference modules run natively in bf16
    e_32 = _rel(g["embeds_bf16"], g["embeds_fp32"])    # the reference's own bf16 error
    assert e_ref < 1e-2 and e_ref < e_32, (e_ref, e_32)


@pytest.mark.parametrize("name,cfg", CASES)
def test_oracle_fp16_tracks_reference_fp16(golden_dir, name, cfg):
    """The reference's shipped setting: the de-tokenizer modules .half()'ed (seed_llama_tokenizer.py:62-63).  The restatement in Prec("fp16")
    against the reference modules run natively in fp16 (tests/golden/detok_*_fp16.npz, oracle/make_golden.py::detok_golden_fp16)."""
    g, sd, ids = _case(golden_dir, name, cfg)
    g16 = np.load(os.path.join(golden_dir, f"detok_{name}_fp16.npz"))
    assert np.array_equal(g16["ids"], g["ids"]) and int(g16["seed_w"]) == int(g["seed_w"])
    taps = {}
    out = O.get_codebook_entry(sd, ids, cfg, "fp16", taps)
    e_ref = _rel(out, g16["embeds_fp16"])
    e_32 = _rel(g16["embeds_fp16"], g["embeds_fp32"])  # the reference's own fp16 error
    assert e_ref < 2e-3 and e_ref < 1.5 * e_32, (e_ref, e_32)
    assert _rel(taps["hidden"][:, :4, :64], g16["hidden_fp16_slice"]) < 2e-3


def test_state_dict_has_reference_key_names():
    sd = make_detokenizer_state_dict(C.TINY)
    for k in ("quantize.embedding.weight", "decode_task_layer.0.weight", "decode_task_layer.2.bias", "pos_embed_image",
              "blocks_image.0.norm1.weight", "blocks_image.1.attn.qkv.bias", "blocks_image.1.mlp.fc2.weight",
              "image_down.0.weight", "image_down.2.weight", "image_down.4.weight", "distill_image_proj.bias"):
        assert k in sd, k
    assert "image_down.0.bias" not in sd           # bias=False, qformer_quantizer.py:280-284


# ------------------------------------------------------------------------------------------------ GPU (C ABI)


@pytest.mark.gpu
@pytest.mark.parametrize("name,cfg", CASES)
def test_detokenize_matches_oracle_and_reference_golden(golden_dir, name, cfg):
    from seed_amd.detokenizer_engine import DetokenizerEngine
    g, sd, ids = _case(golden_dir, name, cfg)
    eng = DetokenizerEngine(sd, cfg, device="cuda")
    taps = {}
    out = eng.codebook_entry(ids.cuda(), taps)
    torch.cuda.synchronize()
    assert out.dtype == torch.bfloat16 and tuple(out.shape) == (ids.shape[0], cfg.image_features_dim)
    o_taps = {}
    ref16 = O.get_codebook_entry(sd, ids, cfg, "bf16", o_taps)
    e_oracle = _rel(out.float(), ref16)
    e_hidden = _rel(taps["hidden"].float(), o_taps["hidden"])
    e_fp32 = _rel(out.float(), g["embeds_fp32"])
    e_ref_bf16 = _rel(g["embeds_bf16"], g["embeds_fp32"])
    # same rounding points as the oracle: only fp32 accumulation order and the fast erf differ
    assert e_hidden < 8e-3, e_hidden
    assert e_oracle < 8e-3, e_oracle
    # and no further from the fp32 ground truth than the reference's own bf16 run (x1.5 slack)
    assert e_fp32 < 1.5 * e_ref_bf16 + 1e-3, (e_fp32, e_ref_bf16)


@pytest.mark.gpu
def test_detokenize_batch_is_a_pure_map_and_edge_shapes(golden_dir):
    """Each image's embeds depend only on its own 32 ids: rows of a ragged batch equal the single-image results bit
    for bit; a 1-D id vector is accepted like the reference's indexing; the extreme ids 0 and n_embed-1 are valid."""
    from seed_amd.detokenizer_engine import DetokenizerEngine
    cfg = C.TINY
    sd = make_detokenizer_state_dict(cfg)
    eng = DetokenizerEngine(sd, cfg, device="cuda")
    gen = torch.Generator().manual_seed(3)
    ids = torch.randint(0, cfg.n_embed, (5, cfg.n_query), generator=gen)
    ids[0, :] = 0
    ids[1, :] = cfg.n_embed - 1
    ids = ids.cuda()
    full = eng.codebook_entry(ids)
    for i in range(5):
        one = eng.codebook_entry(ids[i])
        assert torch.equal(one[0], full[i]), i
    ref = O.get_codebook_entry(sd, ids.cpu(), cfg, "bf16")
    assert _rel(full.float(), ref) < 6e-3
    with pytest.raises(ValueError):
        eng.codebook_entry(ids[:, :5])


@pytest.mark.gpu
@pytest.mark.parametrize("name,cfg", CASES)
def test_detokenize_fp16_matches_fp16_oracle_and_reference_golden(golden_dir, name, cfg):
    """The fp16 build (libseedmi_f16.so) of the same path against the fp16 oracle and the reference modules' native-fp16 run."""
    from seed_amd.detokenizer_engine import DetokenizerEngine
    g, sd, ids = _case(golden_dir, name, cfg)
    g16 = np.load(os.path.join(golden_dir, f"detok_{name}_fp16.npz"))
    eng = DetokenizerEngine(sd, cfg, device="cuda", dtype=torch.float16)
    taps = {}
    out = eng.codebook_entry(ids.cuda(), taps)
    torch.cuda.synchronize()
    assert out.dtype == torch.float16 and tuple(out.shape) == (ids.shape[0], cfg.image_features_dim)
    o_taps = {}
    ref16 = O.get_codebook_entry(sd, ids, cfg, "fp16", o_taps)
    e_oracle, e_hidden = _rel(out.float(), ref16), _rel(taps["hidden"].float(), o_taps["hidden"])
    e_gold = _rel(out.float(), g16["embeds_fp16"])
    e_fp32, e_ref_fp16 = _rel(out.float(), g["embeds_fp32"]), _rel(g16["embeds_fp16"], g["embeds_fp32"])
    print(f"[detok fp16 {name}] vs fp16 oracle {e_oracle:.2e} (hidden {e_hidden:.2e}), vs reference fp16 run {e_gold:.2e}, "
          f"vs fp32 {e_fp32:.2e} (reference fp16 vs fp32 {e_ref_fp16:.2e})")
    assert e_hidden < 2e-3 and e_oracle < 2e-3, (e_hidden, e_oracle)
    assert e_fp32 < 1.5 * e_ref_fp16 + 2e-4, (e_fp32, e_ref_fp16)
    # 8x closer to fp32 than the bf16 build's bound: the point of the fp16 build
    assert e_fp32 < 0.25 * _rel(g["embeds_bf16"], g["embeds_fp32"])


@pytest.mark.gpu
def test_relu_epilogue_and_dropin_surface(golden_dir):
    """image_down's bias-free Linear+ReLU epilogue on its own, and the reference-shaped call surface
    (Blip2QformerQuantizer.get_codebook_entry, ImageTokenizer.decode_embeds / decode)."""
    from seed_amd import lib as L
    lib = L.load()
    g = torch.Generator(device="cuda").manual_seed(1)
    A = torch.randn(96, 128, device="cuda", generator=g).bfloat16()
    W = (torch.randn(64, 128, device="cuda", generator=g) * 0.1).bfloat16()
    Cm = torch.empty(96, 64, device="cuda", dtype=torch.bfloat16)
    L.check(lib.seedmi_gemm_bf16(96, 64, 128, L.ptr(A), 128, L.ptr(W), 128, None, None, 0, L.EPI_RELU, L.ptr(Cm), 64, 0, 0,
                                 L.stream_ptr()), "gemm relu")
    torch.cuda.synchronize()
    ref = torch.relu((A.float() @ W.float().t()).bfloat16().float())
    assert torch.allclose(Cm.float(), ref, atol=2e-2, rtol=2e-2)
    assert (Cm.float() >= 0).all()

    from models.seed_qformer.qformer_quantizer import Blip2QformerQuantizer
    cfg = C.TINY
    sd = make_detokenizer_state_dict(cfg)
    model = Blip2QformerQuantizer.from_pretrained(sd, cfg=cfg, device="cuda").eval().half()
    ids = torch.randint(0, cfg.n_embed, (2, cfg.n_query)).cuda()
    emb = model.get_codebook_entry(ids)
    assert tuple(emb.shape) == (2, cfg.image_features_dim)
    assert emb.dtype == torch.float16                    # .half(): the fp16 build computes it (round 5)
    assert _rel(emb.float(), O.get_codebook_entry(sd, ids.cpu(), cfg, "fp16")) < 2e-3
    emb_bf = model.bfloat16().get_codebook_entry(ids)
    assert emb_bf.dtype == torch.bfloat16 and _rel(emb_bf.float(), O.get_codebook_entry(sd, ids.cpu(), cfg, "bf16")) < 6e-3
