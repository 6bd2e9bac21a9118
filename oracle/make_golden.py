"""Generate tests/golden/*.npz from the reference's OWN modules (run in the build container only).

    python -m oracle.make_golden

TEST INFRASTRUCTURE.  Imports /root/reference through oracle/ref_shims.py, loads the seeded
synthetic state dicts of seed_amd/weights.py into the reference modules (which also proves the
state-dict key names/shapes match the reference's), runs them on CPU in fp32 and in bf16, and stores
inputs' seeds + outputs.  The committed vectors are what pins oracle/seed_oracle.py on the GPU box,
where /root/reference does not exist.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_shims  # noqa: E402
from seed_amd import config as C  # noqa: E402
from seed_amd.weights import (make_tokenizer_state_dict, make_llama_state_dict, calibrate_codebook,  # noqa: E402
                              make_detokenizer_state_dict, make_tokenizer_peaked_state_dict, peaked_codebook, PEAKED_CASE,
                              peaked_case_images)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_tokenizer_weights(mods, sd):
    ve = {k[len("visual_encoder."):]: v for k, v in sd.items() if k.startswith("visual_encoder.")}
    missing, unexpected = mods.visual_encoder.load_state_dict(ve, strict=True)
    mods.ln_vision.load_state_dict({"weight": sd["ln_vision.weight"], "bias": sd["ln_vision.bias"]}, strict=True)
    qf = {k[len("Qformer."):]: v for k, v in sd.items() if k.startswith("Qformer.")}
    res = mods.Qformer.load_state_dict(qf, strict=False)
    # every key we generate must exist in the reference module; the reference may hold extras
    # (embeddings.position_ids buffer) that the encode path never reads
    assert not res.unexpected_keys, res.unexpected_keys
    leftover = [k for k in res.missing_keys if "position_ids" not in k]
    assert not leftover, leftover
    mods.encode_task_layer.load_state_dict(
        {k[len("encode_task_layer."):]: v for k, v in sd.items() if k.startswith("encode_task_layer.")}, strict=True)
    mods.quantize.embedding.weight.data.copy_(sd["quantize.embedding.weight"])


def tokenizer_golden(name, cfg, batch, seed_w, seed_x, ref):
    torch.manual_seed(0)
    sd = make_tokenizer_state_dict(cfg, seed=seed_w, ln_jitter=0.05)
    gen = torch.Generator().manual_seed(seed_x)
    image = torch.randn(batch, 3, cfg.img_size, cfg.img_size, generator=gen)
    mods = ref_shims.build_reference_tokenizer_modules(ref, cfg)
    load_tokenizer_weights(mods, sd)
    qt = sd["query_tokens"].clone()
    # pass 1 (reference init codebook) only to obtain z; then a codebook with realistic margins
    _, taps = ref_shims.reference_get_codebook_indices(mods, qt, image)
    cb = calibrate_codebook(taps["z"], cfg.n_embed, seed=7)
    sd["quantize.embedding.weight"] = cb
    mods.quantize.embedding.weight.data.copy_(cb)
    ids32, taps32 = ref_shims.reference_get_codebook_indices(mods, qt, image)

    # native bf16 run of the same reference modules (CPU: maybe_autocast is a nullcontext, every op in bf16)
    for m in (mods.visual_encoder, mods.Qformer, mods.quantize, mods.encode_task_layer):
        m.bfloat16()
    # blip2.LayerNorm feeds fp32 activations into F.layer_norm (blip2.py:183); on CPU that needs fp32
    # parameters (GPU autocast upcasts them), so hold the bf16-rounded values in fp32.
    for prm in mods.ln_vision.parameters():
        prm.data = prm.data.bfloat16().float()
    try:
        ids16, taps16 = ref_shims.reference_get_codebook_indices(mods, qt.bfloat16(), image.bfloat16())
        have16 = True
    except Exception as e:  # pragma: no cover
        print("bf16 reference run failed:", e)
        have16 = False

    out = dict(cfg=np.array(repr(cfg.to_dict())), seed_w=seed_w, seed_x=seed_x, batch=batch, ln_jitter=0.05,
               codebook=cb.numpy(), image_sum=np.float64(image.double().sum().item()),
               ids_fp32=ids32.numpy(), z_fp32=taps32["z"].numpy(), qformer_out_fp32=taps32["qformer_out"].numpy(),
               image_embeds_fp32_slice=taps32["image_embeds"][:, :8, :64].numpy(),
               image_embeds_fp32_absmean=np.float64(taps32["image_embeds"].abs().mean().item()))
    if have16:
        out.update(ids_bf16=ids16.numpy(), z_bf16=taps16["z"].float().numpy(),
                   qformer_out_bf16=taps16["qformer_out"].float().numpy(),
                   image_embeds_bf16_slice=taps16["image_embeds"][:, :8, :64].float().numpy())
    np.savez_compressed(os.path.join(GOLDEN, f"tokenizer_{name}.npz"), **out)
    print(name, "ids[0,:8] fp32", ids32[0, :8].tolist(), "bf16", ids16[0, :8].tolist() if have16 else None,
          "agree", (ids32 == ids16).float().mean().item() if have16 else None)


def tokenizer_golden_fp16(name, cfg, batch, seed_w, seed_x, ref, ln_jitter=0.05, full=False):
    """The reference's own modules in their SHIPPED precision: native fp16 (`fp16: True`, configs/tokenizer/seed_llama_tokenizer_hf.yaml:3;
    model.half() / img.half(), models/seed_llama_tokenizer.py:58-59,86-87; on CPU maybe_autocast is a nullcontext, blip2.py:45, so every op
    runs in fp16 with ln_vision's fp32 island kept).  Same weights, images and calibrated codebook as the fp32 / bf16 golden of that name
    (the codebook is a function of the fp32 run's z), stored as tokenizer_<name>_fp16.npz next to it."""
    torch.manual_seed(0)
    sd = make_tokenizer_state_dict(cfg, seed=seed_w, ln_jitter=ln_jitter) if ln_jitter else make_tokenizer_state_dict(cfg, seed=seed_w)
    image = torch.randn(batch, 3, cfg.img_size, cfg.img_size, generator=torch.Generator().manual_seed(seed_x))
    mods = ref_shims.build_reference_tokenizer_modules(ref, cfg)
    load_tokenizer_weights(mods, sd)
    qt = sd["query_tokens"].clone()
    _, taps = ref_shims.reference_get_codebook_indices(mods, qt, image)
    cb = calibrate_codebook(taps["z"], cfg.n_embed, seed=7)
    mods.quantize.embedding.weight.data.copy_(cb)
    ids32, taps32 = ref_shims.reference_get_codebook_indices(mods, qt, image)
    for m in (mods.visual_encoder, mods.Qformer, mods.quantize, mods.encode_task_layer):
        m.half()
    for prm in mods.ln_vision.parameters():
        prm.data = prm.data.half().float()
    ids16, taps16 = ref_shims.reference_get_codebook_indices(mods, qt.half(), image.half())
    out = dict(seed_w=seed_w, seed_x=seed_x, batch=batch, ln_jitter=ln_jitter, image_sum=np.float64(image.double().sum().item()),
               ids_fp32=ids32.numpy().astype(np.int16), ids_fp16=ids16.numpy().astype(np.int16), z_fp32=taps32["z"].numpy(),
               z_fp16=taps16["z"].float().numpy(), qformer_out_fp16=taps16["qformer_out"].float().numpy(),
               image_embeds_fp16_slice=taps16["image_embeds"][:, :8, :64].float().numpy(),
               image_embeds_fp32_slice=taps32["image_embeds"][:, :8, :64].numpy())
    if not full:
        out["codebook"] = cb.numpy()
    np.savez_compressed(os.path.join(GOLDEN, f"tokenizer_{name}_fp16.npz"), **out)
    print(name, "fp16: ids[0,:8] fp32", ids32[0, :8].tolist(), "fp16", ids16[0, :8].tolist(), "reference fp16 vs fp32 id agreement",
          (ids32 == ids16).float().mean().item(), "z rel", ((taps16["z"].float() - taps32["z"]).norm() / taps32["z"].norm()).item())


def tokenizer_golden_full(ref, batch=16, seed_w=0, seed_x=1234):
    """The full SEED-2 tokenizer (EVA-ViT-g/14, 39 blocks + 12-layer Q-Former + 8192 x 32 codebook) through the reference's own
    modules, fp32 and native bf16, on ``batch`` images: pins the oracle AND the HIP path at the real size.
    The codebook is calibrate_codebook(z_fp32, seed 7) - a pure function of the stored z - so it is not stored (1 MB)."""
    cfg = C.SEED2
    torch.manual_seed(0)
    sd = make_tokenizer_state_dict(cfg, seed=seed_w)
    image = torch.randn(batch, 3, cfg.img_size, cfg.img_size, generator=torch.Generator().manual_seed(seed_x))
    mods = ref_shims.build_reference_tokenizer_modules(ref, cfg)
    load_tokenizer_weights(mods, sd)
    qt = sd["query_tokens"].clone()
    _, taps = ref_shims.reference_get_codebook_indices(mods, qt, image)
    cb = calibrate_codebook(taps["z"], cfg.n_embed, seed=7)
    mods.quantize.embedding.weight.data.copy_(cb)
    ids32, taps32 = ref_shims.reference_get_codebook_indices(mods, qt, image)
    assert torch.equal(taps32["z"], taps["z"])
    for m in (mods.visual_encoder, mods.Qformer, mods.quantize, mods.encode_task_layer):
        m.bfloat16()
    for prm in mods.ln_vision.parameters():
        prm.data = prm.data.bfloat16().float()
    ids16, taps16 = ref_shims.reference_get_codebook_indices(mods, qt.bfloat16(), image.bfloat16())
    np.savez_compressed(os.path.join(GOLDEN, "tokenizer_full.npz"), seed_w=seed_w, seed_x=seed_x, batch=batch,
                        image_sum=np.float64(image.double().sum().item()), ids_fp32=ids32.numpy().astype(np.int16),
                        ids_bf16=ids16.numpy().astype(np.int16), z_fp32=taps32["z"].numpy(),
                        z_bf16=taps16["z"].float().numpy().astype(np.float32),
                        image_embeds_fp32_slice=taps32["image_embeds"][:, :4, :32].numpy(),
                        image_embeds_bf16_slice=taps16["image_embeds"][:, :4, :32].float().numpy())
    print("full ids[0,:8] fp32", ids32[0, :8].tolist(), "bf16", ids16[0, :8].tolist(), "fp32 vs bf16 agreement of the reference "
          "with itself", (ids32 == ids16).float().mean().item())


def vq_golden(ref):
    """VectorQuantizer2 alone on the reference module: z, codebook -> ids (fp32 and bf16)."""
    gen = torch.Generator().manual_seed(11)
    vq = ref.quantizer.VectorQuantizer2(8192, 32, beta=0.25)
    cb = torch.randn(8192, 32, generator=gen) * 0.3
    z = cb[torch.randint(0, 8192, (4, 32), generator=gen)] + torch.randn(4, 32, 32, generator=gen) * 0.05
    # force exact ties: duplicate code rows (first index must win) and a z that sits exactly on a code
    cb[4001] = cb[17]
    cb[7000] = cb[17]
    z[0, 0] = cb[17]
    z[0, 1] = cb[4001]
    vq.embedding.weight.data.copy_(cb)
    with torch.no_grad():
        _, _, ids32 = vq(z)
        vq.bfloat16()
        _, _, ids16 = vq(z.bfloat16())
    np.savez_compressed(os.path.join(GOLDEN, "vq_reference.npz"), z=z.numpy(), codebook=cb.numpy(),
                        ids_fp32=ids32.reshape(4, 32).numpy(), ids_bf16=ids16.reshape(4, 32).numpy())
    print("vq golden ids", ids32[:4].tolist(), ids16[:4].tolist())


def llama_golden(ref):
    cfg = C.LLAMA_TINY
    from transformers.models.llama.configuration_llama import LlamaConfig as HFLlamaConfig
    hf = HFLlamaConfig(vocab_size=cfg.vocab, hidden_size=cfg.hidden, intermediate_size=cfg.ffn,
                       num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads, rms_norm_eps=cfg.rms_eps,
                       max_position_embeddings=cfg.max_pos, hidden_act="silu", pad_token_id=0)
    model = ref.llama.LlamaForCausalLM(hf).eval()
    sd = make_llama_state_dict(cfg, seed=3, norm_jitter=0.05)
    res = model.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    assert all("rotary_emb" in k for k in res.missing_keys), res.missing_keys
    gen = torch.Generator().manual_seed(5)
    B, T, n_new = 2, 12, 4
    ids = torch.randint(3, cfg.vocab, (B, T), generator=gen)
    outs = {}
    for tag, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        model.to(dt)
        with torch.no_grad():
            o = model(input_ids=ids, use_cache=True)
            logits = [o.logits]
            past = o.past_key_values
            tok = o.logits[:, -1].float().argmax(-1, keepdim=True)
            toks = [tok]
            for _ in range(n_new - 1):
                o = model(input_ids=tok, past_key_values=past, use_cache=True)
                past = o.past_key_values
                logits.append(o.logits)
                tok = o.logits[:, -1].float().argmax(-1, keepdim=True)
                toks.append(tok)
        outs[f"prefill_logits_{tag}"] = logits[0].float().numpy()
        outs[f"decode_logits_{tag}"] = torch.cat(logits[1:], dim=1).float().numpy()
        outs[f"tokens_{tag}"] = torch.cat(toks, dim=1).numpy()
        outs[f"k_cache0_{tag}"] = past[0][0].float().numpy()
    np.savez_compressed(os.path.join(GOLDEN, "llama_tiny.npz"), input_ids=ids.numpy(), seed_w=3, norm_jitter=0.05,
                        n_new=n_new, **outs)
    print("llama golden tokens", outs["tokens_fp32"].tolist(), outs["tokens_bf16"].tolist())


def detok_golden(name, cfg, batch, seed_w, seed_ids, ref):
    """Blip2QformerQuantizer.get_codebook_entry on the reference's own sub-modules (vit.Block, VectorQuantizer2, ...):
    ids -> image embeds, fp32 and natively-bf16 runs."""
    sd = make_detokenizer_state_dict(cfg, seed=seed_w)
    mods = ref_shims.build_reference_detokenizer_modules(ref, cfg)
    mods.load_state_dict(sd, strict=True)          # also proves the key names / shapes match the reference modules
    gen = torch.Generator().manual_seed(seed_ids)
    ids = torch.randint(0, cfg.n_embed, (batch, cfg.n_query), generator=gen)
    out32, hid32 = ref_shims.reference_get_codebook_entry(mods, ids)
    out16, hid16 = ref_shims.reference_get_codebook_entry(mods.bfloat16(), ids)
    np.savez_compressed(os.path.join(GOLDEN, f"detok_{name}.npz"), seed_w=seed_w, ids=ids.numpy(),
                        embeds_fp32=out32.numpy(), embeds_bf16=out16.float().numpy(),
                        hidden_fp32_slice=hid32[:, :4, :64].numpy(), hidden_bf16_slice=hid16[:, :4, :64].float().numpy())
    print("detok", name, "embeds[0,:4]", out32[0, :4].tolist(), "bf16 rel", ((out16.float() - out32).norm() / out32.norm()).item())


def detok_golden_fp16(name, cfg, batch, seed_w, seed_ids, ref):
    """The same ids and weights as detok_golden through the reference's sub-modules .half()'ed (seed_llama_tokenizer.py:62-63: the shipped
    setting), natively in fp16 on the CPU."""
    sd = make_detokenizer_state_dict(cfg, seed=seed_w)
    mods = ref_shims.build_reference_detokenizer_modules(ref, cfg)
    mods.load_state_dict(sd, strict=True)
    gen = torch.Generator().manual_seed(seed_ids)
    ids = torch.randint(0, cfg.n_embed, (batch, cfg.n_query), generator=gen)
    out32, _ = ref_shims.reference_get_codebook_entry(mods, ids)
    out16, hid16 = ref_shims.reference_get_codebook_entry(mods.half(), ids)
    assert out16.dtype == torch.float16
    np.savez_compressed(os.path.join(GOLDEN, f"detok_{name}_fp16.npz"), seed_w=seed_w, ids=ids.numpy(),
                        embeds_fp16=out16.float().numpy(), hidden_fp16_slice=hid16[:, :4, :64].float().numpy())
    print("detok fp16", name, "rel vs fp32", ((out16.float() - out32).norm() / out32.norm()).item())


def tokenizer_golden_peaked(ref):
    """The peaked full-size case (seed_amd/weights.py::make_tokenizer_peaked_state_dict) through the reference's OWN
    modules: calibration z in fp32 -> codebook rows 0..511 = those z; the evaluated images (calibration + 0.02 pixel noise) in fp32, native
    bf16 and native fp16.  Stored: z_cal (the codebook is a pure function of it), the three runs' ids and z, and per row the fp32 run's
    distance from its own code to the nearest OTHER code (what the margin gate of tests/test_gpu_tokenizer.py needs)."""
    cfg = C.SEED2
    P = PEAKED_CASE
    sd = make_tokenizer_peaked_state_dict(cfg, seed=P["seed_w"], value_gain=P["value_gain"], qk_gain=P["qk_gain"])
    cal, image = peaked_case_images(cfg)
    mods = ref_shims.build_reference_tokenizer_modules(ref, cfg)
    load_tokenizer_weights(mods, sd)
    qt = sd["query_tokens"].clone()
    _, taps_cal = ref_shims.reference_get_codebook_indices(mods, qt, cal)
    cb = peaked_codebook(taps_cal["z"], cfg.n_embed, seed=7)
    mods.quantize.embedding.weight.data.copy_(cb)
    ids32, taps32 = ref_shims.reference_get_codebook_indices(mods, qt, image)
    out = dict(z_cal=taps_cal["z"].numpy(), image_sum=np.float64(image.double().sum().item()), ids_fp32=ids32.numpy().astype(np.int16),
               z_fp32=taps32["z"].numpy(), **{k: np.float64(v) if isinstance(v, float) else np.int64(v) for k, v in P.items()})
    for tag, dt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
        mods = ref_shims.build_reference_tokenizer_modules(ref, cfg)
        load_tokenizer_weights(mods, sd)
        mods.quantize.embedding.weight.data.copy_(cb)
        for m in (mods.visual_encoder, mods.Qformer, mods.quantize, mods.encode_task_layer):
            m.to(dt)
        for prm in mods.ln_vision.parameters():
            prm.data = prm.data.to(dt).float()
        ids16, taps16 = ref_shims.reference_get_codebook_indices(mods, qt.to(dt), image.to(dt))
        out[f"ids_{tag}"] = ids16.numpy().astype(np.int16)
        out[f"z_{tag}"] = taps16["z"].float().numpy()
        print("peaked", tag, "reference vs its fp32 run: ids equal", (ids16 == ids32).float().mean().item(), "z rel",
              ((taps16["z"].float() - taps32["z"]).norm() / taps32["z"].norm()).item())
    want = torch.arange(P["batch"] * cfg.n_query).reshape(P["batch"], cfg.n_query)
    print("peaked fp32 ids == own calibration row:", (ids32 == want).float().mean().item())
    np.savez_compressed(os.path.join(GOLDEN, "tokenizer_peaked.npz"), **out)


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    torch.set_num_threads(os.cpu_count())
    ref = ref_shims.load_reference_modules()
    if sys.argv[1:] == ["peaked"]:
        return tokenizer_golden_peaked(ref)
    vq_golden(ref)
    tokenizer_golden("tiny", C.TINY, 3, 0, 1234, ref)
    tokenizer_golden("mid", C.MID, 2, 1, 4321, ref)
    llama_golden(ref)
    detok_golden("tiny", C.TINY, 3, 11, 5, ref)
    detok_golden("full", C.SEED2, 2, 12, 6, ref)
    detok_golden_fp16("tiny", C.TINY, 3, 11, 5, ref)
    detok_golden_fp16("full", C.SEED2, 2, 12, 6, ref)
    tokenizer_golden_full(ref)
    tokenizer_golden_fp16("tiny", C.TINY, 3, 0, 1234, ref)
    tokenizer_golden_fp16("mid", C.MID, 2, 1, 4321, ref)
    tokenizer_golden_fp16("full", C.SEED2, 16, 0, 1234, ref, ln_jitter=0.0, full=True)
    tokenizer_golden_peaked(ref)


if __name__ == "__main__":
    main()
