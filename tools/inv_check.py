#!/usr/bin/env python
"""Batch invariance in ten seconds (MID config): 40 images in one call, as 4 x 10 and as singles must give identical ids under every
option set that only chooses between kernels (the 64x64 kernel's tile order, the LayerNorm fold on / off).  The quick check behind every
small-M kernel edit of round 6 (the full statement is tests/test_gpu_tokenizer.py::test_batch_independence_and_raggedness and
test_full_size_batch256_properties)."""
import sys, torch
sys.path.insert(0, '.')
from seed_amd import config as C, lib as L
from seed_amd.tokenizer_engine import TokenizerEngine
from seed_amd.weights import make_tokenizer_state_dict, calibrate_codebook
lib = L.load()
cfg = C.MID
sd = make_tokenizer_state_dict(cfg, seed=4, ln_jitter=0.02)
img = torch.randn(40, 3, cfg.img_size, cfg.img_size, generator=torch.Generator().manual_seed(8)).cuda()
eng = TokenizerEngine(sd, cfg)
t = {}
eng.encode(img[:8], t)
eng.set_codebook(calibrate_codebook(t["z"].float().cpu(), cfg.n_embed, seed=7))
for xcd in (1, 0):
    L.check(lib.seedmi_set_option(b"gemm64_xcd", xcd), "opt")
    for fold in (1, 0):
        L.check(lib.seedmi_set_option(b"tokenize_lnfold", fold), "opt")
        want = eng.encode(img)
        ta, tb = {}, {}
        eng.encode(img[:10], ta); eng.encode(img, tb)
        got = torch.cat([eng.encode(img[i:i + 10]) for i in range(0, 40, 10)])
        one = torch.cat([eng.encode(img[i:i + 1]) for i in range(0, 8)])
        dz = (ta["z"].float() - tb["z"].float()[:10 * 32 if tb["z"].dim() == 2 else 10]).abs().max().item()
        print(f"xcd={xcd} lnfold={fold}: 40 vs 4x10 equal {torch.equal(want, got)} ({(want != got).sum().item()} ids differ); 40 vs singles {torch.equal(want[:8], one)}; max |dz| {dz:.3e}", flush=True)
