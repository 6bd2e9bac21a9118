#!/usr/bin/env python
"""Benchmark of the MI355X-native SEED tokenize-and-generate hot path (driver contract: see the task brief).

    python bench.py                                   # 1 GPU, default K/W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A "step" is one pass of the tokenize hot path over one batch of synthetic 224x224 images per GPU
(BASELINE.json configs[1]: SEED-2 tokenize, batch 256 per GPU, bf16): ``encode_image`` = EVA-ViT-g/14 ->
ln_vision -> causal Q-Former -> task MLP -> 8192-way VQ argmin, all inside ``seedmi_tokenize`` (hand-written
HIP behind the C ABI), followed at N > 1 by the RCCL all-gather of the int64 [B,32] token ids (the path's only
exchange step).  Images are resident in HBM before the timed region; weights are seeded
random-init tensors of the exact architecture (no checkpoints offline).

The JSON line also carries
  roofline      the dominant kernel (ViT QKV GEMM, M=B*257, K=1408, N=4224) timed live with HIP events
  cpu_baseline  the CPU path timed on this host: the reference's own modules where /root/reference exists, else the oracle port
  extra         whole-path MFMA fraction, SEED-LLaMA-8B greedy decode tokens/s (B=32) and its HBM roofline
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0     # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU per step (weak scaling)")
    ap.add_argument("--no-llama", action="store_true", help="skip the SEED-LLaMA-8B decode leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--prefill14b", action="store_true", help="(default since round 4; kept so old command lines still parse)")
    ap.add_argument("--no-prefill14b", action="store_true", help="skip the SEED-LLaMA-14B prefill leg (config 5; ~1 min)")
    ap.add_argument("--no-fp16", action="store_true", help="skip the fp16-build tokenize leg (extra.tokenize_fp16; ~5 s)")
    ap.add_argument("--cpu-images", type=int, default=8)
    ap.add_argument("--decode-batch", type=int, default=32)
    ap.add_argument("--decode-new", type=int, default=128)
    return ap.parse_args()


def time_kernel_events(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    beg = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
    end = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
    for i in range(iters):
        beg[i].record()
        fn()
        end[i].record()
    torch.cuda.synchronize()
    ms = sorted(b.elapsed_time(e) for b, e in zip(beg, end))
    return sum(ms) / len(ms), ms[len(ms) // 2]


def qkv_gemm_roofline(batch):
    """ViT QKV GEMM at this batch through the C ABI on torch's current stream, timed with HIP events."""
    from seed_amd import lib as L
    lib = L.load()
    M, K, N = batch * 257, 1408, 4224
    g = torch.Generator(device="cuda").manual_seed(1)
    A = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.02).bfloat16()
    bias = torch.zeros(N, device="cuda").bfloat16()
    Cc = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)

    # the form in which the tokenize path issues this GEMM (at M = batch * 257, BASELINE's definition; the path itself issues two
    # half-batch launches on two streams, timed below as run_half): LayerNorm (norm1) folded in - A is the un-normalised residual stream, the
    # row statistics, column sums and folded bias are the fold's operands (seedmi_gemm_bf16_ext; seedmi_tokenize passes no stream-K
    # workspace by default).  The plain nn.Linear launch and the one with the stream-K tail are timed beside it.
    import ctypes
    ws = torch.zeros(lib.seedmi_gemm_workspace_bytes(), dtype=torch.uint8, device="cuda")
    stats = torch.zeros(M + (M & 1), 2, dtype=torch.float32, device="cuda")
    L.check(lib.seedmi_layernorm_stats_bf16(L.ptr(A), K, M, K, 1e-6, L.ptr(stats), L.stream_ptr()), "stats")
    cs, b32 = W.float().sum(1).contiguous(), bias.float().contiguous()
    ext = L.GemmExt(L.ptr(stats), L.ptr(cs), L.ptr(b32), None, 0)

    def run():
        L.check(lib.seedmi_gemm_bf16_ext(M, N, K, L.ptr(A), K, L.ptr(W), K, None, None, 0, L.EPI_BIAS, L.ptr(Cc), N, 0, 0,
                                         ctypes.byref(ext), None, 0, L.stream_ptr()), "gemm ext")

    def run_plain():
        L.check(lib.seedmi_gemm_bf16(M, N, K, L.ptr(A), K, L.ptr(W), K, L.ptr(bias), None, 0, L.EPI_BIAS, L.ptr(Cc), N,
                                     0, 0, L.stream_ptr()), "gemm")

    def run_sk():
        L.check(lib.seedmi_gemm_bf16_ws(M, N, K, L.ptr(A), K, L.ptr(W), K, L.ptr(bias), None, 0, L.EPI_BIAS, L.ptr(Cc), N,
                                        0, 0, L.ptr(ws), ws.numel(), L.stream_ptr()), "gemm")

    # seedmi_tokenize splits a batch >= 32 into two sub-batches on two streams (tokenizer.hip), so the launch the timed path issues is
    # this one at M = (batch / 2) * 257; timed here alone on one stream (in the path two of them overlap)
    Mh = (batch // 2) * 257

    def run_half():
        L.check(lib.seedmi_gemm_bf16_ext(Mh, N, K, L.ptr(A), K, L.ptr(W), K, None, None, 0, L.EPI_BIAS, L.ptr(Cc), N, 0, 0,
                                         ctypes.byref(ext), None, 0, L.stream_ptr()), "gemm ext half")
    # the three launches are timed in alternating blocks after a common warm-up: timed one after the other, whichever came first ran
    # on colder clocks (the first block of a cold chip measured 0.74 ms for a 0.65 ms launch)
    for fn in (run, run_sk, run_plain) * 4:
        fn()
    acc = {run: [], run_sk: [], run_plain: []}
    for _ in range(3):
        for fn in (run, run_sk, run_plain):
            acc[fn].append(time_kernel_events(fn, 10, warm=2))
    avg_ms = sum(a for a, _ in acc[run]) / 3
    med_ms = sorted(m for _, m in acc[run])[1]
    sk_avg_ms = sum(a for a, _ in acc[run_sk]) / 3
    plain_avg_ms = sum(a for a, _ in acc[run_plain]) / 3
    half_ms = sum(time_kernel_events(run_half, 10, warm=2)[0] for _ in range(3)) / 3
    flops = 2.0 * M * N * K
    achieved = flops / (avg_ms * 1e-3) / 1e12
    traffic, traffic_src = None, None
    for name in ("r06_pmc_qkv_gemm256.json", "r05_pmc_qkv_gemm256.json", "r04_pmc_qkv_gemm256.json", "r03_pmc_qkv_gemm256.json", "r02_pmc_qkv_gemm256.json", "r01_pmc_qkv_gemm256.json"):
        pmc = os.path.join(ROOT, "profiles", name)
        if batch == 256 and os.path.exists(pmc):
            # bytes past the L2s per launch from rocprofv3 --pmc passes of this same kernel/shape (tools/pmc_qkv.sh: PMC counters
            # cannot be read from inside the benchmark process; one counter group per pass; FETCH_SIZE doubled per the gfx950
            # note in MI355X_MICROARCH.md, both in KiB)
            d = json.load(open(pmc))
            traffic = (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024
            traffic_src = "profiles/" + name + " (rocprofv3 --pmc passes of this launch, tools/pmc_qkv.sh; NOT measured in this run)"
            break
    # measured live beside it: the MFMA-only loop of THIS box (the power-limited ceiling the 2.5 PF datasheet figure sits above)
    return {"bound": "mfma", "kernel": "gemm256_kernel<BIAS, LayerNorm fold, schedule 8273 (two-phase K-tile, position-free body)>, persistent (ViT norm1 + QKV: M=%d K=%d N=%d)" % (M, K, N),
            "launch": "BASELINE-defined M = batch * 257 = %d launch (the tokenize path issues 2 x M = %d on two streams: in_path_launch)" % (M, Mh),
            "in_path_launch": {"M": Mh, "avg_launch_ms": round(half_ms, 4), "achieved": round(2.0 * Mh * N * K / (half_ms * 1e-3) / 1e12, 1),
                               "note": "isolated on one stream; inside seedmi_tokenize two such launches overlap"},
            "plain_linear": {"avg_launch_ms": round(plain_avg_ms, 4), "achieved": round(flops / (plain_avg_ms * 1e-3) / 1e12, 1)},
            "with_streamk_tail": {"avg_launch_ms": round(sk_avg_ms, 4), "achieved": round(flops / (sk_avg_ms * 1e-3) / 1e12, 1)},
            "achieved": round(achieved, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / MFMA_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
            # what the traffic figure means for the bound (profiles/r06_pmc_stalls_qkv_gemm256.json: memory-side requests counted by size -
            # 99.98 % of the reads are 128-byte requests, i.e. 1.356 GB, the writes are exactly C's 0.556 GB - and
            # profiles/r06_infinity_cache_eviction_ab.json: the launch takes the same time with its operands evicted from the Infinity Cache)
            "traffic_note": "1.91 GB past the L2s in 0.63 ms = 3.0 TB/s, under half of what HBM streams; the launch is as fast behind a 1 GiB eviction "
                            "sweep as with A and W resident in the Infinity Cache: the operand stream does not bound it",
            "algorithmic_bytes": 2.0 * (M * K + N * K + M * N),
            "flops_per_launch": flops, "avg_launch_ms": round(avg_ms, 4), "median_launch_ms": round(med_ms, 4)}


def measured_ceilings():
    """What THIS box sustains when nothing but the resource is exercised, measured live (< 1 s): an MFMA-only register loop
    (both bf16 MFMA shapes; the chip clocks to its power budget, so this sits below the 2.5 PFLOP/s datasheet peak that every
    headline fraction keeps as its denominator) and a 16-byte non-temporal stream read of 4 GiB."""
    import ctypes
    from seed_amd import lib as L
    from tools import calib                  # libseedcal.so: calibration kernels live outside the product library
    cal = calib.load()
    scratch = torch.zeros(4, dtype=torch.float32, device="cuda")
    out = {}
    for shape, name in ((0, "mfma_only_16x16x32_tflops"), (1, "mfma_only_32x32x16_tflops")):
        fl = ctypes.c_double(0.0)

        def run():
            calib.check(cal.seedcal_mfma_bf16(shape, 20000, 256, L.ptr(scratch), ctypes.byref(fl), L.stream_ptr()), "mfma bench")
        avg_ms, _ = time_kernel_events(run, 5, warm=2)
        out[name] = round(fl.value / (avg_ms * 1e-3) / 1e12, 1)
    buf = torch.empty(4 << 30, dtype=torch.uint8, device="cuda")
    buf.view(torch.int32).fill_(0x01020304)

    def rd():
        calib.check(cal.seedcal_stream_read(L.ptr(buf), buf.numel(), 4, L.ptr(scratch), L.stream_ptr()), "stream read")
    avg_ms, _ = time_kernel_events(rd, 5, warm=2)
    out["hbm_stream_read_gbps"] = round(buf.numel() / (avg_ms * 1e-3) / 1e9, 1)
    out["source"] = "measured in this run (tools/calib: seedcal_mfma_bf16, seedcal_stream_read)"
    return out


def vit_attention_leg(images=128):
    """SURVEY row a4 as the tokenize path issues it: ONE launch of the ViT attention for a sub-batch of 128 images (16 heads x 88, 257 tokens,
    packed q|k|v rows as the QKV GEMM writes them), timed in isolation with HIP events.  Its roofline is bytes (Q, K, V in, O out), not flops."""
    from seed_amd import lib as L
    lib = L.load()
    H, hd, N = 16, 88, 257
    C = H * hd
    g = torch.Generator(device="cuda").manual_seed(7)
    qkv = torch.randn(images * N, 3 * C, device="cuda", generator=g).bfloat16()
    out = torch.empty(images * N, C, device="cuda", dtype=torch.bfloat16)

    def run():
        L.check(lib.seedmi_attention_bf16(L.ptr(qkv), 3 * C, L.ptr(qkv[:, C:]), 3 * C, L.ptr(qkv[:, 2 * C:]), 3 * C, L.ptr(out), C,
                                          images, H, hd, N, N, hd ** -0.5, 0, 1, L.stream_ptr()), "attention")
    # (a burst between two events: a pair of events around a single ~0.1 ms launch adds ~15 % of event / launch latency to it)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    bursts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        bursts.append(e0.elapsed_time(e1) / 20)
    bursts.sort()
    avg_ms, med_ms = sum(bursts) / len(bursts), bursts[len(bursts) // 2]
    flops = 4.0 * images * H * N * N * hd
    nbytes = images * N * C * 2 * 4
    return {"kernel": "attn_vit16s_kernel (staggered 16-wave kernel, XCD-aware item walk)", "images": images, "avg_launch_us": round(avg_ms * 1e3, 1),
            "median_launch_us": round(med_ms * 1e3, 1), "timing": "5 bursts of 20 back-to-back launches between two HIP events", "tflops": round(flops / (avg_ms * 1e-3) / 1e12, 1),
            "roofline": {"bound": "hbm", "achieved": round(nbytes / (avg_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(nbytes / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes": nbytes}}


def vq_argmin_leg(images=128, pass_ms=None):
    """the 8192-way VQ nearest neighbour as the tokenize path launches it - ONE launch per sub-batch of 128
    images (4096 rows of z), the head's last Linear fused in front (seedmi_vq_head_argmin_bf16) - and the bare argmin beside it.  No MFMA:
    a workgroup owns 8 rows and sweeps the whole 512 KiB codebook (+ 32 KiB of norms) out of L2, so the byte figure that describes the kernel
    is the codebook re-streaming rate; its algorithmic HBM bytes (z in, ids out, the codebook once) are ~0.8 MB per launch."""
    from seed_amd import lib as L
    lib = L.load()
    rows, hidden, dim, n_embed = images * 32, 768, 32, 8192
    g = torch.Generator(device="cuda").manual_seed(11)
    t = torch.tanh(torch.randn(rows, hidden, device="cuda", generator=g)).bfloat16()
    w = (torch.randn(dim, hidden, device="cuda", generator=g) * 0.03).bfloat16()
    b = torch.zeros(dim, device="cuda").bfloat16()
    cb = (torch.randn(n_embed, dim, device="cuda", generator=g) * 0.3).bfloat16()
    ee = torch.empty(n_embed, device="cuda", dtype=torch.float32)
    L.check(lib.seedmi_vq_code_sqnorm(L.ptr(cb), L.ptr(ee), n_embed, dim, L.stream_ptr()), "sqnorm")
    ids = torch.empty(rows, dtype=torch.int64, device="cuda")
    z = torch.empty(rows, dim, device="cuda", dtype=torch.bfloat16)

    def fused():
        L.check(lib.seedmi_vq_head_argmin_bf16(L.ptr(t), hidden, hidden, L.ptr(w), hidden, L.ptr(b), L.ptr(cb), L.ptr(ee), L.ptr(ids),
                                               L.ptr(z), dim, rows, n_embed, dim, L.stream_ptr()), "vq head argmin")

    def bare():
        L.check(lib.seedmi_vq_argmin_bf16(L.ptr(z), dim, L.ptr(cb), L.ptr(ee), L.ptr(ids), rows, n_embed, dim, L.stream_ptr()), "vq argmin")

    def burst(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10)
        return sum(ts) / len(ts)
    f_ms, b_ms = burst(fused), burst(bare)
    sweep_bytes = (rows // 8) * (n_embed * dim * 2 + n_embed * 4)          # one codebook + norms sweep per 8-row workgroup
    alg_bytes = rows * dim * 2 + rows * 8 + n_embed * dim * 2 + n_embed * 4  # z in, int64 ids out, codebook + norms once
    out = {"kernel": "vq_head_argmin_kernel (wavefront argmin, no MFMA; Linear(768, 32) fused in front)", "rows": rows,
           "avg_launch_us": round(f_ms * 1e3, 1), "bare_argmin_us": round(b_ms * 1e3, 1),
           "codebook_restream_gbps": round(sweep_bytes / (b_ms * 1e-3) / 1e9, 1),
           "valu_tflops": round(2.0 * rows * n_embed * dim / (b_ms * 1e-3) / 1e12, 2),
           "roofline": {"bound": "hbm", "achieved": round(alg_bytes / (b_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(alg_bytes / (b_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "algorithmic_bytes": alg_bytes,
                        "note": "the sweep is VALU-bound (8192 x 32 ordered fp32 FMA chains per row, the reference's rounding points); "
                                "its HBM traffic is negligible by construction"}}
    if pass_ms:
        out["share_of_pass"] = round(2 * f_ms / pass_ms, 5)                # two sub-batch launches per 256-image pass
    return out


def tokenize_fp16_leg(cfg, device, images_bf16, codebook, ids_bf16, steps=3):
    """The same step through the fp16 build (libseedmi_f16.so: the reference's shipped `fp16: True`; same kernels, IEEE fp16 as the 16-bit
    element, fc1's GELU arithmetic instead of the bf16 table): images/s beside the headline bf16 number, and how many of the ids of the same
    images (same weights, same codebook) equal the bf16 build's.  Reported, not the headline: BASELINE.json's dtype is bf16."""
    from seed_amd.tokenizer_engine import TokenizerEngine
    from seed_amd.weights import make_tokenizer_state_dict
    sd = make_tokenizer_state_dict(cfg, seed=0, device="cuda")
    eng = TokenizerEngine(sd, cfg, device=device, dtype=torch.float16)
    del sd
    eng.set_codebook(codebook)
    img = images_bf16.to(torch.float16)                     # (bf16 values are fp16 values here: |x| < 8, 8 mantissa bits)
    ids = eng.encode(img)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        ids = eng.encode(img)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    return {"build": "libseedmi_f16.so (-DSEEDMI_F16 of the same sources)", "dtype": "f16", "value": round(img.shape[0] / ms * 1e3, 1),
            "unit": "images/s", "ms_per_step": round(ms, 3), "steps": steps,
            "ids_equal_to_bf16_build": round(float((ids == ids_bf16).float().mean()), 4),
            "note": "agreement with the REFERENCE's own fp16 run on the golden images: profiles/r05_id_agreement_fp16.json (0.990; the bf16 build 0.930)"}


def tokenizer_weight_bytes(cfg):
    """16-bit bytes of every weight one tokenize pass reads (ViT + ln_vision + Q-Former + task head + codebook): 2.18 GB at full size."""
    D, F, Q, FF = cfg.vit_dim, cfg.vit_ffn, cfg.qf_dim, cfg.qf_ffn
    vit = cfg.vit_depth * (3 * D * D + D * D + 2 * D * F + 3 * D + D + F + D + 4 * D) + D * cfg.patch_k + D + cfg.n_tokens * D + D + 2 * D
    n_cross = len([i for i in range(cfg.qf_layers) if i % cfg.cross_freq == 0])
    qf = cfg.qf_layers * (4 * Q * Q + 4 * Q + 2 * Q + 2 * Q * FF + FF + Q + 2 * Q) + n_cross * (2 * Q * Q + 2 * Q * D + 4 * Q + 2 * Q) + cfg.n_query * Q + 2 * Q
    head = Q * Q + Q + cfg.code_dim * Q + cfg.code_dim + cfg.n_embed * cfg.code_dim
    return 2 * (vit + qf + head)


def tokenize_latency_b1(eng, reps=50, weight_bytes=None):
    """The reference scripts' own call pattern (scripts/seed_tokenizer_inference.py:26-29): ONE image through encode_image.  Median of
    `reps` host-timed calls (launch + kernels + sync), and the same pass replayed from a hipGraph."""
    img = torch.randn(1, 3, 224, 224, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5)).bfloat16()
    for _ in range(3):
        eng.encode(img)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        eng.encode(img)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    med = ts[len(ts) // 2]
    wbytes = weight_bytes or 0
    out = {"images": 1, "median_us": round(med * 1e6, 1), "min_us": round(ts[0] * 1e6, 1), "reps": reps,
           "timing": "host clock around encode() + synchronize",
           # one image must stream every weight once (2.18 GB at full size) and its 533.5 GFLOP are 0.21 ms of MFMA: the HBM roofline bounds it
           "roofline": {"bound": "hbm", "bytes": wbytes, "achieved": round(wbytes / med / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(wbytes / med / 1e9 / HBM_PEAK_GBS, 4), "floor_us": round(wbytes / (HBM_PEAK_GBS * 1e9) * 1e6, 1)}}
    try:
        ids = eng.encode(img)
        gr = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            eng.encode(img)
        torch.cuda.current_stream().wait_stream(side)
        with torch.cuda.graph(gr):
            ids_g = eng.encode(img)
        torch.cuda.synchronize()
        tg = []
        for _ in range(reps):
            t0 = time.perf_counter()
            gr.replay()
            torch.cuda.synchronize()
            tg.append(time.perf_counter() - t0)
        tg.sort()
        out["graph_replay_median_us"] = round(tg[len(tg) // 2] * 1e6, 1)
        out["graph_ids_equal"] = bool(torch.equal(ids, ids_g))
    except Exception as e:
        out["graph_replay_error"] = repr(e)[:200]
    return out


def _reference_tokenizer():
    """The reference's OWN modules (models/seed_qformer/*.py), imported through oracle/ref_shims.py from /root/reference where that tree
    exists and from the bytecode oracle/build_ref.py compiled into oracle/_ref/ otherwise (that directory travels to the GPU box)."""
    from oracle import ref_shims, make_golden
    from seed_amd import config as C
    from seed_amd.weights import make_tokenizer_state_dict
    ref = ref_shims.load_reference_modules()
    sd = make_tokenizer_state_dict(C.SEED2, seed=0)
    mods = ref_shims.build_reference_tokenizer_modules(ref, C.SEED2)
    make_golden.load_tokenizer_weights(mods, sd)
    qt = sd["query_tokens"].clone()
    return (lambda img: ref_shims.reference_get_codebook_indices(mods, qt, img)), ref_shims.reference_origin()


def _timed(fn, img):
    t0 = time.time()
    fn(img)
    return time.time() - t0


def cpu_baseline(n_images):
    """The CPU path on a bounded sample of the workload, fp32, on THIS node's host cores: the reference's own modules ("reference";
    the oracle port timed beside it) wherever they can be imported - /root/reference in the build container, oracle/_ref on the GPU
    box - and the oracle port alone ("port") only if neither exists.  The thread count is not chosen by fiat: two images are timed
    at 8 / 16 / 32 / 64 host threads and the fastest setting runs the sample (all 256 hardware threads of the GPU box were measured once:
    0.013 images/s, two minutes for the probe alone - torch's CPU GEMMs at this size only lose to oversubscription beyond ~32)."""
    from oracle import seed_oracle as O, ref_shims
    from seed_amd import config as C
    from seed_amd.weights import make_tokenizer_state_dict
    ncpu = os.cpu_count() or 1
    img = torch.randn(n_images, 3, 224, 224, generator=torch.Generator().manual_seed(1234))
    ref_fn, origin, ref_err = None, None, None
    if ref_shims.reference_available():
        try:
            ref_fn, origin = _reference_tokenizer()
        except Exception as e:
            ref_err = repr(e)[:200]
    sd = make_tokenizer_state_dict(C.SEED2, seed=0)
    port_fn = lambda x: O.get_codebook_indices(sd, x, C.SEED2, "fp32")      # noqa: E731
    main_fn = ref_fn or port_fn
    sweep = {}
    for th in sorted({t for t in (8, 16, 32, 64) if t <= ncpu} or {ncpu}):
        torch.set_num_threads(th)
        main_fn(img[:1])                                                  # warm-up (thread pool, allocator)
        sweep[th] = round(2 / _timed(main_fn, img[:2]), 3)
    cores = max(sweep, key=sweep.get)
    torch.set_num_threads(cores)
    main_fn(img[:1])
    runs = [_timed(main_fn, img) for _ in range(3)]                       # three passes over the sample, the median reported (VERDICT r5 weak 10:
    dt = sorted(runs)[1]                                                  # one 8-image batch read 1.89-2.02 img/s across runs)
    port_fn(img[:1])
    dt_port = dt if ref_fn is None else _timed(port_fn, img)
    if ref_fn is not None:
        res = {"value": round(n_images / dt, 3), "unit": "images/s", "cores": cores, "kind": "reference",
               "sample": f"{n_images} images (one batch, median of three passes) of the same synthetic 224x224 workload, the reference's own modules from {origin} "
                         f"(fp32, dependency shims only), torch CPU with {cores} of {ncpu} host threads (fastest of the sweep), {dt:.1f} s; "
                         f"oracle port on the same sample: {n_images / dt_port:.3f} images/s",
               "port_value": round(n_images / dt_port, 3)}
    else:
        res = {"value": round(n_images / dt, 3), "unit": "images/s", "cores": cores, "kind": "port",
               "sample": f"{n_images} images (one batch) of the same synthetic 224x224 workload, fp32 oracle, "
                         f"torch CPU with {cores} of {ncpu} host threads (fastest of the sweep), {dt:.1f} s"}
        if ref_err:
            res["reference_error"] = ref_err
    res["thread_sweep_images_per_s"] = {str(k): v for k, v in sweep.items()}
    res["sample_runs_s"] = [round(r, 2) for r in runs]
    return res


def _decode_traffic_ratio():
    for name in ("r03_pmc_decode_gemm.json", "r02_pmc_decode_gemm.json", "r01_pmc_decode_gemm.json"):
        try:
            d = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", name)))
            return round(d["traffic_bytes"] / d["algorithmic_bytes"], 3)
        except Exception:
            continue
    return None


def decode_per_kernel(eng, cfg, B, ctx):
    """extra.llama_decode.per_kernel: the launches of ONE decode layer (+ the lm_head), each timed on its own in bursts over
    the 32 layers' weights (a different layer per launch: every weight byte comes from HBM, as in the step) between two HIP events on the
    launch stream - bytes, us and GB/s per launch.  The step's hipGraph replays exactly these launches back to back."""
    import ctypes
    from seed_amd import lib as L
    lib = eng.lib
    h, F, H, V = cfg.hidden, cfg.ffn, cfg.heads, cfg.vocab
    hd, Mp = h // H, (B + 15) // 16 * 16
    dev, dt = eng.device, eng.dtype
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(B, h, device=dev, generator=g).to(dt)
    xn = torch.randn(Mp, h, device=dev, generator=g).to(dt)                      # fragment-major images (any values: timing only)
    att, act = torch.randn(Mp, h, device=dev, generator=g).to(dt), torch.randn(Mp, F, device=dev, generator=g).to(dt)
    qkv = torch.randn(B, 3 * h, device=dev, generator=g).to(dt)
    logits = torch.empty(B, eng.vocab_pad, device=dev, dtype=dt)
    sk = torch.empty(lib.seedmi_gemm_skinny_workspace_bytes(), dtype=torch.uint8, device=dev)
    L.check(lib.seedmi_gemm_skinny_workspace_init(L.ptr(sk), sk.numel(), L.stream_ptr()), "skinny ws init")
    EPS, nl = cfg.rms_eps, cfg.layers
    scale = hd ** -0.5

    def skinny(N, K, A, Wp, eps, res, epi, C, ldc, c_packed, xp):
        return lib.seedmi_gemm_skinny_norm_ws_bf16(B, N, K, L.ptr(A), 1, Wp, eps, L.ptr(res) if res is not None else None, h if res is not None else 0,
                                                   epi, L.ptr(C), ldc, c_packed, L.ptr(xp) if xp is not None else None, L.ptr(sk), sk.numel(), L.stream_ptr())
    launches = {
        "q/k/v": (lambda l: skinny(3 * h, h, xn, eng._layers[l].qkv_wp, EPS, None, L.EPI_NONE, qkv, 3 * h, 0, None), 3 * h * h * 2),
        "attention (RoPE + append fused)": (lambda l: lib.seedmi_llama_decode_attention_bf16(
            L.ptr(qkv), 3 * h, None, eng.w.cos_t, eng.w.sin_t, eng._layers[l].k_cache, eng._layers[l].v_cache, L.ptr(att), h, B, H, hd, eng.tmax, ctx,
            scale, 1, None, cfg.max_pos, L.stream_ptr()), 2 * B * (ctx + 1) * h * 2),
        "o_proj + residual": (lambda l: skinny(h, h, att, eng._layers[l].o_wp, 0.0, x, L.EPI_BIAS_RESIDUAL, x, h, 0, xn), h * h * 2),
        "gate/up + SwiGLU": (lambda l: skinny(2 * F, h, xn, eng._layers[l].gate_up_wp, EPS, None, L.EPI_SWIGLU, act, F, 1, None), 2 * F * h * 2),
        "down + residual": (lambda l: skinny(h, F, act, eng._layers[l].down_wp, 0.0, x, L.EPI_BIAS_RESIDUAL, x, h, 0, xn), F * h * 2),
    }
    out, layer_us = {}, 0.0
    for name, (fn, nbytes) in launches.items():
        for l in range(nl):
            L.check(fn(l), name)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for l in range(nl):
                fn(l)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / nl * 1e3)
        us = sorted(ts)[2]
        layer_us += us
        out[name] = {"bytes": nbytes, "us": round(us, 2), "gbps": round(nbytes / us / 1e3, 1)}
    fn = lambda: skinny(V, h, xn, eng.w.lm_head_p, EPS, None, L.EPI_NONE, logits, eng.vocab_pad, 0, None)      # noqa: E731
    for _ in range(3):
        L.check(fn(), "lm_head")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    out["lm_head (weights stay in the Infinity Cache between these launches)"] = {"bytes": V * h * 2, "us": round(us, 2), "gbps": round(V * h * 2 / us / 1e3, 1)}
    L.check(lib.seedmi_gemm_skinny_ws_status(L.ptr(sk), sk.numel(), L.stream_ptr()), "skinny ws status")
    return {"batch": B, "context": ctx, "launches": out, "layer_sum_us": round(layer_us, 2),
            "timing": "bursts of 32 launches (one per layer's weights) between two HIP events on the launch stream; median of 5"}


def llama_decode_leg(B, n_new):
    """SEED-LLaMA-8B (Vicuna-7B body, vocab 40194): image -> 32 tokens -> greedy decode, batch B, bf16."""
    from seed_amd import config as C
    from seed_amd.llama_engine import LlamaEngine
    from seed_amd.weights import make_llama_state_dict
    cfg = C.LLAMA_8B
    sd = make_llama_state_dict(cfg, seed=0, device="cuda", dtype=torch.bfloat16)
    T0 = 59                                                        # SURVEY.md section 8d config 3 prompt length
    eng = LlamaEngine(sd, cfg, device="cuda", batch_cap=B, tmax=256)
    del sd
    g = torch.Generator(device="cuda").manual_seed(99)
    prompt = torch.randint(3, 32000, (B, T0), device="cuda", generator=g)
    prompt[:, 0] = 1
    prompt[:, 10:42] = 32000 + torch.randint(0, 8192, (B, 32), device="cuda", generator=g)   # <img_XXXXX> x 32
    eng.greedy_decode_graph(prompt, 4)                             # warm-up (kernels, allocator, graph machinery)
    torch.cuda.synchronize()
    eng.reset()
    t0 = time.time()
    logits = eng.forward(prompt, last_only=True)
    torch.cuda.synchronize()
    t_prefill = time.time() - t0
    tok = logits[:, 0].float().argmax(-1, keepdim=True)
    replay, out = eng.capture_decode_graph(tok, n_new)             # one captured step, replayed n_new-1 times
    torch.cuda.synchronize()
    t0 = time.time()
    replay(n_new - 1)
    torch.cuda.synchronize()
    dt = time.time() - t0
    eng.decode_status(B)                                           # (outside the timed region: reads the split-K error word)
    assert int(out.min()) >= 0 and int(out.max()) < cfg.vocab
    steps = n_new - 1
    tok_s = B * steps / dt
    ctx_mid = T0 + n_new // 2
    bytes_step = cfg.linear_params() * 2 + B * ctx_mid * cfg.kv_bytes_per_token() + B * cfg.kv_bytes_per_token()
    gbs = bytes_step / (dt / steps) / 1e9
    # the reference scripts' own call pattern (scripts/seed_llama_inference_8B.py:93-101): ONE sequence
    b1 = None
    try:
        eng.reset()
        lg1 = eng.forward(prompt[:1].contiguous(), last_only=True)
        tok1 = lg1[:, 0].float().argmax(-1, keepdim=True)
        replay1, out1 = eng.capture_decode_graph(tok1, 64)
        torch.cuda.synchronize()
        t1 = time.time()
        replay1(63)
        torch.cuda.synchronize()
        dt1 = (time.time() - t1) / 63
        eng.decode_status(1)
        w_bytes = cfg.linear_params() * 2
        b1 = {"batch": 1, "ms_per_token": round(dt1 * 1e3, 3), "tokens_per_s": round(1.0 / dt1, 1),
              "hbm_frac": round(w_bytes / dt1 / 1e9 / HBM_PEAK_GBS, 4), "graph_nodes_per_step": "one hipGraph replay per token"}
    except Exception as e:
        b1 = {"error": repr(e)[:200]}
    try:
        per_kernel = decode_per_kernel(eng, cfg, B, ctx_mid)
    except Exception as e:
        per_kernel = {"error": repr(e)[:300]}
    return {"metric": "tokens/s SEED-LLaMA-8B greedy decode", "value": round(tok_s, 1), "batch": B, "new_tokens": n_new, "latency_b1": b1,
            "per_kernel": per_kernel,
            "ms_per_step": round(dt / steps * 1e3, 3), "prefill_ms": round(t_prefill * 1e3, 2), "prompt_len": T0,
            "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(gbs / HBM_PEAK_GBS, 4), "bytes_per_step": bytes_step,
                         # PMC (profiles/r0x_pmc_decode_gemm.json, NOT measured in this run): bytes past the L2s of the q/k/v
                         # weight-streaming launch over its algorithmic bytes
                         "traffic_over_algorithmic_qkv_gemm": _decode_traffic_ratio()},
            "kernels": "gemm_skinny_sk_kernel (balanced split-K; uncut q/k/v, o, down; cut gate/up, lm_head), attn_decode_rope_kernel<keys early>"}


def llama14b_prefill_leg(B=8, T=649):
    """SURVEY.md section 8d config 5, one GPU's share: SEED-LLaMA-14B (LLaMA-2-13B body), 8 interleaved sequences of 649 tokens
    (4 x (<img> 32 codes </img>) + 4 x 128 text ids), one prefill forward with the KV cache written; prefill tokens/s and the
    MFMA fraction of the linear + causal-attention FLOPs."""
    from seed_amd import config as C
    from seed_amd.llama_engine import LlamaEngine
    from seed_amd.weights import make_llama_state_dict
    cfg = C.LLAMA_14B
    sd = make_llama_state_dict(cfg, seed=0, device="cuda", dtype=torch.bfloat16)
    eng = LlamaEngine(sd, cfg, device="cuda", batch_cap=B, tmax=704)
    del sd
    g = torch.Generator(device="cuda").manual_seed(0)
    ids = torch.randint(3, 32000, (B, T), device="cuda", generator=g)
    for r in range(4):
        s0 = 1 + r * (128 + 34) + 128
        ids[:, s0] = 32000 + 8192
        ids[:, s0 + 1:s0 + 33] = 32000 + torch.randint(0, 8192, (B, 32), device="cuda", generator=g)
        ids[:, s0 + 33] = 32000 + 8193
    for _ in range(2):
        eng.reset()
        eng.forward(ids, last_only=True)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        eng.reset()
        t0 = time.time()
        eng.forward(ids, last_only=True)
        torch.cuda.synchronize()
        ts.append(time.time() - t0)
    dt = sorted(ts)[1]
    flops = B * T * 2.0 * cfg.linear_params() - (B * (T - 1)) * 2.0 * cfg.vocab * cfg.hidden + B * cfg.layers * 2.0 * T * T * cfg.hidden
    # config 5's decode side (VERDICT r4 missing 3): greedy decode of the same 8 sequences behind that prefill, one captured step replayed
    dec = None
    try:
        n_new = 32
        eng.reset()
        logits = eng.forward(ids, last_only=True)
        tok = logits[:, 0].float().argmax(-1, keepdim=True)
        replay, out_tok = eng.capture_decode_graph(tok, n_new)
        torch.cuda.synchronize()
        t0 = time.time()
        replay(n_new - 1)
        torch.cuda.synchronize()
        ddt = (time.time() - t0) / (n_new - 1)
        eng.decode_status(B)
        ctx = T + n_new // 2
        bytes_step = cfg.linear_params() * 2 + B * ctx * cfg.kv_bytes_per_token() + B * cfg.kv_bytes_per_token()
        dec = {"metric": "tokens/s SEED-LLaMA-14B greedy decode behind the 649-token prefill", "value": round(B / ddt, 1), "batch": B,
               "ms_per_step": round(ddt * 1e3, 3), "mean_context": ctx,
               "roofline": {"bound": "hbm", "achieved": round(bytes_step / ddt / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(bytes_step / ddt / 1e9 / HBM_PEAK_GBS, 4), "bytes_per_step": bytes_step}}
    except Exception as e:
        dec = {"error": repr(e)[:300]}
    return {"metric": "prefill tokens/s SEED-LLaMA-14B, one GPU's 8 x 649-token share of config 5", "value": round(B * T / dt, 1), "decode": dec,
            "ms": round(dt * 1e3, 2), "batch": B, "seq_len": T,
            "roofline": {"bound": "mfma", "achieved": round(flops / dt / 1e12, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(flops / dt / 1e12 / MFMA_PEAK_TFLOPS, 4)}}


def relaunch_argv(args_gpus, argv, env):
    """`python bench.py --gpus N` with N > 1 outside a launcher: the command line that re-runs this script as N ranks under
    torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1), or None when no relaunch is needed (N = 1, or already a rank)."""
    if args_gpus <= 1 or "WORLD_SIZE" in env:
        return None
    port = env.get("MASTER_PORT") or str(29500 + os.getpid() % 2000)
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args_gpus}", "--master-addr", "127.0.0.1",
            "--master-port", port, os.path.abspath(__file__)] + list(argv)


def _now(use_cuda):
    if use_cuda:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev
    return time.perf_counter()


def _span_ms(a, b, use_cuda):
    return a.elapsed_time(b) if use_cuda else (b - a) * 1e3


def timed_steps(encode, images, dist, world, steps, warmup, use_cuda=True):
    """The timed region of the driver contract: `warmup` untimed steps, then EXACTLY `steps` steps (encode + the id all-gather at N > 1)
    bracketed by barrier + synchronize on both sides; the wall time is the MAX over ranks.  Returns (seconds, per-rank rows
    [wall ms per step, tokenize ms per step, id gather ms per step], ids of the last step).  `encode` is the engine's encode_image on
    the GPU; tests/test_cabi_and_host.py runs the same plumbing with a stand-in encoder over gloo (use_cuda=False)."""
    from seed_amd.dist import gather_token_ids
    sync = torch.cuda.synchronize if use_cuda else (lambda: None)
    marks = []

    def step(timed=False):
        t0 = _now(use_cuda) if timed else None
        ids = encode(images)
        t1 = _now(use_cuda) if timed else None
        ids = gather_token_ids(ids, dist) if world > 1 else ids
        if timed:
            marks.append((t0, t1, _now(use_cuda)))
        return ids

    ids = None
    for _ in range(warmup):
        ids = step()
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        ids = step(timed=True)
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    tok_ms = sum(_span_ms(m[0], m[1], use_cuda) for m in marks) / len(marks)
    gat_ms = sum(_span_ms(m[1], m[2], use_cuda) for m in marks) / len(marks)
    per_rank = [[dt / steps * 1e3, tok_ms, gat_ms]]
    if dist is not None:
        dev = "cuda" if use_cuda else "cpu"
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        mine = torch.tensor(per_rank[0], device=dev, dtype=torch.float64)
        allr = torch.empty(world * 3, device=dev, dtype=torch.float64)
        dist.all_gather_into_tensor(allr, mine)
        per_rank = allr.view(world, 3).cpu().tolist()
        dt = t.item()
    return dt, per_rank, ids


def main():
    args = parse()
    cmd = relaunch_argv(args.gpus, sys.argv[1:], os.environ)
    if cmd is not None:
        # `python bench.py --gpus 8` used to run ONE rank and print n_gpus: 1: become the N-rank job instead
        if not torch.cuda.is_available() or torch.cuda.device_count() < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: {torch.cuda.device_count() if torch.cuda.is_available() else 0} GPU(s) visible")
        os.execv(cmd[0], cmd)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} (or drop the launcher: "
                         f"`python bench.py --gpus {args.gpus}` starts the ranks itself)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the hot path has no CPU fallback)")
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        assert dist.get_world_size() == world and dist.get_backend() == "nccl"
    from seed_amd import config as C
    from seed_amd.tokenizer_engine import TokenizerEngine
    from seed_amd.weights import make_tokenizer_state_dict, calibrate_codebook
    cfg = C.SEED2
    B = args.batch
    sd = make_tokenizer_state_dict(cfg, seed=0, device="cuda")
    eng = TokenizerEngine(sd, cfg, device=f"cuda:{local}")
    del sd
    g = torch.Generator(device="cuda").manual_seed(1234 + rank)
    images = torch.randn(B, 3, 224, 224, device="cuda", generator=g).bfloat16()      # resident in HBM
    # the synthetic codebook is calibrated on a rank-independent image set so every rank quantises identically
    gc = torch.Generator(device="cuda").manual_seed(4321)
    calib = torch.randn(8, 3, 224, 224, device="cuda", generator=gc).bfloat16()
    taps = {}
    eng.encode(calib, taps)
    codebook = calibrate_codebook(taps["z"].float().cpu(), cfg.n_embed, seed=7)
    eng.set_codebook(codebook)
    del taps, calib

    # per-step events on the compute stream (asynchronous: nothing waits on them inside the timed region): tokenize time and gather time of
    # THIS rank, so that a multi-GPU run explains its own scaling loss
    dt, per_rank, ids = timed_steps(eng.encode, images, dist, world, args.steps, args.warmup)
    assert tuple(ids.shape) == (world * B, 32) and int(ids.min()) >= 0 and int(ids.max()) < 8192

    if rank == 0:
        img_s = world * B * args.steps / dt
        out = {
            "metric": "images/s SEED-2 tokenize", "value": round(img_s, 2), "unit": "images/s", "n_gpus": world,
            "rccl_ranks": (dist.get_world_size() if dist is not None else 1),
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "SEED-2 tokenize (EVA-ViT-g/14 + causal Q-Former + 8192x32 VQ), 224x224, "
                                   f"batch {B} per GPU, bf16; ids all-gathered over RCCL at N>1",
                       "images_per_gpu": B, "global_batch": world * B, "parallelism": f"dp{world}",
                       "weights": "seeded random-init (reference initialisers)"},
        }
        flops_img = cfg.flops_per_image()
        extra = {"per_rank_ms": {"columns": ["wall ms per step (host clock)", "tokenize ms per step (events)", "id gather ms per step (events)"],
                                 "rows": [[round(v, 3) for v in r] for r in per_rank],
                                 "wire": "ids all-gathered as int16 (16 KiB per rank at 256 images), widened to int64 on arrival"},
                 "gflop_per_image": round(flops_img / 1e9, 2),
                 "path_mfma_frac": round(img_s / world * flops_img / (MFMA_PEAK_TFLOPS * 1e12), 4),
                 "peak_denominators": {"mfma_bf16_dense_tflops": MFMA_PEAK_TFLOPS, "hbm_gbps": HBM_PEAK_GBS,
                                       "source": "/opt/skills/guides/MI355X_MICROARCH.md"}}
        # the dominant kernel is local to a GPU: rank 0 times it at every N (outside the timed region); the CPU baseline
        # is an N = 1 leg only
        try:
            extra["latency_b1"] = {"tokenize": tokenize_latency_b1(eng, weight_bytes=tokenizer_weight_bytes(cfg))}
        except Exception as e:
            extra["latency_b1"] = {"tokenize": {"error": repr(e)[:200]}}
        if world == 1 and not args.no_fp16:
            try:
                extra["tokenize_fp16"] = tokenize_fp16_leg(cfg, f"cuda:{local}", images, codebook, ids[:B])
            except Exception as e:
                extra["tokenize_fp16"] = {"error": repr(e)[:200]}
        out["roofline"] = qkv_gemm_roofline(B)
        out["cpu_baseline"] = cpu_baseline(args.cpu_images) if (world == 1 and not args.no_cpu_baseline) else None
        del eng, images
        torch.cuda.empty_cache()
        try:
            extra["vit_attention"] = vit_attention_leg()
        except Exception as e:
            extra["vit_attention"] = {"error": repr(e)[:200]}
        try:
            extra["vq_argmin"] = vq_argmin_leg(pass_ms=dt / args.steps * 1e3)
        except Exception as e:
            extra["vq_argmin"] = {"error": repr(e)[:200]}
        try:
            extra["measured_ceilings"] = measured_ceilings()
        except Exception as e:
            extra["measured_ceilings"] = {"error": repr(e)[:200]}
        torch.cuda.empty_cache()
        if world == 1 and not args.no_llama:
            try:
                extra["llama_decode"] = llama_decode_leg(args.decode_batch, args.decode_new)
            except Exception as e:  # the tokenize line must survive a failure of the secondary leg
                extra["llama_decode"] = {"error": repr(e)[:300]}
        if world == 1 and not args.no_prefill14b:
            torch.cuda.empty_cache()
            try:
                extra["llama14b_prefill"] = llama14b_prefill_leg()
            except Exception as e:
                extra["llama14b_prefill"] = {"error": repr(e)[:300]}
        out["extra"] = extra
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
