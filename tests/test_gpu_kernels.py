"""Kernel-level parity on a real MI355X: every C-ABI kernel against a plain fp32 restatement of the same op
(with the reference's rounding points) on the same seeded inputs.  Integer/index work (VQ ids) is bit-exact
against the oracle; floating point within the tolerance written in each test.
"""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from seed_amd import lib as L  # noqa: E402


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "GPU tests selected but no HIP device visible"
    l = L.load()                      # raises if libseedmi.so is missing: no silent fallback
    L.check(l.seedmi_check_device(), "seedmi_check_device")
    return l


def bf(x):
    return x.to(torch.bfloat16)


def r(x):
    return x.to(torch.bfloat16).float()


def gelu(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def rand(gen, *shape, scale=1.0):
    return (torch.randn(*shape, generator=gen) * scale)


def assert_close_bf16(got, want, what, atol_ulps=1.5, frac=0.999):
    """got: bf16 tensor from the kernel; want: fp32 tensor already rounded where the reference rounds.
    At least ``frac`` of elements must be within atol_ulps bf16 ulps of |want|, and the normalised error small."""
    got = got.float().cpu()
    want = want.float().cpu()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert torch.isfinite(got).all(), what
    err = (got - want).abs()
    ulp = torch.clamp(want.abs(), min=1e-30) * 2.0 ** -8
    floor = want.abs().mean() * 2.0 ** -8
    ok = err <= atol_ulps * torch.maximum(ulp, floor)
    rel = (got - want).norm() / want.norm().clamp(min=1e-30)
    print(f"[{what}] rel={rel:.3e} within={ok.float().mean():.5f} max_err={err.max():.3e}")
    assert rel < 4e-3, (what, rel.item())
    assert ok.float().mean() >= frac, (what, ok.float().mean().item())


def run_gemm(lib, A, W, bias, res, epi, out_rows=None, out_cols=None, row_group=0, row_extra=0, C_init=None):
    M, K = A.shape
    N = W.shape[0]
    out_rows = out_rows or M
    out_cols = out_cols or N
    C = torch.zeros(out_rows, out_cols, dtype=torch.bfloat16, device="cuda") if C_init is None else C_init
    rc = lib.seedmi_gemm_bf16(M, N, K, L.ptr(A), A.stride(0), L.ptr(W), W.stride(0), L.ptr(bias), L.ptr(res),
                              0 if res is None else res.stride(0), epi, L.ptr(C), C.stride(0), row_group, row_extra,
                              L.stream_ptr())
    L.check(rc, "seedmi_gemm_bf16")
    torch.cuda.synchronize()
    return C


@pytest.fixture(params=[64, 65, 66, 67, 68, 70, 128, 256, (256, 0), (256, 24657), (256, 57425)],
                ids=["gemm64", "gemm64_4wave", "gemms_64x64", "gemms_128x64", "gemms_64x128", "gemms_32x64", "gemm128", "gemm256", "gemm256_sched0",
                     "gemm256_seam", "gemm256_peel"])
def gemm_variant(request, lib):
    """Every GEMM parity test runs once per tile kernel (64x64 deep-ring small-M kernel, 128x128 two-barrier and 256x256 staggered deep pipeline), the 256x256 kernel
    under its default schedule (gemm_sched 8273: two-phase K-tile, position-free body, round 4), under the round-2 schedule (0) and
    under round 5's seam (24657) and seam + peeled store-tolerant K-tiles (57425) - the values the product library keeps selectable."""
    variant, sched = request.param if isinstance(request.param, tuple) else (request.param, -1)
    L.check(lib.seedmi_set_option(b"gemm", variant), "set_option")
    L.check(lib.seedmi_set_option(b"gemm_sched", sched), "set_option")
    yield variant
    lib.seedmi_set_option(b"gemm", 0)
    lib.seedmi_set_option(b"gemm_sched", -1)


GEMM_SHAPES = [
    (257 * 5, 1408, 1408),      # ViT proj shape, ragged M and (for the 256 tile) ragged N
    (300, 4224, 1408),          # QKV
    (128, 6144, 1408),          # fc1
    (200, 1408, 6144),          # fc2 (long K)
    (64, 32, 768),              # VQ head: N smaller than a tile
    (1000, 768, 768),
    (130, 1536, 1408),          # cross K|V
    (48, 40194 // 2 * 2, 256),  # ragged N (not a multiple of 16)
]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_gemm_bias(lib, gemm_variant, M, N, K):
    gen = torch.Generator().manual_seed(M * 7 + N)
    A = bf(rand(gen, M, K)).cuda()
    W = bf(rand(gen, N, K, scale=0.05)).cuda()
    # asymmetric structure so a transposed / permuted C write cannot pass
    W[: min(N, 64)] *= torch.arange(1, min(N, 64) + 1, device="cuda").unsqueeze(1).to(torch.bfloat16) / 16
    bias = bf(rand(gen, N)).cuda()
    ldc = (N + 7) // 8 * 8
    C = torch.zeros(M, ldc, dtype=torch.bfloat16, device="cuda")
    run_gemm(lib, A, W, bias, None, L.EPI_BIAS, C_init=C)
    want = r(A.float() @ W.float().t() + bias.float())
    assert_close_bf16(C[:, :N], want, f"gemm_bias {M}x{N}x{K}")
    if ldc > N:
        assert (C[:, N:] == 0).all(), "wrote past N"


@pytest.mark.parametrize("M", [257, 514, 40])
def test_small_m_kernels_are_bit_identical(lib, M):
    """One image is M = 257: the library then picks, per GEMM, the shape of the small-M kernel whose busiest CU streams the fewest bytes
    (64x64 / 128x64 / 64x128 tiles, eight producer waves: round 6).  Whatever it picks - and whatever a batch of 256 takes (the 128x128 and
    256x256 kernels) - every output element is the same k-ordered MFMA chain: all kernels bit-equal on the four ViT GEMM shapes with the
    epilogues the path uses (an image alone must give the ids it gives inside a batch)."""
    gen = torch.Generator().manual_seed(900 + M)
    D, F = 1408, 6144
    for name, N, K, epi in (("qkv", 3 * D, D, L.EPI_BIAS), ("proj", D, D, L.EPI_BIAS_RESIDUAL), ("fc1", F, D, L.EPI_BIAS_GELU), ("fc2", D, F, L.EPI_BIAS_RESIDUAL)):
        A = bf(rand(gen, M, K)).cuda()
        W = bf(rand(gen, N, K, scale=0.03)).cuda()
        bias = bf(rand(gen, N, scale=0.1)).cuda()
        R = bf(rand(gen, M, N)).cuda() if epi == L.EPI_BIAS_RESIDUAL else None
        outs = {}
        try:
            for v in (0, 64, 66, 67, 68, 70, 128, 256):
                L.check(lib.seedmi_set_option(b"gemm", v), "set_option")
                outs[v] = run_gemm(lib, A, W, bias, R, epi)
        finally:
            lib.seedmi_set_option(b"gemm", 0)
        for v, C in outs.items():
            assert torch.equal(C.view(torch.int16), outs[128].view(torch.int16)), f"{name} M={M}: kernel {v} differs from the 128x128 kernel in {(C != outs[128]).sum().item()} values"


def test_gemm_identity_layout(lib, gemm_variant):
    """A = I  =>  C == W^T exactly; catches any row/column permutation in fragment or epilogue mapping."""
    K = N = 256
    M = 256
    A = torch.eye(M, K, dtype=torch.bfloat16, device="cuda")
    gen = torch.Generator().manual_seed(3)
    W = bf(rand(gen, N, K)).cuda()
    C = run_gemm(lib, A, W, None, None, L.EPI_NONE)
    assert torch.equal(C, W.t().contiguous())


@pytest.mark.parametrize("epi", ["gelu", "tanh", "residual", "swiglu", "none"])
def test_gemm_epilogues(lib, gemm_variant, epi):
    gen = torch.Generator().manual_seed(11)
    M, N, K = 321, 768, 512
    A = bf(rand(gen, M, K)).cuda()
    W = bf(rand(gen, N, K, scale=0.06)).cuda()
    bias = bf(rand(gen, N, scale=0.5)).cuda()
    acc = A.float() @ W.float().t()
    if epi == "gelu":
        C = run_gemm(lib, A, W, bias, None, L.EPI_BIAS_GELU)
        want = r(gelu(r(acc + bias.float())))
    elif epi == "tanh":
        C = run_gemm(lib, A, W, bias, None, L.EPI_BIAS_TANH)
        want = r(torch.tanh(r(acc + bias.float())))
    elif epi == "residual":
        res = bf(rand(gen, M, N)).cuda()
        C = run_gemm(lib, A, W, bias, res, L.EPI_BIAS_RESIDUAL)
        want = r(r(acc + bias.float()) + res.float())
    elif epi == "swiglu":
        C = run_gemm(lib, A, W, None, None, L.EPI_SWIGLU, out_cols=N // 2)
        g, u = r(acc[:, 0::2]), r(acc[:, 1::2])
        want = r(r(torch.nn.functional.silu(g)) * u)
    else:
        C = run_gemm(lib, A, W, None, None, L.EPI_NONE)
        want = r(acc)
    assert_close_bf16(C, want, f"gemm_{epi}", frac=0.998)


def test_gemm_gelu_epilogue_is_torch_gelu_bit_for_bit(lib, gemm_variant):
    """The GELU epilogue is a bf16 -> bf16 table of torch's own nn.GELU() (tools/gen_gelu_lut.py): with an identity GEMM
    (A = I, bias = 0) the output must EQUAL F.gelu(W^T) for every bf16 bit pattern, including the inputs outside the table
    (|x| < 2^-16, |x| >= 16, zeros, infinities' neighbours) that take the polynomial path."""
    K = 256
    bits = torch.arange(0, 65536, dtype=torch.int32)
    vals = (bits << 16).view(torch.float32)
    keep = torch.isfinite(vals) & ((vals == 0) | (vals.abs() > 1e-37))          # (torch flushes bf16 subnormals in gelu)
    vals = vals[keep].to(torch.bfloat16)
    N = ((vals.numel() + K - 1) // K + 7) // 8 * 8                             # (ldc must be a multiple of 8)
    pad = torch.zeros(N * K - vals.numel(), dtype=torch.bfloat16)
    W = torch.cat([vals, pad]).view(N, K).cuda()
    A = torch.eye(K, K, dtype=torch.bfloat16, device="cuda")
    C = run_gemm(lib, A, W, torch.zeros(N, dtype=torch.bfloat16, device="cuda"), None, L.EPI_BIAS_GELU)     # C = gelu(W^T)
    want = torch.nn.functional.gelu(W.cpu()).t().contiguous()
    got = C.cpu()
    same = (got.view(torch.int16) == want.view(torch.int16)) | ((got == 0) & (want == 0))
    in_table = (W.cpu().float().abs() >= 2.0 ** -16) & (W.cpu().float().abs() < 16)
    print(f"[gelu table] {same.float().mean().item():.6f} equal; in-table mismatches {(~same.t() & in_table).sum().item()}")
    assert not (~same.t() & in_table).any(), "table range must be bit-identical to torch"
    # outside the table the polynomial form runs: compare where squares stay finite in fp32 (|x| < 1e15); the huge values are x or -0
    sane = (W.cpu().float().abs() < 1e15).t()
    assert_close_bf16(torch.where(sane, got.float(), torch.zeros(())), torch.where(sane, want.float(), torch.zeros(())),
                      "gelu outside the table", frac=0.9999)
    big = ~sane & ((W.cpu().float() > 0) & (W.cpu().float() < 1e38)).t()     # (above 1.7e38 torch's x * (1 + erf) overflows to inf)
    assert torch.equal(got[big], want[big])


@pytest.mark.parametrize("M,N,K,epi", [(257 * 256, 1408, 1408, "residual"), (257 * 256 + 77, 1408, 768, "bias"),
                                        (257 * 128, 4224, 1408, "bias"), (257 * 64, 6144, 1408, "gelu"),
                                        (256 * 300, 768, 64, "bias")])
def test_gemm_streamk_is_bit_identical_to_data_parallel(lib, M, N, K, epi, variant=256):
    """seedmi_gemm_bf16_ws: the last partial round of tiles is cut along K (stream-K); a shared tile's K tail CONTINUES the fp32
    accumulator image its partner published, so every output equals the data-parallel launch bit for bit - and both match the
    fp32 restatement.  Repeated launches reuse the workspace (epoch flags), as the tokenizer does."""
    gen = torch.Generator().manual_seed(M + N + K)
    A = bf(rand(gen, M, K)).cuda()
    W = bf(rand(gen, N, K, scale=0.04)).cuda()
    bias = bf(rand(gen, N, scale=0.3)).cuda()
    res = bf(rand(gen, M, N)).cuda() if epi == "residual" else None
    code = {"residual": L.EPI_BIAS_RESIDUAL, "bias": L.EPI_BIAS, "gelu": L.EPI_BIAS_GELU}[epi]
    ws = torch.empty(lib.seedmi_gemm_workspace_bytes(), dtype=torch.uint8, device="cuda")
    ws[:4096].zero_()
    L.check(lib.seedmi_set_option(b"gemm", variant), "set_option")
    C0 = run_gemm(lib, A, W, bias, res, code)
    for rep in range(3):
        C1 = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
        rc = lib.seedmi_gemm_bf16_ws(M, N, K, L.ptr(A), K, L.ptr(W), K, L.ptr(bias), L.ptr(res), 0 if res is None else N, code,
                                     L.ptr(C1), N, 0, 0, L.ptr(ws), ws.numel(), L.stream_ptr())
        L.check(rc, "seedmi_gemm_bf16_ws")
        torch.cuda.synchronize()
        assert torch.equal(C0.view(torch.int16), C1.view(torch.int16)), f"stream-K result differs from data-parallel (rep {rep})"
        # every flag was cleared by the workgroup that consumed it, and nobody gave up waiting (sticky error word = last word)
        assert not ws[:4096].any(), "stream-K flag area is not all-zero after the launch"
    # hipGraph replay: the epoch is frozen into the captured launch; replays rely on the consumed flags having been cleared
    C2 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")

    def launch():
        L.check(lib.seedmi_gemm_bf16_ws(M, N, K, L.ptr(A), K, L.ptr(W), K, L.ptr(bias), L.ptr(res), 0 if res is None else N, code,
                                        L.ptr(C2), N, 0, 0, L.ptr(ws), ws.numel(), L.stream_ptr()), "seedmi_gemm_bf16_ws")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        launch()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        launch()
    for rep in range(3):
        C2.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(C0.view(torch.int16), C2.view(torch.int16)), f"stream-K graph replay {rep} differs from data-parallel"
        assert not ws[:4096].any()
    rows = torch.cat([torch.arange(0, 300), torch.arange(M - 300, M)])          # fp32 check on the first / last tiles' rows
    acc = A[rows].float() @ W.float().t() + bias.float()
    want = {"bias": r(acc), "gelu": r(gelu(r(acc))), "residual": r(r(acc) + (res[rows].float() if res is not None else 0))}[epi]
    lib.seedmi_set_option(b"gemm", 0)
    assert_close_bf16(C0[rows], want, f"gemm_streamk[{variant}] {M}x{N}x{K} {epi}", frac=0.998)


@pytest.mark.parametrize("M,N,K,epi", [(300, 4224, 1408, "bias"), (514, 6144, 1408, "gelu"), (2000, 768, 768, "bias"),
                                       (4113, 4224, 1408, "bias")])      # last: odd M, 289 tiles (a workgroup walks two: both operand sets)
def test_gemm_layernorm_fold(lib, gemm_variant, M, N, K, epi):
    """seedmi_gemm_bf16_ext, consumer side: LayerNorm(x) W^T + b (eva_vit.py:199-202 + :135 / :60) computed as
    rstd * (x (W * gamma)^T - mean * colsum) + (b + W beta) with the statistics from seedmi_layernorm_stats_bf16, against the fp32
    restatement (LayerNorm in fp32, one rounding at the end) and against the reference's choreography (LayerNorm output rounded to half
    before the GEMM): the fold must not be further from fp32 than the reference's own rounding point puts it."""
    import ctypes
    gen = torch.Generator().manual_seed(M + N)
    x = bf(rand(gen, M, K) * (1.0 + 2.0 * torch.rand(M, 1, generator=gen)) + 0.7 * rand(gen, M, 1))     # rows with their own mean / scale
    W = bf(rand(gen, N, K, scale=0.03))
    b = bf(rand(gen, N, scale=0.2))
    gamma = bf(1.0 + 0.2 * rand(gen, K))
    beta = bf(0.1 * rand(gen, K))
    eps = 1e-6
    xd = x.cuda()
    stats = torch.zeros(M + (M & 1), 2, dtype=torch.float32, device="cuda")      # an even number of rows (fetched in pairs)
    L.check(lib.seedmi_layernorm_stats_bf16(L.ptr(xd), K, M, K, eps, L.ptr(stats), L.stream_ptr()), "stats")
    torch.cuda.synchronize()
    stats = stats[:M]
    mu = x.double().mean(1)
    rstd = torch.rsqrt(x.double().var(1, unbiased=False) + eps)
    assert torch.allclose(stats[:, 0].cpu().double(), mu, rtol=1e-5, atol=1e-6)
    assert torch.allclose(stats[:, 1].cpu().double(), rstd, rtol=1e-5)
    Wg = bf(W.float() * gamma.float().unsqueeze(0))
    cs = Wg.float().sum(1).cuda()
    bfold = (b.float() + W.float() @ beta.float()).cuda()
    Wgd = Wg.cuda()
    C = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    ext = L.GemmExt(L.ptr(stats), L.ptr(cs), L.ptr(bfold), None, 0)
    code = L.EPI_BIAS_GELU if epi == "gelu" else L.EPI_BIAS
    L.check(lib.seedmi_gemm_bf16_ext(M, N, K, L.ptr(xd), K, L.ptr(Wgd), K, None, None, 0, code, L.ptr(C), N, 0, 0, ctypes.byref(ext), None, 0,
                                     L.stream_ptr()), "gemm_ext")
    torch.cuda.synchronize()
    ln32 = torch.nn.functional.layer_norm(x.float(), (K,), gamma.float(), beta.float(), eps)
    y32 = ln32.double() @ W.double().t() + b.double()
    yref = r(ln32).double() @ W.double().t() + b.double()                                       # the reference rounds LN(x) to half first
    if epi == "gelu":
        want32, wantref = gelu(r(y32.float())), gelu(r(yref.float()))
    else:
        want32, wantref = y32.float(), yref.float()
    got = C.float().cpu()
    e_fold = ((got - want32).norm() / want32.norm()).item()
    e_ref = ((r(wantref) - want32).norm() / want32.norm()).item()
    print(f"[ln fold {M}x{N}x{K} {epi}] rel err vs fp32: folded GEMM {e_fold:.3e}, reference choreography {e_ref:.3e}")
    assert e_fold < max(1.2 * e_ref, 3e-3), (e_fold, e_ref)
    # element-wise: GELU's negative tail turns one half-ulp of its input into several ulps of a tiny output, hence the lower fraction
    assert_close_bf16(C, r(want32), f"gemm_lnfold {M}x{N}x{K} {epi}", atol_ulps=2.0, frac=0.95 if epi == "gelu" else 0.995)


def test_gemm_residual_emits_layernorm_statistics(lib, gemm_variant):
    """seedmi_gemm_bf16_ext, producer side: the BIAS_RESIDUAL epilogue writes (sum, sum of squares) of its half outputs per row and
    64-column span; seedmi_layernorm_stats_finalize turns them into the (mean, rstd) of the rows it wrote."""
    import ctypes
    gen = torch.Generator().manual_seed(77)
    M, N, K = 1030, 1408, 1408
    A = bf(rand(gen, M, K)).cuda()
    W = bf(rand(gen, N, K, scale=0.03)).cuda()
    bias = bf(rand(gen, N, scale=0.1)).cuda()
    x = bf(rand(gen, M, N) * 2.0 + 0.5).cuda()
    spans = (N + 63) // 64
    part = torch.full((spans, M, 2), float("nan"), dtype=torch.float32, device="cuda")       # span-major planes, stats_ld = M
    ext = L.GemmExt(None, None, None, L.ptr(part), M)
    C0 = run_gemm(lib, A, W, bias, x, L.EPI_BIAS_RESIDUAL)
    C1 = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    L.check(lib.seedmi_gemm_bf16_ext(M, N, K, L.ptr(A), K, L.ptr(W), K, L.ptr(bias), L.ptr(x), N, L.EPI_BIAS_RESIDUAL, L.ptr(C1), N, 0, 0,
                                     ctypes.byref(ext), None, 0, L.stream_ptr()), "gemm_ext")
    stats = torch.empty(M, 2, dtype=torch.float32, device="cuda")
    L.check(lib.seedmi_layernorm_stats_finalize(L.ptr(part), spans, M, M, N, 1e-6, L.ptr(stats), L.stream_ptr()), "finalize")
    torch.cuda.synchronize()
    assert torch.equal(C0, C1)                                            # emitting statistics does not change the output
    y = C1.double().cpu()
    assert torch.allclose(part[:, :, 0].sum(0).cpu().double(), y.sum(1), rtol=1e-5, atol=1e-3)
    assert torch.allclose(stats[:, 0].cpu().double(), y.mean(1), rtol=1e-4, atol=1e-5)
    assert torch.allclose(stats[:, 1].cpu().double(), torch.rsqrt(y.var(1, unbiased=False) + 1e-6), rtol=1e-4)


@pytest.mark.parametrize("sched", [0, -1], ids=["sched0", "default"])
def test_layernorm_statistics_by_tile_with_outlier_channels(lib, sched):
    """The LayerNorm fold chain as the tokenizer runs it at large batch (eva_vit.py:199-202): proj / fc2 (BIAS_RESIDUAL) emit one
    (sum, sum of squares) pair per row and 256-column TILE, the consuming qkv / fc1 GEMM finalizes its tiles' rows itself - no
    seedmi_layernorm_stats_finalize launch - on a residual stream with MASSIVE-ACTIVATION channels (|x| ~ 300 in three channels with a
    small gamma, plus rows with a large common mean: where the one-pass variance E[x^2] - mean^2 and the mean * colsum cancellation of the
    fold are weakest).  Checked: the tile planes against the span planes and fp64 sums; the in-kernel finalize against the
    finalize kernel and against the two-pass seedmi_layernorm_stats_bf16; the folded GEMM from tile statistics against the explicit
    LayerNorm -> GEMM path and the fp64 restatement (not further from it than the reference's own rounding choreography)."""
    import ctypes
    gen = torch.Generator().manual_seed(2024)
    M, D, N2 = 257 * 32, 1408, 4224                       # 33 m-tiles: every GEMM of the chain takes the 256x256 kernel
    assert lib.seedmi_gemm_tile_stats_supported(M, D) == 1 and lib.seedmi_gemm_tile_stats_supported(M, N2) == 1
    eps = 1e-6
    x_in = rand(gen, M, D)
    hot = [7, 500, 1300]
    x_in[:, hot] *= 300.0                                  # massive-activation channels
    x_in[::5] += 40.0 * torch.rand(M, 1, generator=gen)[::5]      # rows with a large common mean
    x_in = bf(x_in)
    A = bf(rand(gen, M, D)).cuda()
    Wp = bf(rand(gen, D, D, scale=0.03)).cuda()
    bp = bf(rand(gen, D, scale=0.1)).cuda()
    gamma = 1.0 + 0.2 * rand(gen, D)
    gamma[hot] = 0.01                                      # what a trained block does to such channels
    gamma, beta = bf(gamma), bf(rand(gen, D, scale=0.1))
    W2 = bf(rand(gen, N2, D, scale=0.03))
    b2 = bf(rand(gen, N2, scale=0.2))
    Wg = bf(W2.float() * gamma.float())
    cs = Wg.float().sum(1).cuda()
    bfold = (b2.float() + W2.float() @ beta.float()).cuda()
    L.check(lib.seedmi_set_option(b"gemm_sched", sched), "gemm_sched")
    try:
        xd = x_in.cuda()
        ld = (M + 1) & ~1
        planes = torch.full((6, ld, 2), float("nan"), dtype=torch.float32, device="cuda")
        spans = torch.full((22, M, 2), float("nan"), dtype=torch.float32, device="cuda")
        y_t = torch.zeros(M, D, dtype=torch.bfloat16, device="cuda")
        y_s = torch.zeros(M, D, dtype=torch.bfloat16, device="cuda")
        e_t = L.GemmExt(None, None, None, L.ptr(planes), ld, 1)
        e_s = L.GemmExt(None, None, None, L.ptr(spans), M)
        for ext, out in ((e_t, y_t), (e_s, y_s)):
            L.check(lib.seedmi_gemm_bf16_ext(M, D, D, L.ptr(A), D, L.ptr(Wp), D, L.ptr(bp), L.ptr(xd), D, L.EPI_BIAS_RESIDUAL, L.ptr(out), D, 0, 0,
                                             ctypes.byref(ext), None, 0, L.stream_ptr()), "producer")
        torch.cuda.synchronize()
        assert torch.equal(y_t, y_s)                       # the statistics layout does not touch the output
        y = y_t.double().cpu()
        pt, ps = planes[:, :M].double().cpu(), spans.double().cpu()
        assert torch.isfinite(pt).all()
        # a tile's pair is the sum of its (up to) four spans, in span order: EQUAL in fp32; all of them are the row's sums
        for t in range(6):
            acc = torch.zeros(M, 2, dtype=torch.float32)
            for sp in range(4 * t, min(4 * t + 4, 22)):
                acc += spans[sp].cpu()
            assert torch.equal(planes[t, :M].cpu(), acc), f"tile plane {t} is not the in-order sum of its spans"
        assert torch.allclose(pt.sum(0)[:, 0], y.sum(1), rtol=1e-5, atol=1e-2)
        assert torch.allclose(pt.sum(0)[:, 1], (y * y).sum(1), rtol=1e-5)
        # finished statistics three ways: finalize kernel on the span planes, two-pass kernel on y, fp64
        st_fin = torch.empty(M + 1, 2, dtype=torch.float32, device="cuda")
        st_two = torch.empty(M + 1, 2, dtype=torch.float32, device="cuda")
        L.check(lib.seedmi_layernorm_stats_finalize(L.ptr(spans), 22, M, M, D, eps, L.ptr(st_fin), L.stream_ptr()), "finalize")
        L.check(lib.seedmi_layernorm_stats_bf16(L.ptr(y_t), D, M, D, eps, L.ptr(st_two), L.stream_ptr()), "two-pass")
        torch.cuda.synchronize()
        mean64, rstd64 = y.mean(1), torch.rsqrt(y.var(1, unbiased=False) + eps)
        for nm, st in (("one-pass (spans)", st_fin), ("two-pass", st_two)):
            em = ((st[:M, 0].double().cpu() - mean64).abs() / (mean64.abs() + y.std(1))).max().item()
            er = ((st[:M, 1].double().cpu() - rstd64).abs() / rstd64).max().item()
            print(f"[outlier stats {nm}] max rel err: mean {em:.2e} rstd {er:.2e}")
            assert em < 1e-5 and er < 2e-4, (nm, em, er)
        # consumer: tile planes finalized inside the GEMM vs finished statistics vs explicit LayerNorm
        Wgd, W2d, b2d = Wg.cuda(), W2.cuda(), b2.cuda()
        z_tile = torch.zeros(M, N2, dtype=torch.bfloat16, device="cuda")
        z_fin = torch.zeros(M, N2, dtype=torch.bfloat16, device="cuda")
        e_c = L.GemmExt(L.ptr(planes), L.ptr(cs), L.ptr(bfold), None, 0, 0, 6, ld, D, eps)
        e_f = L.GemmExt(L.ptr(st_fin), L.ptr(cs), L.ptr(bfold), None, 0)
        for ext, out in ((e_c, z_tile), (e_f, z_fin)):
            L.check(lib.seedmi_gemm_bf16_ext(M, N2, D, L.ptr(y_t), D, L.ptr(Wgd), D, None, None, 0, L.EPI_BIAS, L.ptr(out), N2, 0, 0,
                                             ctypes.byref(ext), None, 0, L.stream_ptr()), "consumer")
        xn = torch.empty(M, D, dtype=torch.bfloat16, device="cuda")
        L.check(lib.seedmi_layernorm_bf16(L.ptr(y_t), D, L.ptr(gamma.cuda()), L.ptr(beta.cuda()), eps, L.ptr(xn), D, M, D, L.stream_ptr()), "ln")
        z_exp = run_gemm(lib, xn, W2d, b2d, None, L.EPI_BIAS)
        torch.cuda.synchronize()
        same = (z_tile.view(torch.int16) == z_fin.view(torch.int16)).float().mean().item()
        print(f"[outlier fold] in-kernel finalize vs finalize kernel: {same:.6f} of the outputs bit-equal")
        assert same > 0.999                                # (the two sums associate differently: the last ulp of rstd may differ)
        rows = torch.arange(0, M, 7)                       # fp64 restatement on a row sample
        ln64 = torch.nn.functional.layer_norm(y[rows], (D,), gamma.double(), beta.double(), eps)
        want = ln64 @ W2.double().t() + b2.double()
        def rel(t):
            return ((t[rows.cuda()].double().cpu() - want).norm() / want.norm()).item()
        e_tile, e_fin, e_exp = rel(z_tile), rel(z_fin), rel(z_exp)
        print(f"[outlier fold] rel err vs fp64: fold from tile statistics {e_tile:.3e}, from finished statistics {e_fin:.3e}, explicit LayerNorm -> GEMM {e_exp:.3e}")
        assert e_tile < max(1.2 * e_exp, 3e-3), (e_tile, e_exp)
    finally:
        lib.seedmi_set_option(b"gemm_sched", -1)


def test_gemm_residual_inplace(lib, gemm_variant):
    """The tokenizer calls proj/fc2 with C aliasing the residual (x += ...)."""
    gen = torch.Generator().manual_seed(12)
    M, N, K = 514, 1408, 1408
    A = bf(rand(gen, M, K)).cuda()
    W = bf(rand(gen, N, K, scale=0.03)).cuda()
    bias = bf(rand(gen, N, scale=0.1)).cuda()
    x = bf(rand(gen, M, N)).cuda()
    want = r(r(A.float() @ W.float().t() + bias.float()) + x.float())
    run_gemm(lib, A, W, bias, x, L.EPI_BIAS_RESIDUAL, C_init=x)
    assert_close_bf16(x, want, "gemm_residual_inplace")


def test_patch_embed_path(lib, gemm_variant):
    """im2col + GEMM(PATCH_EMBED) + cls rows == conv2d + cat(cls) + pos (eva_vit.py:224-230, 369-377)."""
    gen = torch.Generator().manual_seed(13)
    B, D, S, P = 3, 256, 56, 14
    g = S // P
    img = rand(gen, B, 3, S, S).cuda()
    wconv = bf(rand(gen, D, 3, P, P, scale=0.04)).cuda()
    bconv = bf(rand(gen, D, scale=0.1)).cuda()
    pos = bf(rand(gen, g * g + 1, D, scale=0.1)).cuda()
    cls = bf(rand(gen, D, scale=0.1)).cuda()
    kpad = 640
    col = torch.empty(B * g * g, kpad, dtype=torch.bfloat16, device="cuda")
    for is_fp32, src in ((1, img), (0, bf(img))):
        L.check(lib.seedmi_im2col_patch(L.ptr(src), is_fp32, L.ptr(col), B, 3, S, P, kpad, L.stream_ptr()), "im2col")
        torch.cuda.synchronize()
        ref_col = torch.nn.functional.unfold(r(img), kernel_size=P, stride=P).transpose(1, 2).reshape(B * g * g, -1)
        assert torch.equal(col[:, :588].float(), ref_col), "im2col mismatch"
        assert (col[:, 588:] == 0).all()
    wpad = torch.zeros(D, kpad, dtype=torch.bfloat16, device="cuda")
    wpad[:, :588] = wconv.reshape(D, -1)
    x = torch.zeros(B * (g * g + 1), D, dtype=torch.bfloat16, device="cuda")
    run_gemm(lib, col, wpad, bconv, pos, L.EPI_PATCH_EMBED, row_group=g * g, row_extra=1, C_init=x)
    cls_pos0 = bf(cls.float() + pos[0].float())
    L.check(lib.seedmi_fill_rows(L.ptr(x), D, g * g + 1, 0, B, L.ptr(cls_pos0), D, 1, D, L.stream_ptr()), "fill_rows")
    torch.cuda.synchronize()
    conv = r(torch.nn.functional.conv2d(r(img), wconv.float(), bconv.float(), stride=P))
    tok = conv.flatten(2).transpose(1, 2)
    want = r(torch.cat((cls.float().expand(B, 1, D), tok), 1) + pos.float())
    assert_close_bf16(x.view(B, g * g + 1, D), want, "patch_embed")


@pytest.mark.parametrize("rows,cols,eps", [(514, 1408, 1e-6), (96, 768, 1e-12), (33, 128, 1e-5), (40, 704, 1e-6)])
def test_layernorm(lib, rows, cols, eps):
    gen = torch.Generator().manual_seed(rows)
    x = bf(rand(gen, rows, cols, scale=2.0) + 0.5).cuda()
    w = bf(1 + rand(gen, cols, scale=0.1)).cuda()
    b = bf(rand(gen, cols, scale=0.1)).cuda()
    out = torch.empty_like(x)
    L.check(lib.seedmi_layernorm_bf16(L.ptr(x), cols, L.ptr(w), L.ptr(b), eps, L.ptr(out), cols, rows, cols,
                                      L.stream_ptr()), "layernorm")
    torch.cuda.synchronize()
    want = r(torch.nn.functional.layer_norm(x.float(), (cols,), w.float(), b.float(), eps))
    assert_close_bf16(out, want, f"layernorm {rows}x{cols}", atol_ulps=1.01)


@pytest.mark.parametrize("rows,cols", [(64, 4096), (7, 5120), (50, 256)])
def test_rmsnorm(lib, rows, cols):
    gen = torch.Generator().manual_seed(cols)
    x = bf(rand(gen, rows, cols)).cuda()
    w = bf(1 + rand(gen, cols, scale=0.1)).cuda()
    out = torch.empty_like(x)
    L.check(lib.seedmi_rmsnorm_bf16(L.ptr(x), cols, L.ptr(w), 1e-6, L.ptr(out), cols, rows, cols, L.stream_ptr()), "rmsnorm")
    torch.cuda.synchronize()
    xf = x.float()
    h = r(xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6))
    want = r(w.float() * h)
    assert_close_bf16(out, want, f"rmsnorm {rows}x{cols}", atol_ulps=1.01)


def ref_attention(q, k, v, heads, scale, causal, round_s=True):
    """q [B,nq,H*hd] etc. fp32 holding bf16 values; reference rounding points (eva_vit.py:139-156)."""
    B, nq, C = q.shape
    nk = k.shape[1]
    hd = C // heads
    qh = r(q.view(B, nq, heads, hd).transpose(1, 2) * scale)
    kh = k.view(B, nk, heads, hd).transpose(1, 2)
    vh = v.view(B, nk, heads, hd).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2)
    if round_s:
        s = r(s)
    if causal:
        keep = torch.ones(nq, nk, dtype=torch.bool, device=q.device).tril()
        s = s.masked_fill(~keep, float("-inf"))
    p = r(torch.softmax(s, dim=-1))
    return r(p @ vh).transpose(1, 2).reshape(B, nq, C)


DEFAULT_ATTN_VIT = 5          # the library's default ViT attention kernel (seed_amd/csrc/attn_vit.hip: g_attn_vit)


@pytest.mark.parametrize("B,H,hd,nq,nk,causal", [
    (3, 16, 88, 257, 257, False),     # ViT
    (40, 16, 88, 257, 257, False),    # ViT, more items than CUs (persistent kernel walks > 1 item per workgroup)
    (72, 16, 88, 257, 257, False),    # ViT, uneven rounds of the staggered kernel's XCD-aware item walk (9 images per XCD: 4.5 rounds of its 32 workgroups; 40 images: 5 per XCD)
    (2, 3, 88, 100, 100, False),
    (2, 4, 88, 17, 17, False),        # small ViT (NKP=32 path)
    (5, 12, 64, 32, 32, True),        # Q-Former causal self-attention
    (4, 12, 64, 32, 257, False),      # Q-Former cross-attention
    (2, 2, 64, 17, 17, False),
])
@pytest.mark.parametrize("trv", [6, 4, 3, 2, 1, 0], ids=["vit_16wave_staggered", "vit_16wave_wide", "vit_16wave", "vit_pipeline", "tr_read", "vt_image"])
def test_attention_fullrow(lib, B, H, hd, nq, nk, causal, trv):
    if trv >= 3 and not (nq == nk == 257 and hd == 88 and not causal):
        pytest.skip("the 16-wave kernel serves the 257-token ViT shape only (other shapes take the same kernels as vit_pipeline)")
    if B == 72 and trv not in (6, 4):
        pytest.skip("the 72-image case exists for the staggered kernel's XCD-aware walk (and one lock-step run beside it)")
    default_vit = DEFAULT_ATTN_VIT
    L.check(lib.seedmi_set_option(b"attn_vit", {6: 5, 4: 3, 3: 2, 2: 1}.get(trv, 0)), "set_option")
    L.check(lib.seedmi_set_option(b"attn_trv", min(trv, 1)), "set_option")
    gen = torch.Generator().manual_seed(B * 100 + nk)
    C = H * hd
    # packed [q|k|v] buffer like the ViT QKV GEMM output
    if nq == nk:
        qkv = bf(rand(gen, B * nq, 3 * C)).cuda()
        Q, K, V, ld = qkv, qkv[:, C:], qkv[:, 2 * C:], 3 * C
        ldq = ld
    else:
        qb = bf(rand(gen, B * nq, C)).cuda()
        kv = bf(rand(gen, B * nk, 2 * C)).cuda()
        Q, K, V, ldq, ld = qb, kv, kv[:, C:], C, 2 * C
    out = torch.zeros(B * nq, C, dtype=torch.bfloat16, device="cuda")
    scale = hd ** -0.5
    rc = lib.seedmi_attention_bf16(L.ptr(Q), ldq, L.ptr(K), ld, L.ptr(V), ld, L.ptr(out), C, B, H, hd, nq, nk, scale,
                                   1 if causal else 0, 1, L.stream_ptr())
    L.check(rc, "attention")
    torch.cuda.synchronize()
    qf = Q[:, :C].float().view(B, nq, C)
    kf = K[:, :C].float().reshape(B, nk, C)
    vf = V[:, :C].float().reshape(B, nk, C)
    want = ref_attention(qf, kf, vf, H, scale, causal)
    lib.seedmi_set_option(b"attn_trv", 1)
    if trv == 6:
        # the staggered form does every wave's work exactly as the lock-step 16-wave kernel does: bit-identical, row 256 included
        # (the "flash"-normalisation pair 4 / 6 of attn_vit moves a rounding point and lives in the devtools build only)
        lib.seedmi_set_option(b"attn_vit", 3)
        ref16 = torch.full_like(out, float("nan"))
        L.check(lib.seedmi_attention_bf16(L.ptr(Q), ldq, L.ptr(K), ld, L.ptr(V), ld, L.ptr(ref16), C, B, H, hd, nq, nk, scale, 0, 1,
                                          L.stream_ptr()), "attention")
        torch.cuda.synchronize()
        assert torch.equal(out.view(torch.int16), ref16.view(torch.int16)), "the staggered 16-wave kernel differs from the lock-step one"
    if trv in (3, 4, 6):
        # same arithmetic and rounding points as the 12-wave kernel (row 256 goes through a differently shaped reduction: compared to 1 ulp)
        lib.seedmi_set_option(b"attn_vit", 1)
        ref12 = torch.zeros_like(out)
        L.check(lib.seedmi_attention_bf16(L.ptr(Q), ldq, L.ptr(K), ld, L.ptr(V), ld, L.ptr(ref12), C, B, H, hd, nq, nk, scale, 0, 1,
                                          L.stream_ptr()), "attention")
        torch.cuda.synchronize()
        o3, r3 = out.view(B, nq, C), ref12.view(B, nq, C)
        assert torch.equal(o3[:, :256], r3[:, :256]), "rows 0..255 of the 16-wave kernel differ from the 12-wave kernel"
        assert_close_bf16(o3[:, 256], r3[:, 256].float(), "attention row 256 (side path) vs the 12-wave kernel", atol_ulps=1.0, frac=0.99)
    lib.seedmi_set_option(b"attn_vit", default_vit)
    assert_close_bf16(out.view(B, nq, C), want, f"attention hd{hd} {nq}x{nk} causal={causal} trv={trv}", atol_ulps=2.5, frac=0.995)


def test_vq_argmin_bit_exact(lib, golden_dir):
    """ids identical to the oracle (fixed summation order) and to the reference module's golden ids, incl. ties."""
    from oracle import seed_oracle as O
    g = np.load(os.path.join(golden_dir, "vq_reference.npz"))
    z = torch.from_numpy(g["z"]).reshape(-1, 32)
    cb = torch.from_numpy(g["codebook"])
    gen = torch.Generator().manual_seed(5)
    z_big = torch.cat([z, rand(gen, 1000, 32, scale=0.3)], 0)          # ragged row count (1128 rows: the batch shape; 128 and 33 rows: the small-launch shape)
    for zz in (z, z_big, z_big[95:128]):
        zd, cbd = bf(zz).cuda(), bf(cb).cuda()
        ee = torch.empty(cb.shape[0], dtype=torch.float32, device="cuda")
        ids = torch.full((zz.shape[0],), -1, dtype=torch.int64, device="cuda")
        L.check(lib.seedmi_vq_code_sqnorm(L.ptr(cbd), L.ptr(ee), cb.shape[0], 32, L.stream_ptr()), "sqnorm")
        L.check(lib.seedmi_vq_argmin_bf16(L.ptr(zd), 32, L.ptr(cbd), L.ptr(ee), L.ptr(ids), zz.shape[0], cb.shape[0], 32,
                                          L.stream_ptr()), "vq")
        torch.cuda.synchronize()
        want = O.vq_argmin(zz, cb, O.Prec("bf16"))
        assert torch.equal(ids.cpu(), want), f"{(ids.cpu() != want).sum().item()} / {want.numel()} ids differ from the oracle"
    ref_ids = torch.from_numpy(g["ids_bf16"]).reshape(-1)
    ids_small = O.vq_argmin(z, cb, O.Prec("bf16"))
    assert ids_small[0] == 17 and ids_small[1] == 17
    # vs the reference module itself on the committed vectors: EQUAL (128 rows incl. the forced ties; measured 0 / 640 over the VQ and
    # full-size goldens in both precisions)
    assert torch.equal(ids_small, ref_ids), f"{(ids_small != ref_ids).sum().item()} ids differ from VectorQuantizer2's"


@pytest.mark.parametrize("B", [1, 2, 4, 7])
def test_vit_attention_small_launch_split_is_bit_identical(lib, B):
    """Round 6: a ViT attention launch with fewer (image, head) items than half the CUs - one image is 16 items for 256 CUs, and one image is
    what the reference scripts tokenize - splits every item's query tiles over 2 / 4 / 8 / 16 workgroups of the lock-step 16-wave kernel.
    A query tile is one wave's private work, so the split may not change a bit: every factor against the unsplit staggered kernel (the one
    large batches take), i.e. an image alone and the same image inside a 256-batch still see the same attention output."""
    H, hd, n = 16, 88, 257
    C = H * hd
    gen = torch.Generator().manual_seed(40 + B)
    qkv = bf(rand(gen, B * n, 3 * C)).cuda()
    scale = hd ** -0.5

    def run(small):
        L.check(lib.seedmi_set_option(b"attn_small", small), "set_option")
        out = torch.full((B * n, C), float("nan"), dtype=torch.bfloat16, device="cuda")
        L.check(lib.seedmi_attention_bf16(L.ptr(qkv), 3 * C, L.ptr(qkv[:, C:]), 3 * C, L.ptr(qkv[:, 2 * C:]), 3 * C, L.ptr(out), C, B, H, hd, n, n, scale,
                                          0, 1, L.stream_ptr()), "attention")
        torch.cuda.synchronize()
        return out
    try:
        ref = run(0)                                             # the staggered 16-wave kernel, one workgroup per item
        assert torch.isfinite(ref.float()).all()
        for small in (1, 2, 4, 8, 16):
            got = run(small)
            assert torch.equal(got.view(torch.int16), ref.view(torch.int16)), f"attn_small={small} changed {(got != ref).sum().item()} values"
    finally:
        lib.seedmi_set_option(b"attn_small", 1)
    want = ref_attention(qkv[:, :C].float().view(B, n, C), qkv[:, C:2 * C].float().reshape(B, n, C), qkv[:, 2 * C:].float().reshape(B, n, C), H, scale, False)
    assert_close_bf16(ref.view(B, n, C), want, f"vit attention B={B}", atol_ulps=2.5, frac=0.995)


@pytest.mark.parametrize("mode,dtype", [("bf16", torch.bfloat16), ("fp16", torch.float16)])
def test_vq_argmin_nonfinite_rows_follow_torch_argmin(mode, dtype):
    """a row whose distances are all +inf (fp16: |z|^2 beyond 65504) or contain NaN.  torch.argmin
    (qformer_quantizer.py:98) returns index 0 / the FIRST NaN; the kernel used to leave its start value 0x7fffffff in ids_i64 (an
    out-of-range embedding row once 32000 is added).  Both builds, plain sweep and the head-fused sweep, against the oracle - which
    tests/test_oracle_golden.py::test_vq_nonfinite_rows_follow_torch_argmin pins on torch and on the reference module."""
    from oracle import seed_oracle as O
    from _cases import nonfinite_vq_case
    lib = L.load(dtype)
    z, cb, cb_nan = nonfinite_vq_case(dtype)
    for e in (cb, cb_nan):
        zd, cbd = z.to(dtype).cuda(), e.to(dtype).cuda()
        ee = torch.empty(e.shape[0], dtype=torch.float32, device="cuda")
        ids = torch.full((z.shape[0],), -1, dtype=torch.int64, device="cuda")
        L.check(lib.seedmi_vq_code_sqnorm(L.ptr(cbd), L.ptr(ee), e.shape[0], 32, L.stream_ptr()), "sqnorm")
        L.check(lib.seedmi_vq_argmin_bf16(L.ptr(zd), 32, L.ptr(cbd), L.ptr(ee), L.ptr(ids), z.shape[0], e.shape[0], 32, L.stream_ptr()), "vq")
        torch.cuda.synchronize()
        want = O.vq_argmin(z, e, O.Prec(mode))
        assert int(ids.min()) >= 0 and int(ids.max()) < e.shape[0], ids.tolist()
        assert torch.equal(ids.cpu(), want), (mode, ids.cpu().tolist(), want.tolist())
        # a codebook shorter than one sweep step (lanes without a code must never win a tie on +inf)
        n_small = 100
        ids_s = torch.full((z.shape[0],), -1, dtype=torch.int64, device="cuda")
        cbs = cbd[:n_small].contiguous()
        L.check(lib.seedmi_vq_code_sqnorm(L.ptr(cbs), L.ptr(ee), n_small, 32, L.stream_ptr()), "sqnorm")
        L.check(lib.seedmi_vq_argmin_bf16(L.ptr(zd), 32, L.ptr(cbs), L.ptr(ee), L.ptr(ids_s), z.shape[0], n_small, 32, L.stream_ptr()), "vq")
        torch.cuda.synchronize()
        assert torch.equal(ids_s.cpu(), O.vq_argmin(z, e[:n_small], O.Prec(mode)))


@pytest.mark.parametrize("rows", [1003, 37])
def test_vq_head_argmin_fused(lib, rows):
    """seedmi_vq_head_argmin_bf16 (SURVEY 8a a13 -> a14): Linear(768, 32) + bias of encode_task_layer (qformer_quantizer.py:219-223) fused in
    front of the VQ sweep.  z equals the fp64 Linear rounded to half on all but a handful of round-to-nearest ties of the fp32 chain, and the
    ids are EXACTLY the oracle's argmin of the z the kernel itself produced (ragged row count).  1003 rows take the batch shape of the sweep
    (8 rows x 4 waves per workgroup), 37 the small-launch shape (1 row x 16 waves: one image is 32 rows)."""
    from oracle import seed_oracle as O
    gen = torch.Generator().manual_seed(21)
    hidden, n_embed = 768, 8192
    t = bf(torch.tanh(rand(gen, rows, hidden)))
    w = bf(rand(gen, 32, hidden, scale=0.05))
    b = bf(rand(gen, 32, scale=0.1))
    cb = bf(rand(gen, n_embed, 32, scale=0.35))
    td, wd, bd, cbd = t.cuda(), w.cuda(), b.cuda(), cb.cuda()
    ee = torch.empty(n_embed, dtype=torch.float32, device="cuda")
    ids = torch.full((rows,), -1, dtype=torch.int64, device="cuda")
    z = torch.zeros(rows, 32, dtype=torch.bfloat16, device="cuda")
    L.check(lib.seedmi_vq_code_sqnorm(L.ptr(cbd), L.ptr(ee), n_embed, 32, L.stream_ptr()), "sqnorm")
    L.check(lib.seedmi_vq_head_argmin_bf16(L.ptr(td), hidden, hidden, L.ptr(wd), hidden, L.ptr(bd), L.ptr(cbd), L.ptr(ee), L.ptr(ids), L.ptr(z), 32,
                                           rows, n_embed, 32, L.stream_ptr()), "vq head")
    ids2 = torch.full((rows,), -1, dtype=torch.int64, device="cuda")
    L.check(lib.seedmi_vq_head_argmin_bf16(L.ptr(td), hidden, hidden, L.ptr(wd), hidden, L.ptr(bd), L.ptr(cbd), L.ptr(ee), L.ptr(ids2), None, 0,
                                           rows, n_embed, 32, L.stream_ptr()), "vq head, no tap")
    torch.cuda.synchronize()
    want_z = r((t.double() @ w.double().t() + b.double()).float())
    zc = z.float().cpu()
    exact = (zc == want_z).float().mean().item()
    print(f"[vq head] z equal to the fp64 Linear rounded to half: {exact:.5f}")
    assert exact > (0.995 if rows > 100 else 0.99)
    assert_close_bf16(z, want_z, "vq head z", atol_ulps=1.01, frac=1.0)
    want_ids = O.vq_argmin(zc, cb.float(), O.Prec("bf16"))
    assert torch.equal(ids.cpu(), want_ids)
    assert torch.equal(ids2.cpu(), want_ids)


def test_vq_codebook_sqnorm_exact(lib):
    from oracle import seed_oracle as O
    gen = torch.Generator().manual_seed(9)
    cb = bf(rand(gen, 8192, 32, scale=0.3))
    ee = torch.empty(8192, dtype=torch.float32, device="cuda")
    L.check(lib.seedmi_vq_code_sqnorm(L.ptr(cb.cuda()), L.ptr(ee), 8192, 32, L.stream_ptr()), "sqnorm")
    torch.cuda.synchronize()
    e = cb.float().numpy()
    acc = np.zeros(8192, np.float32)
    for k in range(32):
        acc = (acc + O._bf16_round_np(e[:, k] * e[:, k])).astype(np.float32)
    assert np.array_equal(ee.cpu().numpy(), O._bf16_round_np(acc))


# ------------------------------------------------------------------------------------------------ LLaMA pieces

@pytest.mark.parametrize("M,N,K,epi", [(32, 4096, 4096, "none"), (32, 512, 11008, "residual"), (4, 1024, 256, "swiglu"),
                                       (17, 40194, 256, "none"), (64, 768, 512, "residual"), (33, 256, 1024, "none")])
def test_gemm_skinny(lib, M, N, K, epi):
    gen = torch.Generator().manual_seed(M + N)
    A = bf(rand(gen, M, K)).cuda()
    W = bf(rand(gen, N, K, scale=0.03)).cuda()
    acc = A.float() @ W.float().t()
    res = None
    if epi == "residual":
        res = bf(rand(gen, M, N)).cuda()
        want, code, ncol = r(r(acc) + res.float()), L.EPI_BIAS_RESIDUAL, N
    elif epi == "swiglu":
        g, u = r(acc[:, 0::2]), r(acc[:, 1::2])
        want, code, ncol = r(r(torch.nn.functional.silu(g)) * u), L.EPI_SWIGLU, N // 2
    else:
        want, code, ncol = r(acc), L.EPI_NONE, N
    ldc = (ncol + 15) // 16 * 16
    C = torch.zeros(M, ldc, dtype=torch.bfloat16, device="cuda")
    rc = lib.seedmi_gemm_skinny_bf16(M, N, K, L.ptr(A), K, L.ptr(W), K, L.ptr(res), 0 if res is None else N, code,
                                     L.ptr(C), ldc, L.stream_ptr())
    L.check(rc, "gemm_skinny")
    torch.cuda.synchronize()
    assert_close_bf16(C[:, :ncol], want, f"gemm_skinny {M}x{N}x{K} {epi}", frac=0.998)
    # fragment-major (packed) weights: same arithmetic in the same order -> bit-identical output
    Wp = torch.empty(lib.seedmi_pack_skinny_weights_bytes(N, K) // 2, dtype=torch.bfloat16, device="cuda")
    L.check(lib.seedmi_pack_skinny_weights(L.ptr(W), K, N, K, L.ptr(Wp), L.stream_ptr()), "pack")
    C2 = torch.zeros_like(C)
    rc = lib.seedmi_gemm_skinny_packed_bf16(M, N, K, L.ptr(A), K, L.ptr(Wp), L.ptr(res), 0 if res is None else N, code,
                                            L.ptr(C2), ldc, 0, 0, L.stream_ptr())
    L.check(rc, "gemm_skinny_packed")
    torch.cuda.synchronize()
    assert torch.equal(C, C2), "packed-weight decode GEMM differs from the row-major one"


def _rope_tables(max_pos, hd):
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2).float() / hd))
    t = torch.arange(max_pos, dtype=torch.float32)
    emb = torch.cat((torch.outer(t, inv),) * 2, dim=-1)
    return bf(emb.cos()), bf(emb.sin())


def _ref_rope(x, cos, sin):
    h = x.shape[-1] // 2
    rot = torch.cat((-x[..., h:], x[..., :h]), -1)
    return r(r(x * cos) + r(rot * sin))


@pytest.fixture(params=[1, 0], ids=["prefill_tiled", "prefill_v1"])
def prefill_variant(request, lib):
    """Prefill attention runs once per kernel: the LDS-tiled one (default) and the first-round direct-from-L2 one."""
    L.check(lib.seedmi_set_option(b"prefill_tiled", request.param), "set_option")
    yield request.param
    lib.seedmi_set_option(b"prefill_tiled", 1)


@pytest.mark.parametrize("B,T,H,past", [(2, 12, 2, 0), (3, 1, 4, 9), (2, 70, 2, 0), (1, 5, 2, 7), (1, 649, 2, 0), (2, 130, 1, 63),
                                        (1, 128, 1, 0), (1, 129, 3, 1)])
def test_rope_and_llama_attention(lib, prefill_variant, B, T, H, past):
    hd, tmax = 128, max(128, past + T)
    gen = torch.Generator().manual_seed(B * 10 + T)
    h = H * hd
    cos_t, sin_t = _rope_tables(tmax, hd)
    cos_d, sin_d = cos_t.cuda(), sin_t.cuda()          # keep alive: the C ABI takes raw pointers
    kc = torch.zeros(B, H, tmax, hd, dtype=torch.bfloat16, device="cuda")
    vc = torch.zeros_like(kc)
    # pre-existing cache content for positions < past
    k_past = bf(rand(gen, B, H, past, hd)).cuda()
    v_past = bf(rand(gen, B, H, past, hd)).cuda()
    kc[:, :, :past] = k_past
    vc[:, :, :past] = v_past
    qkv = bf(rand(gen, B * T, 3 * h)).cuda()
    pos = torch.arange(past, past + T, dtype=torch.int64).unsqueeze(0).expand(B, T).contiguous().cuda()
    q_out = torch.empty(B * T, h, dtype=torch.bfloat16, device="cuda")
    rc = lib.seedmi_rope_kv_append(L.ptr(qkv), 3 * h, L.ptr(pos), L.ptr(cos_d), L.ptr(sin_d), L.ptr(q_out), h,
                                   L.ptr(kc), L.ptr(vc), B, T, H, hd, tmax, past, None, cos_d.shape[0], L.stream_ptr())
    L.check(rc, "rope_kv_append")
    torch.cuda.synchronize()
    qf = qkv[:, :h].float().view(B, T, H, hd).transpose(1, 2)
    kf = qkv[:, h:2 * h].float().view(B, T, H, hd).transpose(1, 2)
    vf = qkv[:, 2 * h:].float().view(B, T, H, hd).transpose(1, 2)
    c = cos_t.float().cuda()[pos].unsqueeze(1)
    s = sin_t.float().cuda()[pos].unsqueeze(1)
    q_ref, k_ref = _ref_rope(qf, c, s), _ref_rope(kf, c, s)
    assert torch.equal(q_out.float().view(B, T, H, hd).transpose(1, 2), q_ref), "rope(q) not bit-exact"
    assert torch.equal(kc[:, :, past:past + T].float(), k_ref), "rope(k) / cache append not bit-exact"
    assert torch.equal(vc[:, :, past:past + T].float(), vf)
    assert torch.equal(kc[:, :, :past], k_past)

    out = torch.zeros(B * T, h, dtype=torch.bfloat16, device="cuda")
    scale = 1.0 / math.sqrt(hd)
    rc = lib.seedmi_llama_attention_bf16(L.ptr(q_out), h, L.ptr(kc), L.ptr(vc), L.ptr(out), h, B, T, H, hd, tmax, past, scale,
                                         0, None, L.stream_ptr())
    L.check(rc, "llama_attention")
    torch.cuda.synchronize()
    kall = kc[:, :, :past + T].float()
    vall = vc[:, :, :past + T].float()
    sc = (q_ref @ kall.transpose(-1, -2)) * scale
    if T > 1:
        qi = torch.arange(T, device="cuda").unsqueeze(1) + past
        kj = torch.arange(past + T, device="cuda").unsqueeze(0)
        sc = sc.masked_fill(kj > qi, float("-inf"))
    p = r(torch.softmax(sc, -1))
    want = r(p @ vall).transpose(1, 2).reshape(B * T, h)
    assert_close_bf16(out, want, f"llama_attention B{B} T{T} past{past}", atol_ulps=2.5, frac=0.995)


@pytest.mark.parametrize("early", [0, 1, 2], ids=["rows_after_rotation", "keys_early", "keys_values_early"])
@pytest.mark.parametrize("B,H,past,dev_len,packed", [(3, 4, 0, False, False), (2, 2, 37, False, False), (32, 2, 100, True, True),
                                                      (17, 3, 126, True, False), (5, 2, 129, True, True), (4, 3, 300, False, False)])
def test_fused_decode_attention_equals_rope_then_attention(lib, B, H, past, dev_len, packed, early):
    """seedmi_llama_decode_attention_bf16 (RoPE + cache append + attention, one launch) against the two-kernel form on the same
    inputs: rotated key / value rows in the cache and the attention output must be BIT-identical, with the cache length as
    a launch argument or read from device memory, row-major or fragment-major output, for every request order of the cached rows
    (seedmi_set_option("decode_attn_early")) and for caches of one, two and three 128-row batches."""
    L.check(lib.seedmi_set_option(b"decode_attn_early", early), "decode_attn_early")
    try:
        _fused_decode_attention_case(lib, B, H, past, dev_len, packed)
    finally:
        L.check(lib.seedmi_set_option(b"decode_attn_early", 1), "decode_attn_early")


def _fused_decode_attention_case(lib, B, H, past, dev_len, packed):
    hd, tmax, T = 128, 128 if past < 127 else 384, 1
    gen = torch.Generator().manual_seed(1000 + B + past)
    h = H * hd
    cos_t, sin_t = _rope_tables(tmax, hd)
    cos_d, sin_d = cos_t.cuda(), sin_t.cuda()
    kc = torch.zeros(B, H, tmax, hd, dtype=torch.bfloat16, device="cuda")
    vc = torch.zeros_like(kc)
    kc[:, :, :past] = bf(rand(gen, B, H, past, hd)).cuda()
    vc[:, :, :past] = bf(rand(gen, B, H, past, hd)).cuda()
    kc2, vc2 = kc.clone(), vc.clone()
    qkv = bf(rand(gen, B, 3 * h)).cuda()
    past_d = torch.tensor([past], dtype=torch.int32, device="cuda") if dev_len else None
    pos = None if dev_len else torch.full((B, 1), past, dtype=torch.int64, device="cuda")
    scale = 1.0 / math.sqrt(hd)
    rows = (B + 15) // 16 * 16 if packed else B
    q_out = torch.empty(B, h, dtype=torch.bfloat16, device="cuda")
    out_a = torch.zeros(rows, h, dtype=torch.bfloat16, device="cuda")
    out_b = torch.zeros(rows, h, dtype=torch.bfloat16, device="cuda")
    L.check(lib.seedmi_rope_kv_append(L.ptr(qkv), 3 * h, L.ptr(pos), L.ptr(cos_d), L.ptr(sin_d), L.ptr(q_out), h, L.ptr(kc),
                                      L.ptr(vc), B, T, H, hd, tmax, past, L.ptr(past_d), cos_d.shape[0], L.stream_ptr()), "rope")
    L.check(lib.seedmi_llama_attention_bf16(L.ptr(q_out), h, L.ptr(kc), L.ptr(vc), L.ptr(out_a), h, B, T, H, hd, tmax, past, scale,
                                            1 if packed else 0, L.ptr(past_d), L.stream_ptr()), "attention")
    L.check(lib.seedmi_llama_decode_attention_bf16(L.ptr(qkv), 3 * h, L.ptr(pos), L.ptr(cos_d), L.ptr(sin_d), L.ptr(kc2), L.ptr(vc2),
                                                   L.ptr(out_b), h, B, H, hd, tmax, past, scale, 1 if packed else 0, L.ptr(past_d),
                                                   cos_d.shape[0], L.stream_ptr()), "fused decode attention")
    torch.cuda.synchronize()
    assert torch.equal(kc, kc2) and torch.equal(vc, vc2), "cache append differs"
    assert torch.equal(out_a, out_b), "attention output differs"
    assert out_b.float().abs().sum() > 0


def test_embed_rows(lib):
    gen = torch.Generator().manual_seed(2)
    table = bf(rand(gen, 1000, 256)).cuda()
    ids = torch.randint(0, 1000, (77,), generator=gen).cuda()
    out = torch.empty(77, 256, dtype=torch.bfloat16, device="cuda")
    L.check(lib.seedmi_embed_rows(L.ptr(ids), L.ptr(table), 256, L.ptr(out), 256, 77, 256, 1000, L.stream_ptr()), "embed")
    torch.cuda.synchronize()
    assert torch.equal(out, table[ids])


def test_error_reporting(lib):
    rc = lib.seedmi_gemm_bf16(16, 16, 100, None, 8, None, 8, None, None, 0, 0, None, 8, 0, 0, None)
    assert rc == -1 and b"multiple of 64" in lib.seedmi_last_error()
    rc = lib.seedmi_attention_bf16(None, 8, None, 8, None, 8, None, 8, 1, 1, 80, 4, 4, 1.0, 0, 1, None)
    assert rc != 0


def _unpack_activations(packed, rows, cols):
    """Inverse of the fragment-major activation layout: element (m, k) lives at
    (((m>>4)*(cols>>5) + (k>>5))*64 + ((k>>3)&3)*16 + (m&15))*8 + (k&7)."""
    m = torch.arange(rows).unsqueeze(1)
    k = torch.arange(cols).unsqueeze(0)
    idx = (((m >> 4) * (cols >> 5) + (k >> 5)) * 64 + ((k >> 3) & 3) * 16 + (m & 15)) * 8 + (k & 7)
    return packed.cpu()[idx]


def _skinny_ws(lib):
    """Zeroed workspace of the split-K decode GEMM (flag words first) + a checker: flags all zero again, error word clear."""
    nbytes = lib.seedmi_gemm_skinny_workspace_bytes()
    ws = torch.full((nbytes,), 0xa5, dtype=torch.uint8, device="cuda")        # (NOT zeroed: seedmi_gemm_skinny_workspace_init must do all of it)
    L.check(lib.seedmi_gemm_skinny_workspace_init(L.ptr(ws), ws.numel(), L.stream_ptr()), "workspace init")

    def check():
        flags = ws[:4096].view(torch.int32).clone()
        assert int(flags[1022]) == 0x5eed514b, "the workspace tag (word 1022) was overwritten"
        flags[1022] = 0
        assert int(flags.abs().sum()) == 0, f"split-K flag words not cleared / error word set: {flags.nonzero().flatten().tolist()[:8]}"
        L.check(lib.seedmi_gemm_skinny_ws_status(L.ptr(ws), ws.numel(), L.stream_ptr()), "status")
    return ws, check


def _skinny_norm(lib, form, ws, M, N, K, xp, Wp, eps, res, ldr, epi, C, ldc, c_packed, outp, what):
    """form 'one_tile': seedmi_gemm_skinny_norm_bf16; otherwise seedmi_gemm_skinny_norm_ws_bf16 with a workspace."""
    if form != "one_tile":
        L.check(lib.seedmi_gemm_skinny_norm_ws_bf16(M, N, K, xp, 1, Wp, eps, res, ldr, epi, C, ldc, c_packed, outp, L.ptr(ws), ws.numel(),
                                                    L.stream_ptr()), what)
    else:
        L.check(lib.seedmi_gemm_skinny_norm_bf16(M, N, K, xp, 1, Wp, eps, res, ldr, epi, C, ldc, c_packed, outp, L.stream_ptr()), what)


@pytest.mark.parametrize("form", ["one_tile", "split_k_by_shape", "split_k_cut"])
@pytest.mark.parametrize("M,N,K,epi", [(32, 768, 512, "none"), (7, 1536, 1024, "swiglu"), (17, 512, 1408, "residual"),
                                       (32, 12288, 4096, "none"), (64, 256, 256, "none"), (1, 64, 128, "residual"), (16, 4096, 4096, "residual"),
                                       # SEED-LLaMA-8B decode shapes (config 3): gate|up, down (K = 11008), lm_head (vocab 40194 -> 40208)
                                       (32, 22016, 4096, "swiglu"), (32, 4096, 11008, "residual"), (32, 40208, 4096, "none")])
def test_skinny_gemm_with_folded_rmsnorm(lib, M, N, K, epi, form):
    """seedmi_gemm_skinny_norm_bf16 / seedmi_gemm_skinny_norm_ws_bf16 (balanced split-K): un-normalised fragment-major rows in,
    weight * gamma streamed, row scale rsqrt(mean(x^2) + eps) applied to the fp32 accumulators; plus the fragment-major second copy of a
    residual result and seedmi_pack_activations_bf16 (pure permutation, bit-exact).  The split-K form must also leave its flag words
    zero and give the same bits when the launch is repeated (fixed summation order)."""
    if form != "one_tile" and M > 32:
        pytest.skip("the split-K form covers M <= 32 (two activation row tiles)")
    ws, ws_check = _skinny_ws(lib) if form != "one_tile" else (None, lambda: None)
    if form == "split_k_cut":
        L.check(lib.seedmi_set_option(b"skinny_splitk", 2), "skinny_splitk")       # 64-row tiles and the balanced cut for every shape (default: uncut where the shape divides)
    try:
        _skinny_folded_rmsnorm_case(lib, M, N, K, epi, form, ws, ws_check)
    finally:
        L.check(lib.seedmi_set_option(b"skinny_splitk", 1), "skinny_splitk")


def _skinny_folded_rmsnorm_case(lib, M, N, K, epi, form, ws, ws_check):
    gen = torch.Generator().manual_seed(M * 7 + N)
    eps = 1e-6
    x = bf(rand(gen, M, K) * 3.0)
    gamma = bf(1.0 + 0.1 * rand(gen, K))
    W = bf(rand(gen, N, K) * 0.05)
    Wg = bf(W * gamma.unsqueeze(0))
    xd = x.bfloat16().cuda()
    rows_p = (M + 15) // 16 * 16
    xp = torch.zeros(rows_p * K, dtype=torch.bfloat16, device="cuda")
    L.check(lib.seedmi_pack_activations_bf16(L.ptr(xd), K, L.ptr(xp), M, K, L.stream_ptr()), "pack_activations")
    torch.cuda.synchronize()
    assert torch.equal(_unpack_activations(xp, M, K).float(), x)
    Wd = Wg.bfloat16().cuda()
    Wp = torch.empty(lib.seedmi_pack_skinny_weights_bytes(N, K) // 2, dtype=torch.bfloat16, device="cuda")
    L.check(lib.seedmi_pack_skinny_weights(L.ptr(Wd), K, N, K, L.ptr(Wp), L.stream_ptr()), "pack_weights")
    rstd = torch.rsqrt((x.double() ** 2).mean(-1, keepdim=True) + eps).float()
    y = (x.double() @ Wg.double().t()).float() * rstd                     # fp32 accumulate, scale, then ONE rounding
    if epi == "none":
        C = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
        _skinny_norm(lib, form, ws, M, N, K, L.ptr(xp), L.ptr(Wp), eps, None, 0, L.EPI_NONE, L.ptr(C), N, 0, None, "skinny_norm")
        torch.cuda.synchronize()
        assert_close_bf16(C, r(y), f"skinny+rmsnorm M{M} N{N} K{K}", atol_ulps=1.5, frac=0.998)
        first = C.clone()
        C.zero_()
        _skinny_norm(lib, form, ws, M, N, K, L.ptr(xp), L.ptr(Wp), eps, None, 0, L.EPI_NONE, L.ptr(C), N, 0, None, "skinny_norm again")
        torch.cuda.synchronize()
        assert torch.equal(C, first), "a repeated launch gave different bits"
        ws_check()
    elif epi == "swiglu":
        C = torch.zeros(M, N // 2, dtype=torch.bfloat16, device="cuda")
        _skinny_norm(lib, form, ws, M, N, K, L.ptr(xp), L.ptr(Wp), eps, None, 0, L.EPI_SWIGLU, L.ptr(C), N // 2, 0, None, "skinny_norm swiglu")
        torch.cuda.synchronize()
        yh = r(y)
        gate, up = yh[:, 0::2], yh[:, 1::2]
        want = r(r(torch.nn.functional.silu(gate)) * up)
        assert_close_bf16(C, want, f"skinny+rmsnorm+swiglu M{M}", atol_ulps=2.0, frac=0.995)
        if (N % 64) == 0:                                                 # the fragment-major output the decode chain uses: same bits, permuted
            Cp = torch.zeros(rows_p * (N // 2), dtype=torch.bfloat16, device="cuda")
            _skinny_norm(lib, form, ws, M, N, K, L.ptr(xp), L.ptr(Wp), eps, None, 0, L.EPI_SWIGLU, L.ptr(Cp), N // 2, 1, None, "swiglu packed")
            torch.cuda.synchronize()
            assert torch.equal(_unpack_activations(Cp, M, N // 2), C.cpu())
        ws_check()
    else:
        # residual epilogue without the scaling (rms_eps = 0) and with the fragment-major second copy
        res = bf(rand(gen, M, N))
        Rd = res.bfloat16().cuda()
        C = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
        outp = torch.zeros(rows_p * N, dtype=torch.bfloat16, device="cuda")
        _skinny_norm(lib, form, ws, M, N, K, L.ptr(xp), L.ptr(Wp), 0.0, L.ptr(Rd), N, L.EPI_BIAS_RESIDUAL, L.ptr(C), N, 0, L.ptr(outp),
                     "skinny residual + packed copy")
        torch.cuda.synchronize()
        want = r(r((x.double() @ Wg.double().t()).float()) + res)
        assert_close_bf16(C, want, f"skinny residual M{M}", atol_ulps=1.5, frac=0.998)
        assert torch.equal(_unpack_activations(outp, M, N), C.cpu())     # the second copy is the same bits, permuted
        # in place (residual == output), as the decode chain runs o_proj / down_proj on the residual stream
        inplace = Rd.clone()
        _skinny_norm(lib, form, ws, M, N, K, L.ptr(xp), L.ptr(Wp), 0.0, L.ptr(inplace), N, L.EPI_BIAS_RESIDUAL, L.ptr(inplace), N, 0, L.ptr(outp),
                     "skinny residual in place")
        torch.cuda.synchronize()
        assert torch.equal(inplace, C), "in-place residual differs from the out-of-place launch"
        ws_check()
