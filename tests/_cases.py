"""Input cases shared by the CPU (oracle) and GPU (HIP path) tests."""
import torch


def nonfinite_vq_case(dtype):
    """z rows and codebook rows that make VectorQuantizer2's distance row non-finite: |z|^2 overflowing the 16-bit type (every distance +inf),
    NaN and inf in z, a codebook row holding NaN (its column of d is NaN for every z) and one holding inf."""
    gen = torch.Generator().manual_seed(13)
    big = 300.0 if dtype == torch.float16 else 2.0e19           # square overflows fp16 (65504) / bf16-fp32 (3.4e38)
    cb = torch.randn(512, 32, generator=gen) * 0.3
    z = cb[torch.randint(0, 512, (24,), generator=gen)] + torch.randn(24, 32, generator=gen) * 0.05
    z[1] = big                                                    # all distances +inf -> index 0
    z[2, 5] = float("nan")                                        # all distances NaN -> index 0
    z[3, 7] = float("inf")
    z[4] = -big
    z[5, 0] = big
    cb2 = cb.clone()
    cb2[77, 3] = float("nan")                                     # d[:, 77] is NaN: the first NaN is torch's minimum
    cb2[300, 9] = float("nan")
    cb2[40, 1] = float("inf")
    return z.to(dtype).float(), cb.to(dtype).float(), cb2.to(dtype).float()
