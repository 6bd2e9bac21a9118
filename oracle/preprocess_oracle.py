"""CPU restatement of the image pre-processing in front of the tokenizer (the parity ORACLE for seedmi_preprocess_image_u8).

TEST INFRASTRUCTURE (see oracle/seed_oracle.py's header for the import rules).

The reference's pre-processing is `transforms.Resize -> [CenterCrop] -> ToTensor -> Normalize`
(models/seed_llama_tokenizer.py:50-56 bicubic, models/transforms.py:8-21 bilinear + centre crop); the arithmetic lives in
third-party code that is NOT under /root/reference:

* Pillow (torchvision's Resize on a PIL image is `img.resize(size[::-1], interpolation)`) — src/libImaging/Resample.c,
  ImagingResample for 8-bit images: `precompute_coeffs` (double precision, support scaled by max(1, in/out) = antialias,
  per-output-pixel normalisation), `normalize_coeffs_8bpc` (22-bit fixed point, round half away from zero),
  `ImagingResampleHorizontal_8bpc` then `ImagingResampleVertical_8bpc` (accumulator seeded with 1 << 21, `clip8` = shift and
  clamp), uint8 intermediate between the passes.  Restated below in numpy, operation for operation.
* torchvision ToTensor (`uint8 -> float32 / 255`) and Normalize (`(x - mean) / std`, fp32).

Pinned against the real Pillow in tests/test_preprocess.py (Pillow is part of the image, here and on the GPU box).
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _bilinear(x):
    x = abs(x)
    return 1.0 - x if x < 1.0 else 0.0


def _bicubic(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size: int, out_size: int, filt: int):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the full-image box.  Returns (bounds [out,2], kk [out,ksize])."""
    f, fsupport = (_bicubic, 2.0) if filt == 3 else (_bilinear, 1.0)
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = fsupport * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), dtype=np.int64)
    bounds = np.zeros((out_size, 2), dtype=np.int64)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        ss = 1.0 / filterscale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [f((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            v = v * (1 << PRECISION_BITS)
            kk[xx, x] = int(-0.5 + v) if v < 0 else int(0.5 + v)
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img: np.ndarray, bounds, kk, axis: int) -> np.ndarray:
    """One separable pass over `axis` of a uint8 [H,W,3] image."""
    src = img.astype(np.int64)
    out_size = bounds.shape[0]
    shape = list(img.shape)
    shape[axis] = out_size
    out = np.empty(shape, dtype=np.uint8)
    for i in range(out_size):
        lo, n = int(bounds[i, 0]), int(bounds[i, 1])
        k = kk[i, :n]
        if axis == 1:
            acc = (src[:, lo:lo + n, :] * k[None, :, None]).sum(axis=1)
        else:
            acc = (src[lo:lo + n, :, :] * k[:, None, None]).sum(axis=0)
        acc = (acc + (1 << (PRECISION_BITS - 1))) >> PRECISION_BITS
        acc = np.clip(acc, 0, 255).astype(np.uint8)
        if axis == 1:
            out[:, i, :] = acc
        else:
            out[i, :, :] = acc
    return out


def pil_resize_u8(img: np.ndarray, out_h: int, out_w: int, filt: int) -> np.ndarray:
    """Image.resize((out_w, out_h), filt) for an RGB uint8 array: horizontal pass first, then vertical (ImagingResample)."""
    assert img.dtype == np.uint8 and img.ndim == 3 and img.shape[2] == 3
    h, w = img.shape[:2]
    tmp = img
    if w != out_w:
        tmp = _pass(tmp, *precompute_coeffs(w, out_w, filt), axis=1)
    if h != out_h:
        tmp = _pass(tmp, *precompute_coeffs(h, out_h, filt), axis=0)
    return tmp


def resize_geometry(h: int, w: int, size: int, keep_ratio: bool):
    """(resize_h, resize_w, crop_top, crop_left) of transforms.Resize(size)+CenterCrop(size) (keep_ratio, models/transforms.py:
    11-14: shorter side -> size, long side int(size*long/short), crop offsets int(round((dim-size)/2))) or of
    transforms.Resize((size,size)) (:16)."""
    if not keep_ratio:
        return size, size, 0, 0
    if w <= h:
        rw, rh = size, int(size * h / w)
    else:
        rw, rh = int(size * w / h), size
    return rh, rw, int(round((rh - size) / 2.0)), int(round((rw - size) / 2.0))


def preprocess(img: np.ndarray, size: int = 224, filt: int = 3, keep_ratio: bool = False, mean=CLIP_MEAN, std=CLIP_STD):
    """uint8 RGB [H,W,3] -> (float32 [3,size,size], uint8 [size,size,3])."""
    h, w = img.shape[:2]
    rh, rw, top, left = resize_geometry(h, w, size, keep_ratio)
    u8 = pil_resize_u8(img, rh, rw, filt)[top:top + size, left:left + size]
    x = u8.astype(np.float32) / np.float32(255.0)                          # ToTensor
    m = np.asarray(mean, dtype=np.float32)
    s = np.asarray(std, dtype=np.float32)
    x = (x - m) / s                                                        # Normalize (fp32)
    return np.ascontiguousarray(x.transpose(2, 0, 1)), u8
