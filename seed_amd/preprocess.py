"""Device-side image pre-processing: the reference's `Resize -> [CenterCrop] -> ToTensor -> Normalize` on the GPU.

Host mirror of the two processors the reference builds (same geometry rules, same arithmetic, bit-exact uint8 stage):

* ``ImageTokenizer.processor``  models/seed_llama_tokenizer.py:50-56   Resize((S,S), bicubic) / ToTensor / Normalize(CLIP)
* ``get_transform('clip', keep_ratio)``  models/transforms.py:8-21       Resize(S)+CenterCrop(S) or Resize((S,S)), bilinear

JPEG decoding stays on the host (PIL); the decoded uint8 pixels are uploaded once and everything after that — the
antialiased resize, the crop, /255 and the CLIP normalisation — runs in seedmi_preprocess_image_u8, writing straight into
a row of the [B,3,S,S] batch tensor that ``encode_image`` consumes.  No CPU fallback: raises without the HIP library.
"""
import ctypes as C

import numpy as np
import torch

from . import lib as L

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
BILINEAR, BICUBIC = 2, 3


def resize_geometry(h: int, w: int, size: int, keep_ratio: bool):
    """(resize_h, resize_w, crop_top, crop_left).  keep_ratio: torchvision Resize(int) (shorter side -> size, the other
    int(size * long / short)) then CenterCrop (offsets int(round((dim - size) / 2))) — models/transforms.py:11-14."""
    if not keep_ratio:
        return size, size, 0, 0
    if w <= h:
        rw, rh = size, int(size * h / w)
    else:
        rw, rh = int(size * w / h), size
    return rh, rw, int(round((rh - size) / 2.0)), int(round((rw - size) / 2.0))


class DevicePreprocessor:
    def __init__(self, image_size: int = 224, interpolation: int = BICUBIC, keep_ratio: bool = False, mean=CLIP_MEAN,
                 std=CLIP_STD, device="cuda", out_dtype=torch.float32):
        if interpolation not in (BILINEAR, BICUBIC):
            raise ValueError("interpolation must be 2 (PIL bilinear) or 3 (PIL bicubic)")
        if out_dtype not in (torch.float32, torch.bfloat16, torch.float16):
            raise ValueError("out_dtype must be float32, bfloat16 or float16")
        # the 16-bit output is the loaded build's element: float16 comes from libseedmi_f16.so (fp32 output is the same from either build)
        self.lib = L.load(torch.float16 if out_dtype == torch.float16 else None)
        self.size, self.filter, self.keep_ratio = image_size, interpolation, keep_ratio
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise L.SeedmiError("DevicePreprocessor needs a HIP device (cuda:N); the host path is models/transforms.py")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.out_dtype = out_dtype
        self.lib_half = torch.float16 if out_dtype == torch.float16 else torch.bfloat16      # the loaded build's 16-bit element: always accepted for `out`
        self._mean = (C.c_float * 3)(*mean)
        self._std = (C.c_float * 3)(*std)
        self._ws = None

    def _as_u8(self, image) -> np.ndarray:
        if hasattr(image, "convert"):                    # PIL.Image: scripts call Image.open(path).convert('RGB')
            image = np.asarray(image.convert("RGB"), dtype=np.uint8)
        a = np.ascontiguousarray(image)
        if not a.flags.writeable:                        # torch.from_numpy wants a writable buffer (PIL exposes read-only)
            a = a.copy()
        if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
            raise ValueError(f"expected an RGB uint8 [H,W,3] image, got {a.dtype} {a.shape}")
        return a

    def __call__(self, image, out: torch.Tensor = None, tap_u8: bool = False):
        """image: PIL.Image or uint8 [H,W,3] array.  Returns float [3,S,S] on the device (or fills ``out``);
        with tap_u8 also the resized/cropped uint8 [S,S,3] tensor (parity checks)."""
        a = self._as_u8(image)
        h, w = a.shape[:2]
        S = self.size
        rh, rw, top, left = resize_geometry(h, w, S, self.keep_ratio)
        src = torch.from_numpy(a).to(self.device, non_blocking=False)
        if out is None:
            out = torch.empty(3, S, S, dtype=self.out_dtype, device=self.device)
        elif tuple(out.shape) != (3, S, S) or out.device != self.device or not out.is_contiguous() or \
                out.dtype not in (torch.float32, self.lib_half):
            raise ValueError(f"out must be a contiguous [3,S,S] float32 or {self.lib_half} tensor on this device")
        u8 = torch.empty(S, S, 3, dtype=torch.uint8, device=self.device) if tap_u8 else None
        need = self.lib.seedmi_preprocess_workspace_bytes(h, w, rh, rw, self.filter)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            rc = self.lib.seedmi_preprocess_image_u8(L.ptr(src), h, w, 3 * w, rh, rw, self.filter, top, left, S, S, self._mean,
                                                     self._std, L.ptr(out), 1 if out.dtype == torch.float32 else 0, L.ptr(u8),
                                                     L.ptr(self._ws), self._ws.numel(), L.stream_ptr())
        L.check(rc, "seedmi_preprocess_image_u8", self.lib)
        return (out, u8) if tap_u8 else out

    def batch(self, images) -> torch.Tensor:
        """List of images -> [B,3,S,S] on the device, each written in place by its own kernel pair."""
        out = torch.empty(len(images), 3, self.size, self.size, dtype=self.out_dtype, device=self.device)
        for i, im in enumerate(images):
            self(im, out=out[i])
        return out
