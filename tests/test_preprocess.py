"""Image pre-processing in front of the tokenizer.

The arithmetic belongs to Pillow / torchvision (not under /root/reference); the oracle restates Pillow's 8-bit resampler in
numpy and is pinned here against the real Pillow (installed in the image, here and on the GPU box).  The HIP path
(seedmi_preprocess_image_u8, through the C ABI) must be BIT-exact in its uint8 stage and produce identical fp32 values.
"""
import os

import numpy as np
import pytest
import torch
from PIL import Image

from oracle import preprocess_oracle as P

SIZES = [(37, 53), (224, 224), (480, 640), (1000, 333), (225, 223), (13, 700), (1, 1), (2048, 1536)]


def _img(h, w, seed):
    rs = np.random.RandomState(seed)
    if seed % 2:                                       # smooth content with saturated patches (exercises clip8)
        y, x = np.mgrid[0:h, 0:w]
        base = np.stack([(x * 255 // max(w - 1, 1)), (y * 255 // max(h - 1, 1)), ((x + y) % 256)], -1).astype(np.int64)
        base[: h // 3, : w // 3] = 255
        base[h // 2:, w // 2:] = 0
        return np.clip(base + rs.randint(-20, 20, base.shape), 0, 255).astype(np.uint8)
    return rs.randint(0, 256, (h, w, 3)).astype(np.uint8)


def _torch_pipeline(img_u8, size, filt, keep_ratio):
    """The reference processors rebuilt from PIL + torch (what torchvision does for PIL inputs)."""
    im = Image.fromarray(img_u8)
    h, w = img_u8.shape[:2]
    rh, rw, top, left = P.resize_geometry(h, w, size, keep_ratio)
    im = im.resize((rw, rh), filt).crop((left, top, left + size, top + size))
    u8 = np.asarray(im, dtype=np.uint8)
    t = torch.from_numpy(u8.copy()).permute(2, 0, 1).float().div(255.0)
    m = torch.tensor(P.CLIP_MEAN).view(-1, 1, 1)
    s = torch.tensor(P.CLIP_STD).view(-1, 1, 1)
    return (t - m) / s, u8


@pytest.mark.parametrize("hw", SIZES[:6])
@pytest.mark.parametrize("filt", [2, 3])
def test_oracle_resampler_is_bit_exact_against_pillow(hw, filt):
    h, w = hw
    img = _img(h, w, h + w)
    for oh, ow in ((224, 224), (56, 40), (h, max(1, w // 2))):
        ref = np.asarray(Image.fromarray(img).resize((ow, oh), filt))
        assert np.array_equal(P.pil_resize_u8(img, oh, ow, filt), ref), (hw, filt, oh, ow)


@pytest.mark.parametrize("keep_ratio,filt", [(False, 3), (True, 2), (False, 2)])
def test_oracle_pipeline_matches_torch_pipeline(keep_ratio, filt):
    img = _img(300, 411, 7)
    want, want_u8 = _torch_pipeline(img, 224, filt, keep_ratio)
    got, got_u8 = P.preprocess(img, 224, filt, keep_ratio)
    assert np.array_equal(got_u8, want_u8)
    assert np.array_equal(got, want.numpy())           # same fp32 operations in the same order



CAT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cat.jpg")     # the reference's own fixture: images/cat.jpg


def test_reference_fixture_cat_jpg_through_the_scripts_transform():
    """scripts/seed_tokenizer_inference.py:12,22-28: ``images/cat.jpg`` -> ``Image.open(...).convert('RGB')`` ->
    ``hydra.utils.instantiate(configs/transform/clip_transform.yaml)`` (= models.transforms.get_transform: Resize((224,224)) bilinear,
    ToTensor, Normalize) -> tensor.  The reference records no expected ids for it, so the pin is: our
    models/transforms.py and the ImageTokenizer.processor (bicubic, seed_llama_tokenizer.py:50-56) applied to the fixture EQUAL the
    preprocessing oracle (itself bit-exact against Pillow, above) on the same decoded pixels."""
    import yaml
    from models.transforms import get_transform
    from models.seed_llama_tokenizer import _make_processor
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yaml.safe_load(open(os.path.join(root, "configs", "transform", "clip_transform.yaml")))
    assert cfg.pop("_target_") == "models.transforms.get_transform" and cfg == {"type": "clip", "image_size": 224, "keep_ratio": False}
    image = Image.open(CAT).convert("RGB")
    assert image.size == (690, 685)
    px = np.asarray(image, dtype=np.uint8)
    t = get_transform(**cfg)(image)
    assert tuple(t.shape) == (3, 224, 224) and t.dtype == torch.float32
    want, _ = P.preprocess(px, 224, 2, False)
    assert np.array_equal(t.numpy(), want)
    t3 = _make_processor(224)(image)                                                      # the image_path / image_pil branch
    want3, _ = P.preprocess(px, 224, 3, False)
    assert np.array_equal(t3.numpy(), want3)
    assert not np.array_equal(want, want3)                                                # (the two routes really are different filters)


@pytest.mark.gpu
@pytest.mark.parametrize("hw", SIZES)
@pytest.mark.parametrize("keep_ratio,filt", [(False, 3), (True, 2), (False, 2), (True, 3)])
def test_device_preprocess_is_bit_exact(hw, keep_ratio, filt):
    from seed_amd.preprocess import DevicePreprocessor
    h, w = hw
    if keep_ratio and min(h, w) * 8 < max(h, w):
        pytest.skip("extreme aspect ratio: the long side would exceed the reference's own practical range")
    img = _img(h, w, 3 * h + w)
    pre = DevicePreprocessor(224, interpolation=filt, keep_ratio=keep_ratio)
    out, u8 = pre(img, tap_u8=True)
    torch.cuda.synchronize()
    want, want_u8 = _torch_pipeline(img, 224, filt, keep_ratio)
    assert np.array_equal(u8.cpu().numpy(), want_u8)                       # Pillow's resampler, bit for bit
    assert torch.equal(out.cpu(), want)                                    # identical fp32 ToTensor / Normalize
    o_f, o_u8 = P.preprocess(img, 224, filt, keep_ratio)
    assert np.array_equal(o_u8, want_u8) and np.array_equal(o_f, want.numpy())


@pytest.mark.gpu
def test_device_preprocess_batch_pil_input_and_bf16():
    from seed_amd.preprocess import DevicePreprocessor
    imgs = [_img(120 + 31 * i, 90 + 17 * i, i) for i in range(5)]
    pre = DevicePreprocessor(224, interpolation=3, keep_ratio=False)
    batch = pre.batch([Image.fromarray(a) for a in imgs])
    assert tuple(batch.shape) == (5, 3, 224, 224) and batch.dtype == torch.float32
    for i, a in enumerate(imgs):
        assert torch.equal(batch[i].cpu(), _torch_pipeline(a, 224, 3, False)[0])
    half = DevicePreprocessor(224, interpolation=3, out_dtype=torch.bfloat16)(imgs[0])
    assert torch.equal(half.cpu(), _torch_pipeline(imgs[0], 224, 3, False)[0].bfloat16())   # one rounding of the same fp32
    h16 = DevicePreprocessor(224, interpolation=3, out_dtype=torch.float16)(imgs[0])        # the fp16 build: img.half() of the reference (:86-87)
    assert h16.dtype == torch.float16 and torch.equal(h16.cpu(), _torch_pipeline(imgs[0], 224, 3, False)[0].half())
    with pytest.raises(ValueError):
        pre(np.zeros((4, 4), dtype=np.uint8))
