"""CLIP preprocessing, same surface as the reference's models/transforms.py:4-21
(Resize -> [CenterCrop] -> ToTensor -> Normalize with the CLIP mean/std).

torchvision is used when it is installed (identical to the reference); otherwise an equivalent PIL + torch
pipeline runs (PIL bilinear resize is what torchvision applies to PIL inputs).  This is the step *before* the hot
path and stays on the host.
"""
import torch

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


class _Compose:
    def __init__(self, fns):
        self.transforms = fns

    def __call__(self, img):
        for f in self.transforms:
            img = f(img)
        return img


def _pil_resize(size, interpolation=2):
    from PIL import Image

    def f(img):
        if isinstance(size, int):                      # keep ratio: shorter side -> size
            w, h = img.size
            if w <= h:
                nw, nh = size, int(size * h / w)
            else:
                nw, nh = int(size * w / h), size
        else:
            nh, nw = size
        return img.resize((nw, nh), resample=interpolation)
    f.__name__ = "Resize"
    return f


def _center_crop(size):
    def f(img):
        w, h = img.size
        l, t = int(round((w - size) / 2.0)), int(round((h - size) / 2.0))
        return img.crop((l, t, l + size, t + size))
    return f


def _to_tensor(img):
    import numpy as np
    a = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy())
    if a.dim() == 2:
        a = a[:, :, None]
    return a.permute(2, 0, 1).float().div(255.0)


def _normalize(mean, std):
    m = torch.tensor(mean).view(-1, 1, 1)
    s = torch.tensor(std).view(-1, 1, 1)
    return lambda t: (t - m) / s


def get_transform(type='clip', keep_ratio=True, image_size=224):
    if type != 'clip':
        raise NotImplementedError
    try:
        from torchvision import transforms
        tf = []
        if keep_ratio:
            tf.extend([transforms.Resize(image_size), transforms.CenterCrop(image_size)])
        else:
            tf.append(transforms.Resize((image_size, image_size)))
        tf.extend([transforms.ToTensor(), transforms.Normalize(mean=CLIP_MEAN, std=CLIP_STD)])
        return transforms.Compose(tf)
    except ImportError:
        tf = []
        if keep_ratio:
            tf.extend([_pil_resize(image_size), _center_crop(image_size)])
        else:
            tf.append(_pil_resize((image_size, image_size)))
        tf.extend([_to_tensor, _normalize(CLIP_MEAN, CLIP_STD)])
        return _Compose(tf)
