#!/usr/bin/env python
"""Data-parallel dataset pre-tokenizer on the MI355X path (the direct consumer of the tokenize hot path).

Re-creates MultiModalLLM/src/tools/extract_image_ids_to_torchdata_parallel.py:72-127 of the reference: one process per
GPU, every rank tokenizes its own shard of the image list at a large batch size and writes its own webdataset-style
``part-%04d/%07d.tar`` files whose members are pickled ``{'image_ids': [32 ints], 'text': str, 'metadata': dict}`` — the
on-disk format consumed by the reference's training pipes (src/data/torchdata_train.py:95-112).  Like the reference,
no data collective is needed (ranks only rendezvous); launch with

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \
        -m seed_amd.tools.extract_image_ids --images /data/imgs --save_dir /data/ids --batch_size 1024

Inputs: a directory of image files (optionally ``<name>.txt`` captions next to them), or ``synthetic:N`` for N random
224x224 tensors.  JPEG decode runs on the host; the CLIP resize/normalise (models/transforms.py) runs on the host too or, with
--gpu-preprocess, in seedmi_preprocess_image_u8 (bit-exact with the PIL path) — the step before the hot path.
"""
import argparse
import io
import os
import pickle
import tarfile
import time
from typing import Callable, List, Optional

import torch

from seed_amd.dist import shard_range

IMG_EXT = (".jpg", ".jpeg", ".png", ".bmp", ".webp")


class TarShardWriter:
    """Minimal webdataset ShardWriter: ``<dir>/%07d.tar`` with at most ``maxcount`` samples, member ``<key>.pkl``."""

    def __init__(self, directory: str, maxcount: int = 10000):
        os.makedirs(directory, exist_ok=True)
        self.directory, self.maxcount = directory, maxcount
        self.shard, self.count, self.tar = -1, 0, None
        self.total = 0

    def _next(self):
        self.close()
        self.shard += 1
        self.count = 0
        self.tar = tarfile.open(os.path.join(self.directory, f"{self.shard:07d}.tar"), "w")

    def write(self, key: str, sample: dict):
        if self.tar is None or self.count >= self.maxcount:
            self._next()
        data = pickle.dumps(sample)
        info = tarfile.TarInfo(f"{key}.pkl")
        info.size = len(data)
        info.mtime = int(time.time())
        self.tar.addfile(info, io.BytesIO(data))
        self.count += 1
        self.total += 1

    def close(self):
        if self.tar is not None:
            self.tar.close()
            self.tar = None


def list_images(path: str) -> List[str]:
    files = []
    for root, _, names in os.walk(path):
        for n in sorted(names):
            if n.lower().endswith(IMG_EXT):
                files.append(os.path.join(root, n))
    return sorted(files)


def run(encode_fn: Callable[[torch.Tensor], torch.Tensor], items: List, load_fn: Callable, save_dir: str, rank: int,
        world: int, batch_size: int, device: str, maxcount: int = 10000, log: Optional[Callable] = print) -> int:
    """Tokenize this rank's contiguous shard of ``items``; returns the number of samples written."""
    b, e = shard_range(len(items), rank, world)
    writer = TarShardWriter(os.path.join(save_dir, f"part-{rank:04d}"), maxcount)
    t0 = time.time()
    for s in range(b, e, batch_size):
        chunk = items[s:min(s + batch_size, e)]
        tensors, texts, metas = zip(*(load_fn(it) for it in chunk))
        # host-preprocessed samples are CPU tensors; with --gpu-preprocess each is already a [3,224,224] device tensor
        batch = torch.stack(tensors).to(device, non_blocking=True)
        ids = encode_fn(batch)                                    # [B, 32] int64 on device
        ids = ids.view(len(chunk), -1).cpu().tolist()             # the reference's `.view(-1).cpu().tolist()` per sample
        for i, (row, text, meta) in enumerate(zip(ids, texts, metas)):
            writer.write(f"{s + i:09d}", {"image_ids": row, "text": text, "metadata": meta})
        if log:
            log(f"[rank {rank}] {min(s + batch_size, e) - b}/{e - b} images, {(min(s + batch_size, e) - b) / (time.time() - t0):.1f} img/s")
    writer.close()
    return writer.total


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", required=True, help="image directory or synthetic:N")
    ap.add_argument("--save_dir", required=True)
    ap.add_argument("--batch_size", type=int, default=1024)
    ap.add_argument("--weights", default=None, help="seed_quantizer.pt (default: seeded synthetic weights)")
    ap.add_argument("--gpu-preprocess", action="store_true",
                    help="resize/normalise on the device (seedmi_preprocess_image_u8, bit-exact with the PIL path); "
                         "only the JPEG decode stays on the host")
    ap.add_argument("--dtype", choices=("bf16", "fp16"), default="bf16",
                    help="compute type of the tokenizer: bf16 (libseedmi.so, BASELINE.json) or fp16 (libseedmi_f16.so: what the reference tool gets from its "
                         "tokenizer yaml's `fp16: True`, extract_image_ids_to_torchdata_parallel.py:92-93)")
    args = ap.parse_args(argv)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))    # rendezvous only, like the reference (:73)
    from seed_amd import config as C
    from seed_amd.tokenizer_engine import TokenizerEngine
    from seed_amd.weights import make_tokenizer_state_dict
    sd = torch.load(args.weights, map_location="cpu") if args.weights else make_tokenizer_state_dict(C.SEED2, seed=0, device="cuda")
    eng = TokenizerEngine(sd, C.SEED2, device=f"cuda:{local}", dtype=torch.float16 if args.dtype == "fp16" else torch.bfloat16)
    if args.images.startswith("synthetic:"):
        n = int(args.images.split(":")[1])
        items = list(range(n))

        def load(i):
            g = torch.Generator().manual_seed(i)
            return torch.randn(3, 224, 224, generator=g), "", {"index": i}
    else:
        from PIL import Image
        from models.transforms import get_transform
        tf = get_transform(type="clip", keep_ratio=False, image_size=224)
        if args.gpu_preprocess:
            from seed_amd.preprocess import DevicePreprocessor, BILINEAR
            tf = DevicePreprocessor(224, interpolation=BILINEAR, keep_ratio=False, device=f"cuda:{local}")   # same ops as tf
        items = list_images(args.images)

        def load(path):
            cap = os.path.splitext(path)[0] + ".txt"
            text = open(cap).read().strip() if os.path.exists(cap) else ""
            return tf(Image.open(path).convert("RGB")), text, {"path": path}
    n = run(eng.encode, items, load, args.save_dir, rank, world, args.batch_size, f"cuda:{local}")
    print(f"[rank {rank}] wrote {n} samples")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
