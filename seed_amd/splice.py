"""Image-token splice either side of the LLM, as id arithmetic on device.

The reference round-trips ids through strings: ``'<img_{:05d}>'.format(id)`` -> SentencePiece -> ids
(scripts/seed_llama_inference_8B.py:21-23,99-100) and parses generated spans back with tokenizer lookups (:41-64).
The added vocabulary is contiguous (32000 + code, then <img>, </img>), so the same sequences are built directly:

    BOS "USER: " <img> (32000 + ids[0..31]) </img> question "\nASSISTANT:"
"""
from typing import List, Optional, Tuple

import torch

IMAGE_ID_SHIFT = 32000          # scripts/seed_llama_inference_8B.py:23
NUM_IMG_TOKENS = 32
NUM_IMG_CODES = 8192


def image_span(image_ids: torch.Tensor, boi_id: int, eoi_id: int, shift: int = IMAGE_ID_SHIFT) -> torch.Tensor:
    """[B,32] codebook ids -> [B,34] LLM token ids  <img> img_XXXXX x32 </img>  (same device, no host sync)."""
    assert image_ids.dim() == 2
    B = image_ids.shape[0]
    boi = torch.full((B, 1), boi_id, dtype=torch.int64, device=image_ids.device)
    eoi = torch.full((B, 1), eoi_id, dtype=torch.int64, device=image_ids.device)
    return torch.cat((boi, image_ids.to(torch.int64) + shift, eoi), dim=1)


def splice_prompt(pieces: List[torch.Tensor]) -> torch.Tensor:
    """Concatenate per-row pieces ([B,T_i] text ids and [B,34] image spans) into the prompt [B, sum T_i]; all rows
    have equal length (the reference's eval path is only well defined for unpadded equal-length batches)."""
    B = pieces[0].shape[0]
    assert all(p.dim() == 2 and p.shape[0] == B for p in pieces)
    return torch.cat([p.to(torch.int64) for p in pieces], dim=1)


def parse_generated(generate_ids: torch.Tensor, boi_id: int, eoi_id: int, shift: int = IMAGE_ID_SHIFT
                    ) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """decode_image_text of the reference scripts (:41-64) on ids: returns (text_ids, image_ids [1,32] or None)."""
    boi = torch.where(generate_ids == boi_id)[0]
    eoi = torch.where(generate_ids == eoi_id)[0]
    if len(boi) == 0 and len(eoi) == 0:
        return generate_ids, None
    b, e = int(boi[0]), int(eoi[0])
    return generate_ids[:b], (generate_ids[b + 1:e] - shift).reshape(1, -1)


def sample_top_p(logits: torch.Tensor, top_p: float = 0.5, temperature: float = 1.0,
                 generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """Nucleus sampling on device (the scripts' generation_config: do_sample, top_p 0.5, temperature 1.0) without
    leaving the GPU: returns [B,1] token ids.  Same rule as HF's TopPLogitsWarper (keep the smallest prefix of the
    sorted distribution whose mass reaches top_p, always at least one token)."""
    probs = torch.softmax(logits.float() / temperature, dim=-1)
    sp, si = torch.sort(probs, dim=-1, descending=True)
    csum = sp.cumsum(-1)
    drop = (csum - sp) >= top_p                      # tokens whose preceding mass already reached top_p
    sp = sp.masked_fill(drop, 0.0)
    sp = sp / sp.sum(-1, keepdim=True)
    pick = torch.multinomial(sp, 1, generator=generator)
    return si.gather(-1, pick)
