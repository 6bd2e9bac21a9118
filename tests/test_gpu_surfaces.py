"""The two named drop-in surfaces on a real MI355X with the HIP engines underneath (VERDICT r1 "what's missing" 1 and 3):

* the tokenizer is built the way scripts/seed_tokenizer_inference.py:18-20 / seed_llama_inference_8B.py:69-71 build it -
  ``hydra.utils.instantiate(OmegaConf.load('configs/tokenizer/seed_llama_tokenizer_hf.yaml'), device=device, load_diffusion=True)``
  (hydra/omegaconf are not installed: ``_instantiate`` below is the ten-line test-only stand-in) - and
  ``tokenizer.encode_image(image_torch=x)`` is compared with the engine and the oracle;
* the LLM is built by ``configs/llm/seed_llama_8b.yaml``'s target (``models.model_tools.get_pretrained_llama_causal_model``) from a
  saved checkpoint directory, moved with ``.eval().to(device)`` and driven through ``model.generate(...)`` exactly as
  scripts/seed_llama_inference_8B.py:28-39 does (greedy, the scripts' top-p sampling config, 2 beams -> ``_reorder_cache``).
"""
import importlib
import json
import os
import warnings

import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu

from oracle import seed_oracle as O  # noqa: E402
from seed_amd import config as C  # noqa: E402
from seed_amd.weights import make_llama_state_dict, make_tokenizer_state_dict, calibrate_codebook  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _instantiate(cfg: dict, **overrides):
    """hydra.utils.instantiate for flat configs: import ``_target_`` and call it with the config's keys + overrides."""
    kw = {k: v for k, v in cfg.items() if k != "_target_"}
    kw.update(overrides)
    mod, _, attr = cfg["_target_"].rpartition(".")
    try:
        fn = getattr(importlib.import_module(mod), attr)
    except ModuleNotFoundError:                       # Class.from_pretrained style targets
        m2, _, cls = mod.rpartition(".")
        fn = getattr(getattr(importlib.import_module(m2), cls), attr)
    return fn(**kw)


@pytest.fixture(scope="module")
def tokenizer_dir(tmp_path_factory):
    """A local stand-in for the 'AILab-CVC/seed-tokenizer-2' hub repo: a tiny text tokenizer + seed_quantizer.pt (MID config)."""
    from tokenizers import Tokenizer, models, pre_tokenizers, trainers
    d = str(tmp_path_factory.mktemp("seed_tokenizer"))
    tok = Tokenizer(models.BPE(unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.Whitespace()
    tr = trainers.BpeTrainer(vocab_size=300, special_tokens=["<unk>", "<s>", "</s>", "<img>", "</img>"])
    tok.train_from_iterator(["USER: what is this animal ? ASSISTANT: a cat on the green grass"] * 50, tr)
    tok.save(os.path.join(d, "tokenizer.json"))
    json.dump({"tokenizer_class": "LlamaTokenizer", "bos_token": "<s>", "eos_token": "</s>", "unk_token": "<unk>"},
              open(os.path.join(d, "tokenizer_config.json"), "w"))
    cfg = C.MID
    sd = make_tokenizer_state_dict(cfg, seed=0, ln_jitter=0.05)
    img = torch.randn(8, 3, cfg.img_size, cfg.img_size, generator=torch.Generator().manual_seed(5))
    t = {}
    O.get_codebook_indices(sd, img, cfg, "fp32", t)
    sd["quantize.embedding.weight"] = calibrate_codebook(t["z"], cfg.n_embed, seed=7)
    torch.save(sd, os.path.join(d, "seed_quantizer.pt"))
    return d, sd, cfg


def test_scripts_tokenizer_construction_and_encode_image(tokenizer_dir):
    d, sd, cfg = tokenizer_dir
    from seed_amd.tokenizer_engine import TokenizerEngine
    tcfg = yaml.safe_load(open(os.path.join(ROOT, "configs/tokenizer/seed_llama_tokenizer_hf.yaml")))
    assert tcfg["_target_"] == "models.seed_llama_tokenizer.SeedLlamaTokenizer.from_pretrained" and tcfg["fp16"] is True
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tokenizer = _instantiate(tcfg, device="cuda", load_diffusion=True,                       # the scripts' call
                                 pretrained_model_name_or_path=d,                               # (offline stand-ins for the hub ids)
                                 encoder_url=os.path.join(d, "seed_quantizer.pt"),
                                 image_tokenizer_kwargs={"cfg": cfg})
    assert tokenizer.load_diffusion and tokenizer.num_image_tokens == 8192 and tokenizer.bos_token == "<s>"
    ids_txt = tokenizer("USER: a cat", add_special_tokens=False, return_tensors="pt").input_ids
    assert ids_txt.dtype == torch.int64 and ids_txt.shape[0] == 1
    x = torch.randn(6, 3, cfg.img_size, cfg.img_size, generator=torch.Generator().manual_seed(11))
    ids = tokenizer.encode_image(image_torch=x.cuda())                                           # seed_llama_tokenizer.py:185-202
    assert ids.dtype == torch.int64 and tuple(ids.shape) == (6, cfg.n_query) and ids.device.type == "cuda"
    assert int(ids.min()) >= 0 and int(ids.max()) < cfg.n_embed
    one = tokenizer.encode_image(image_torch=x[2].cuda())                                        # 3-D input is unsqueezed (:81-82)
    assert torch.equal(one, ids[2:3])
    # the yaml says `fp16: True` (the reference's shipped setting): ImageTokenizer calls model.half(), which since round 5 selects the fp16
    # build of the library - the surface must equal the fp16 engine on the same input, not the bf16 one
    assert tokenizer.image_tokenizer.model.engine.dtype == torch.float16
    eng = TokenizerEngine(sd, cfg, device="cuda", dtype=torch.float16)
    taps = {}
    assert torch.equal(eng.encode(x.cuda(), taps), ids), "encode_image != TokenizerEngine.encode (fp16) on the same input"
    # oracle contract: ids bit-exact on the engine's own z; z within fp16 distance of the oracle
    z = taps["z"].float().cpu()
    same_z = O.vq_argmin(z, sd["quantize.embedding.weight"], O.Prec("fp16")).reshape(ids.shape)
    assert torch.equal(ids.cpu(), same_z)
    t16 = {}
    ids16 = O.get_codebook_indices(sd, x, cfg, "fp16", t16)
    rel = ((z - t16["z"]).norm() / t16["z"].norm()).item()
    agree = (ids.cpu() == ids16).float().mean().item()
    print(f"[surface tokenizer] z rel err vs fp16 oracle {rel:.2e}; id agreement {agree:.3f}")
    assert rel < 3e-3 and agree > 0.95
    # .bfloat16() on the same object switches the encode path back to the bf16 build (BASELINE.json's dtype)
    tokenizer.image_tokenizer.model.bfloat16()
    ids_b = tokenizer.encode_image(image_torch=x.cuda())
    assert torch.equal(ids_b, TokenizerEngine(sd, cfg, device="cuda").encode(x.cuda()))
    tokenizer.image_tokenizer.model.half()
    with pytest.raises(AssertionError):
        tokenizer.encode_image()                                                                 # exactly one input (:192)
    with pytest.raises(RuntimeError, match="StableUnCLIP|de-tokenizer"):
        tokenizer.decode_image(ids[:1])                                                          # the only call that needs diffusers
    # .to('cuda') names the device the engine already sits on: the packed engine must be kept
    before = tokenizer.image_tokenizer.model._engine
    tokenizer.to("cuda")
    assert tokenizer.image_tokenizer.model._engine is before


def test_reference_fixture_cat_jpg_both_preprocessing_routes(tokenizer_dir):
    """scripts/seed_tokenizer_inference.py:12,20-29 on the reference's own fixture (tests/golden/cat.jpg = images/cat.jpg): the tokenizer
    built by the scripts' hydra call, ``transform(image).to(device)`` -> ``encode_image(image_torch=...)``.  The PIL route
    (models/transforms.py on the host) and the device route (``DevicePreprocessor``: decoded uint8 pixels uploaded, resize + normalise in
    seedmi_preprocess_image_u8) must give EQUAL tensors and therefore equal ids, for the scripts' bilinear transform and for the bicubic
    processor behind ``encode_image(image_path=...)`` / ``image_pil=`` (seed_llama_tokenizer.py:50-56,185-202)."""
    from PIL import Image
    from seed_amd.preprocess import DevicePreprocessor, BILINEAR, BICUBIC
    d, sd, cfg = tokenizer_dir
    assert cfg.img_size == 224
    cat = os.path.join(ROOT, "tests", "golden", "cat.jpg")
    tcfg = yaml.safe_load(open(os.path.join(ROOT, "configs/tokenizer/seed_llama_tokenizer_hf.yaml")))
    xcfg = yaml.safe_load(open(os.path.join(ROOT, "configs/transform/clip_transform.yaml")))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tokenizer = _instantiate(tcfg, device="cuda", load_diffusion=True, pretrained_model_name_or_path=d,
                                 encoder_url=os.path.join(d, "seed_quantizer.pt"), image_tokenizer_kwargs={"cfg": cfg})
    transform = _instantiate(xcfg)
    image = Image.open(cat).convert("RGB")                                                   # script line 26
    image_tensor = transform(image).to("cuda")                                               # line 28
    ids = tokenizer.encode_image(image_torch=image_tensor)                                   # line 29
    assert ids.dtype == torch.int64 and tuple(ids.shape) == (1, cfg.n_query) and 0 <= int(ids.min()) and int(ids.max()) < cfg.n_embed
    dev = DevicePreprocessor(224, BILINEAR, keep_ratio=False)(image)
    assert torch.equal(dev.cpu(), image_tensor.cpu()), "device preprocessing != models/transforms.py on cat.jpg"
    assert torch.equal(tokenizer.encode_image(image_torch=dev), ids)
    # the path / PIL branches run the bicubic processor (seed_llama_tokenizer.py:194-200)
    by_path = tokenizer.encode_image(image_path=cat)
    by_pil = tokenizer.encode_image(image_pil=image)
    proc = tokenizer.image_tokenizer.processor(image)
    by_torch = tokenizer.encode_image(image_torch=proc.to("cuda"))
    assert torch.equal(by_path, by_pil) and torch.equal(by_path, by_torch)
    dev3 = DevicePreprocessor(224, BICUBIC, keep_ratio=False)(image)
    assert torch.equal(dev3.cpu(), proc)
    assert torch.equal(tokenizer.encode_image(image_torch=dev3), by_path)
    # and the ids are the oracle's on the engine's own z (VQ bit-exact), as for synthetic inputs
    from seed_amd.tokenizer_engine import TokenizerEngine
    taps = {}
    eng = TokenizerEngine(sd, cfg, device="cuda")
    assert torch.equal(eng.encode(image_tensor[None], taps), ids)
    same_z = O.vq_argmin(taps["z"].float().cpu(), sd["quantize.embedding.weight"], O.Prec("bf16")).reshape(ids.shape)
    assert torch.equal(ids.cpu(), same_z)
    ids16 = O.get_codebook_indices(sd, image_tensor.cpu()[None], cfg, "bf16")
    print(f"[cat.jpg] ids[:8] {ids[0, :8].tolist()}  agreement with the bf16 oracle {(ids.cpu() == ids16).float().mean().item():.3f}")


@pytest.fixture(scope="module")
def llama_dir(tmp_path_factory):
    from transformers.models.llama.configuration_llama import LlamaConfig as HF
    from models.llama_xformer import LlamaForCausalLM
    cfg = C.LLAMA_TINY
    sd = make_llama_state_dict(cfg, seed=3, norm_jitter=0.05, dtype=torch.bfloat16)
    hf = HF(vocab_size=cfg.vocab, hidden_size=cfg.hidden, intermediate_size=cfg.ffn, num_hidden_layers=cfg.layers,
            num_attention_heads=cfg.heads, rms_norm_eps=cfg.rms_eps, max_position_embeddings=cfg.max_pos,
            pad_token_id=0, bos_token_id=1, eos_token_id=2)
    m = LlamaForCausalLM(hf)
    m.load_state_dict({k: v.float() for k, v in sd.items()}, strict=True)
    d = str(tmp_path_factory.mktemp("seed_llama"))
    m.save_pretrained(d)
    return d, sd, cfg


def _script_model(llama_dir):
    d, sd, cfg = llama_dir
    mcfg = yaml.safe_load(open(os.path.join(ROOT, "configs/llm/seed_llama_8b.yaml")))
    assert mcfg["_target_"] == "models.model_tools.get_pretrained_llama_causal_model"
    model = _instantiate(mcfg, pretrained_model_name_or_path=d, torch_dtype=torch.float16)      # seed_llama_inference_8B.py:77-78
    return model.eval().to("cuda"), sd, cfg                                                       # :79


def test_scripts_llm_construction_forward_and_generate(llama_dir):
    model, sd, cfg = _script_model(llama_dir)
    B, T0, n_new = 3, 9, 6
    ids = torch.randint(3, cfg.vocab, (B, T0), generator=torch.Generator().manual_seed(1))
    # LlamaForCausalLM.forward (llama_xformer.py:661-743) with the real engine
    out = model(input_ids=ids.cuda(), use_cache=True)
    ref, past = O.llama_forward(sd, cfg, ids, mode="fp32")
    rel = ((out.logits.float().cpu() - ref).norm() / ref.norm()).item()
    assert tuple(out.logits.shape) == (B, T0, cfg.vocab) and rel < 2e-2, rel
    pkv = out.past_key_values
    assert len(pkv) == cfg.layers and tuple(pkv[0][0].shape) == (B, cfg.heads, T0, cfg.head_dim)   # legacy layout (:236-239)
    assert ((pkv[1][0].float().cpu() - past[1][0]).norm() / past[1][0].norm()).item() < 1e-2
    # generate(), greedy: == the engine's own greedy loop == the oracle's on confident rows
    gen = model.generate(input_ids=ids.cuda(), max_new_tokens=n_new, do_sample=False, num_beams=1, eos_token_id=None, pad_token_id=0)
    assert tuple(gen.shape) == (B, T0 + n_new) and torch.equal(gen[:, :T0].cpu(), ids)
    eng_toks, _ = model.engine.greedy_decode(ids.cuda(), n_new)
    assert torch.equal(gen[:, T0:], eng_toks)
    t32, s32 = O.llama_greedy_decode(sd, cfg, ids, n_new, mode="fp32")
    top2 = s32.topk(2, dim=-1).values
    confident = (top2[..., 0] - top2[..., 1]) > 1e-2 * s32.abs().amax(-1)
    same = gen[:, T0:].cpu() == t32
    alive = torch.ones(B, dtype=torch.bool)
    for i in range(n_new):
        assert (same[:, i] | ~confident[:, i] | ~alive).all(), f"generate() token differs at confident step {i}"
        alive &= same[:, i]
    # the scripts' sampling configuration (seed_llama_inference_8B.py:81-87)
    torch.manual_seed(0)
    smp = model.generate(input_ids=ids[:1].cuda(), temperature=1.0, num_beams=1, max_new_tokens=5, top_p=0.5, do_sample=True,
                         pad_token_id=0)
    assert smp.shape[0] == 1 and smp.shape[1] <= T0 + 5 and torch.equal(smp[:, :T0].cpu(), ids[:1])
    # a batch above the engine's first capacity grows the cache instead of rebuilding from released parameters
    eng = model.engine
    big = torch.randint(3, cfg.vocab, (eng.batch_cap + 3, 5), generator=torch.Generator().manual_seed(2))
    o2 = model(input_ids=big.cuda())
    r2, _ = O.llama_forward(sd, cfg, big, mode="fp32")
    assert model.engine is eng and ((o2.logits.float().cpu() - r2).norm() / r2.norm()).item() < 2e-2


def test_beam_search_reorders_the_static_cache(llama_dir):
    model, sd, cfg = _script_model(llama_dir)
    calls = []
    orig = model._reorder_cache
    model._reorder_cache = lambda pkv, idx: (calls.append(idx.clone()), orig(pkv, idx))[1]
    ids = torch.randint(3, cfg.vocab, (2, 9), generator=torch.Generator().manual_seed(1))
    out = model.generate(input_ids=ids.cuda(), max_new_tokens=4, do_sample=False, num_beams=2, eos_token_id=None, pad_token_id=0)
    assert tuple(out.shape) == (2, 13) and torch.equal(out[:, :9].cpu(), ids)
    # _reorder_cache itself (llama_xformer.py:778-783): rows of the static cache are permuted; the next step equals a fresh run
    four = torch.randint(3, cfg.vocab, (4, 7), generator=torch.Generator().manual_seed(3))
    o = model(input_ids=four.cuda(), use_cache=True)
    idx = torch.tensor([2, 2, 0, 3], device="cuda")
    pkv = model._reorder_cache(o.past_key_values, idx)
    nxt = torch.randint(3, cfg.vocab, (4, 1), generator=torch.Generator().manual_seed(4)).cuda()
    got = model(input_ids=nxt, past_key_values=pkv, use_cache=True).logits
    o2 = model(input_ids=four[idx.cpu()].cuda(), use_cache=True)
    want = model(input_ids=nxt, past_key_values=o2.past_key_values, use_cache=True).logits
    assert torch.equal(got, want)
    print(f"[beam search] _reorder_cache called {len(calls)} times by generate(num_beams=2)")


def test_dp_pretokenizer_tool_end_to_end_on_device(tmp_path, monkeypatch):
    """SURVEY 8f-1 end to end on the device: ``seed_amd.tools.extract_image_ids.main`` on a directory of JPEG /
    PNG files of assorted sizes (incl. the reference's own fixture, tests/golden/cat.jpg = images/cat.jpg) with ``--gpu-preprocess``:
    directory -> JPEG decode -> DevicePreprocessor -> TokenizerEngine (full-size SEED-2, the tool's default weights) -> tar shards.
    The members must follow the reference tool's on-disk format (MultiModalLLM/src/tools/extract_image_ids_to_torchdata_parallel.py:
    114-127: pickled {'image_ids': [32 ints], 'text', 'metadata'}), and the ids must EQUAL ``TokenizerEngine.encode`` of the tensors
    the reference's own PIL transform (models/transforms.py:13-16) produces for the same files - device preprocessing is bit-exact
    and the tokenizer is a pure map over images, so equality is exact whatever the batch composition."""
    import pickle
    import shutil
    import tarfile
    import numpy as np
    from PIL import Image
    from models.transforms import get_transform
    from seed_amd.tokenizer_engine import TokenizerEngine
    from seed_amd.tools import extract_image_ids as tool
    src = tmp_path / "imgs"
    (src / "sub").mkdir(parents=True)
    shutil.copy(os.path.join(ROOT, "tests", "golden", "cat.jpg"), src / "cat.jpg")
    (src / "cat.txt").write_text("a cat on the grass\n")
    rs = np.random.RandomState(0)
    for name, (h, w) in (("a_wide.jpg", (180, 333)), ("b_tall.png", (401, 97)), ("sub/c_small.jpg", (64, 64)), ("sub/d_big.jpg", (512, 640))):
        Image.fromarray(rs.randint(0, 256, (h, w, 3)).astype(np.uint8)).save(src / name)
    out = tmp_path / "ids"
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    tool.main(["--images", str(src), "--save_dir", str(out), "--batch_size", "3", "--gpu-preprocess"])
    files = tool.list_images(str(src))
    assert len(files) == 5
    got = {}
    part = out / "part-0000"
    for tar in sorted(os.listdir(part)):
        with tarfile.open(part / tar) as tf:
            for m in tf.getmembers():
                s = pickle.loads(tf.extractfile(m).read())
                assert set(s) == {"image_ids", "text", "metadata"} and len(s["image_ids"]) == 32
                assert all(isinstance(v, int) and 0 <= v < 8192 for v in s["image_ids"])
                got[s["metadata"]["path"]] = (m.name, s)
    assert sorted(got) == files
    assert [got[f][0] for f in files] == [f"{i:09d}.pkl" for i in range(5)]                  # keys = positions in the sorted list
    assert got[str(src / "cat.jpg")][1]["text"] == "a cat on the grass" and got[str(src / "a_wide.jpg")][1]["text"] == ""
    # the same files through the reference's host transform and the engine directly
    tf_host = get_transform(type="clip", keep_ratio=False, image_size=224)
    x = torch.stack([tf_host(Image.open(f).convert("RGB")) for f in files]).cuda()
    eng = TokenizerEngine(make_tokenizer_state_dict(C.SEED2, seed=0, device="cuda"), C.SEED2, device="cuda")
    want = eng.encode(x).cpu().tolist()
    for f, row in zip(files, want):
        assert got[f][1]["image_ids"] == row, f
