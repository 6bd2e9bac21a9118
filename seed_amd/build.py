"""Build libseedmi.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m seed_amd.build            # incremental; libseedmi.so (bf16) AND libseedmi_f16.so (fp16)
    python -m seed_amd.build --force
    python -m seed_amd.build --bf16 | --f16 | --devtools      # one of them

The .so stays inside the package directory (git-ignored, but it travels with gpurun snapshots) so the
driver sees the native code that the tests load.  hipcc cross-compiles for gfx950 without a GPU.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_build")
LIB = os.path.join(HERE, "libseedmi.so")
LIB_DEV = os.path.join(HERE, "libseedmi_dev.so")
DEVTOOLS_CSRC = os.path.join(os.path.dirname(HERE), "tools", "devtools_csrc")
LIB_F16 = os.path.join(HERE, "libseedmi_f16.so")          # the same sources with -DSEEDMI_F16: IEEE fp16 as the 16-bit element (csrc/common.h)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
SOURCES = ["capi.hip", "gemm_bf16.hip", "attn_fullrow.hip", "attn_vit.hip", "norm_misc.hip", "vq_argmin.hip", "tokenizer.hip", "detokenizer.hip", "preprocess.hip", "sample.hip", "llama.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-unused-result"]


def _deps_mtime():
    latest = 0.0
    for root in (CSRC, os.path.join(os.path.dirname(HERE), "include")):
        for f in os.listdir(root):
            if f.endswith((".h", ".hip")):
                latest = max(latest, os.path.getmtime(os.path.join(root, f)))
    return latest


def _compile(src, force, objdir=OBJ, extra=()):
    obj = os.path.join(objdir, src.replace(".hip", ".o"))
    srcp = os.path.join(CSRC, src)
    hdr_m = max(os.path.getmtime(os.path.join(d, f)) for d in (CSRC, DEVTOOLS_CSRC) if os.path.isdir(d) for f in os.listdir(d) if f.endswith((".h", ".inc")))
    hdr_m = max(hdr_m, os.path.getmtime(os.path.join(os.path.dirname(HERE), "include", "seedmi.h")))
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(srcp), hdr_m):
        return obj, False
    cmd = [HIPCC] + FLAGS + list(extra) + ["-c", srcp, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    return obj, True


def build_variant(name: str, defines, verbose: bool = True) -> str:
    """A/B build of the product library with extra -D flags: seed_amd/libseedmi_<name>.so (tools select it with SEEDMI_LIB_PATH)."""
    objdir = OBJ + "_" + name
    lib = os.path.join(HERE, f"libseedmi_{name}.so")
    os.makedirs(objdir, exist_ok=True)
    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        results = list(ex.map(lambda s: _compile(s, False, objdir, tuple(defines)), SOURCES))
    objs = [o for o, _ in results]
    if any(changed for _, changed in results) or not os.path.exists(lib):
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[seed_amd.build] linked {lib}")
    return lib


def build(force: bool = False, verbose: bool = True, devtools: bool = False, f16: bool = False) -> str:
    """devtools=True builds libseedmi_dev.so with -DSEEDMI_DEVTOOLS: the product library plus timing-only ablation switches,
    rejected kernel variants and micro-benchmarks that tools/ scripts use (SEEDMI_LIB_PATH selects it); never loaded by the
    engines, tests or bench.py.  f16=True builds libseedmi_f16.so (-DSEEDMI_F16): the product library with IEEE fp16 as its
    16-bit element - what the engines load for torch.float16 weights."""
    assert not (devtools and f16)
    objdir = OBJ + ("_dev" if devtools else "_f16" if f16 else "")
    lib = LIB_DEV if devtools else LIB_F16 if f16 else LIB
    # the lab build's rejected kernel variants and ablation switches live outside the product sources (tools/devtools_csrc/*.inc)
    extra = ("-DSEEDMI_DEVTOOLS", "-I" + DEVTOOLS_CSRC) if devtools else ("-DSEEDMI_F16",) if f16 else ()
    os.makedirs(objdir, exist_ok=True)
    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        results = list(ex.map(lambda s: _compile(s, force, objdir, extra), SOURCES))
    objs = [o for o, _ in results]
    LIB_ = lib
    if any(changed for _, changed in results) or not os.path.exists(LIB_) or force:
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[seed_amd.build] linked {LIB_}")
    elif verbose:
        print(f"[seed_amd.build] {LIB_} up to date")
    return LIB_


if __name__ == "__main__":
    if "--variant" in sys.argv:          # python -m seed_amd.build --variant late -DSEEDMI_LATE_PROLOGUE=1
        i = sys.argv.index("--variant")
        build_variant(sys.argv[i + 1], [a for a in sys.argv[i + 2:] if a.startswith("-D")])
    else:
        if "--devtools" in sys.argv or "--f16" in sys.argv or "--bf16" in sys.argv:
            build(force="--force" in sys.argv, devtools="--devtools" in sys.argv, f16="--f16" in sys.argv)
        else:                            # no selector: BOTH product libraries (the engines load libseedmi_f16.so for .half() / torch.float16)
            build(force="--force" in sys.argv)
            build(force="--force" in sys.argv, f16=True)
