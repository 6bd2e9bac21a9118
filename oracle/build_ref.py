"""Compile the reference's own Python modules for this path into ``oracle/_ref/`` (test infrastructure).

    python -m oracle.build_ref            # run in the build container, where /root/reference exists

TEST INFRASTRUCTURE.  The reference is pure Python, so "building" it means byte-compiling: each of the reference
files on the path is compiled from where it lies under ``/root/reference`` with ``py_compile``
and only the OUTPUT (a sourceless ``.pyc``) is written under ``oracle/_ref/`` — the same rule as a C reference
compiled to ``oracle/_ref/*.so``: no reference source enters the repository, ``oracle/_ref/`` is git-ignored (so it
stays out of history) but not gpurun-ignored (so it travels to the GPU box like ``seed_amd/libseedmi.so`` does).
``oracle/ref_shims.py`` imports the modules from ``/root/reference`` when that tree exists and from these compiled
files otherwise, which lets ``bench.py`` time the reference's OWN CPU path on the bench node
(``cpu_baseline.kind == "reference"``) and lets ``-m gpu`` tests compare the HIP path with live reference modules.

Bytecode is tied to the interpreter version; the GPU box runs the same image (CPython 3.10.12).  A ``MANIFEST.json``
records the interpreter tag and a sha256 of every compiled source so that a stale ``_ref`` is detected.
"""
import hashlib
import json
import os
import py_compile
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
SRC_ROOT = "/root/reference"

# the files of the path (SURVEY.md section 8c "CPU restatement must follow") plus the siblings their imports pull in
FILES = [
    "models/seed_qformer/utils.py",
    "models/seed_qformer/eva_vit.py",
    "models/seed_qformer/qformer_causual.py",
    "models/seed_qformer/clip_vit.py",
    "models/seed_qformer/vit.py",
    "models/seed_qformer/blip2.py",
    "models/seed_qformer/qformer_quantizer.py",
    "models/llama_xformer.py",
]


def _sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def available() -> bool:
    """True when a compiled reference for THIS interpreter is present."""
    try:
        m = json.load(open(os.path.join(OUT, "MANIFEST.json")))
    except Exception:
        return False
    return m.get("python") == sys.version.split()[0] and all(
        os.path.exists(os.path.join(OUT, f + "c")) for f in m.get("files", {}))


def build(force: bool = False, verbose: bool = True):
    """No-op where /root/reference is absent (the GPU box only uses the prebuilt files)."""
    if not os.path.isdir(os.path.join(SRC_ROOT, "models", "seed_qformer")):
        if verbose:
            print(f"[oracle.build_ref] {SRC_ROOT} not present: keeping {'the prebuilt' if available() else 'NO'} oracle/_ref")
        return None
    shas = {f: _sha(os.path.join(SRC_ROOT, f)) for f in FILES}
    want = {"python": sys.version.split()[0], "files": shas,
            "what": "sourceless bytecode of the reference's own modules (py_compile output only; no reference source is copied)"}
    man = os.path.join(OUT, "MANIFEST.json")
    if not force and available():
        try:
            if json.load(open(man)) == want:
                if verbose:
                    print("[oracle.build_ref] oracle/_ref up to date")
                return OUT
        except Exception:
            pass
    for f in FILES:
        dst = os.path.join(OUT, f + "c")
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        # dfile: the name tracebacks show; UNCHECKED_HASH: the loader never looks for the (absent) source file
        py_compile.compile(os.path.join(SRC_ROOT, f), cfile=dst, dfile="reference:" + f, doraise=True,
                           invalidation_mode=py_compile.PycInvalidationMode.UNCHECKED_HASH)
    json.dump(want, open(man, "w"), indent=1)
    if verbose:
        print(f"[oracle.build_ref] compiled {len(FILES)} reference modules into {OUT}")
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
