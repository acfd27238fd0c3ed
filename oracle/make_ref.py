#!/usr/bin/env python3
"""Build ``oracle/_ref/``: the REFERENCE's own generator modules, compiled where they lie.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

The reference is Python, so "compiling it from its own few source files" (the recipe a C reference gets a Makefile
for) is ``py_compile``: this script byte-compiles

    /root/reference/src/models/generator.py   (TSCNet, DenseEncoder, TSCB, MaskDecoder, ComplexDecoder)
    /root/reference/src/models/conformer.py   (ConformerBlock, Attention, FeedForward, ConformerConvModule)
    /root/reference/src/utils.py              (power_compress, power_uncompress)

straight from ``/root/reference`` into ``oracle/_ref/{models/generator,models/conformer,utils}.pyc`` - binaries only, no
reference source text enters the repo.  ``oracle/_ref/`` is git-ignored (it stays out of history) but NOT
gpurun-ignored, so it travels to the GPU box like the built ``.so``; both boxes run the same image (CPython 3.10), so
the bytecode loads there.  ``oracle/ref_runner.py`` imports the modules sourcelessly and drives them through the
20-line glue of ``src/evaluation.py:21-53``.

Run in the build container (``__graft_entry__.build()`` does it whenever /root/reference exists):

    python oracle/make_ref.py
"""
import hashlib
import json
import os
import py_compile
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = "/root/reference/src"
OUT = os.path.join(HERE, "_ref")
FILES = ("models/generator.py", "models/conformer.py", "utils.py")


def make_ref(verbose: bool = True) -> bool:
    """Returns True when oracle/_ref is (now) populated, False when there is no reference tree to compile."""
    if not os.path.isdir(REF_SRC):
        if verbose:
            print(f"make_ref: {REF_SRC} absent (GPU box?) - using the prebuilt oracle/_ref as it is")
        return os.path.exists(os.path.join(OUT, "MANIFEST.json"))
    manifest = {"python": sys.version.split()[0], "magic": __import__("importlib.util").util.MAGIC_NUMBER.hex(),
                "files": {}}
    for rel in FILES:
        src = os.path.join(REF_SRC, rel)
        dst = os.path.join(OUT, rel[:-3] + ".pyc")
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        # dfile: the path tracebacks will show - the reference file the bytecode came from
        py_compile.compile(src, cfile=dst, dfile=src, doraise=True,
                           invalidation_mode=py_compile.PycInvalidationMode.UNCHECKED_HASH)
        with open(src, "rb") as f:
            manifest["files"][rel] = {"sha256": hashlib.sha256(f.read()).hexdigest(), "pyc": os.path.relpath(dst, OUT)}
    with open(os.path.join(OUT, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    if verbose:
        print(f"make_ref: compiled {len(FILES)} reference modules into {OUT}")
    return True


if __name__ == "__main__":
    sys.exit(0 if make_ref() else 1)
