"""Build libcmgan_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m cmgan_amd.build [--force]

The library lands in cmgan_amd/lib/ (git-ignored, but shipped to the GPU box by gpurun).
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libcmgan_hip.so")
SOURCES = ["api.hip", "api_train.hip", "conformer.hip", "conformer_x3.hip", "attn32_x3.hip", "ffn32_x3.hip", "conv.hip", "conv_x3.hip",
           "stft.hip", "stft_fft.hip", "train.hip", "train_x3.hip", "disc.hip"]
# pure-VALU kernels that WANT packed fp32 instructions (complex arithmetic of the FFT front / back end): built without
# the "-packed-fp32-ops" feature removal below
PACKED_FP32_SOURCES = {"stft_fft.hip"}
# the F16X1 (single fp16 product) twins of the x3 kernels: the same sources compiled a second time with X1_FLAGS
X1_SOURCES = ["conformer_x3.hip", "attn32_x3.hip", "ffn32_x3.hip", "conv_x3.hip"]
X1_FLAGS = ["-DX3_SINGLE", "-DX3_TERMS=1"]
HEADERS = ["common.hip.h", "kernels.h", "weights.h", "api_internal.h", "train.h", "stft_fft_tables.h",
           os.path.join("..", "..", "include", "cmgan_hip.h")]
# the generator FORWARD path: what bench.py measures and what the PMC evidence under profiles/ was collected on.  The
# training-step slices (train.hip, api_train.hip, train.h) and the public header (which grows with them) are left
# out, so committed counters stay valid while the training side is being built.
INFERENCE_FILES = ["api.hip", "conformer.hip", "conformer_x3.hip", "attn32_x3.hip", "ffn32_x3.hip", "conv.hip", "conv_x3.hip", "stft.hip",
                   "stft_fft.hip",
                   "common.hip.h", "kernels.h", "weights.h", "api_internal.h", "stft_fft_tables.h"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -packed-fp32-ops: v_pk_{fma,mul,add}_f32 issue slower than the scalar pair they replace when the SIMD is
# also feeding MFMAs (measured: the whole step is 0.6% faster without them, dwpw2 with hand-packed FMAs
# was 18% slower), so the SLP vectoriser is told the target has none.  The flag is a device feature; the
# host pass of hipcc prints one "not a recognized feature" line per file for it, filtered below.
NO_PACKED_FP32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast",
         "-Wno-unused-result", "-Wno-unused-value", *NO_PACKED_FP32]
_NOISE = "is not a recognized feature for this target"


def _digest() -> str:
    h = hashlib.sha256()
    for name in SOURCES + HEADERS:
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS + X1_FLAGS).encode())
    return h.hexdigest()


def inference_digest() -> str:
    """sha256 of the inference-path kernel sources + build flags (see INFERENCE_FILES)."""
    h = hashlib.sha256()
    for name in INFERENCE_FILES:
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True, variant: str | None = None, extra_flags=()) -> str:
    """Build the library.  `variant` builds a side-by-side copy under lib/variants/<name>/ (with
    `extra_flags`), which `CMGAN_HIP_LIB=<path>` selects at load time: A/B timing of two builds inside
    ONE GPU session is the only comparison that is not swamped by box-to-box clock differences."""
    libdir = LIBDIR if variant is None else os.path.join(LIBDIR, "variants", variant)
    lib = os.path.join(libdir, "libcmgan_hip.so")
    flags = [*FLAGS, *extra_flags]
    os.makedirs(libdir, exist_ok=True)
    stamp = os.path.join(libdir, "libcmgan_hip.stamp")
    digest = _digest() + "|" + " ".join(extra_flags)
    if not force and os.path.exists(lib) and os.path.exists(stamp) and open(stamp).read() == digest:
        return lib
    objs = []

    def compile_one(job):
        src, extra, suffix = job
        obj = os.path.join(libdir, src.replace(".hip", suffix + ".o"))
        fl = flags if src not in PACKED_FP32_SOURCES else [f for f in flags if f not in NO_PACKED_FP32]
        cmd = [HIPCC, *fl, *extra, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
        err = "\n".join(l for l in r.stderr.splitlines() if _NOISE not in l)
        if err.strip():
            print(err, file=sys.stderr, flush=True)
        if r.returncode != 0:
            raise subprocess.CalledProcessError(r.returncode, cmd)
        return obj

    jobs = [(src, [], "") for src in SOURCES] + [(src, X1_FLAGS, "_x1") for src in X1_SOURCES]
    with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
        objs = list(ex.map(compile_one, jobs))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    with open(stamp, "w") as f:
        f.write(digest)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
