"""Drive the REFERENCE's own modules (``oracle/_ref/*.pyc``, built by ``oracle/make_ref.py``) on CPU.

TEST INFRASTRUCTURE ONLY: imported by ``tests/`` (checker) and by ``bench.py``'s ``cpu_baseline`` leg
(``kind: "reference"``), never by ``cmgan_amd``.

``evaluation.py`` itself cannot be imported (torchaudio / natsort / soundfile / pesq are absent, it parses ``sys.argv``
and hard-codes ``.cuda()``), so the glue of ``enhance_one_track`` (src/evaluation.py:21-53) is restated here around the
reference's ``TSCNet`` / ``power_compress`` / ``power_uncompress``, with the two torch >= 2 adapters for
``torch.stft`` / ``torch.istft`` - the same driver ``tests/golden/make_golden.py`` made the fixtures with.
"""
import importlib.machinery
import importlib.util
import json
import math
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "_ref")


def available() -> bool:
    return os.path.exists(os.path.join(REF, "MANIFEST.json"))


def _load_pyc(name: str, rel: str):
    path = os.path.join(REF, rel)
    loader = importlib.machinery.SourcelessFileLoader(name, path)
    spec = importlib.util.spec_from_loader(name, loader, origin=path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    loader.exec_module(mod)
    return mod


_mods = None


def modules():
    """(generator, conformer, utils) modules of the reference, loaded from bytecode.  generator.py does
    ``from models.conformer import ConformerBlock`` (generator.py:1), so a ``models`` namespace is registered."""
    global _mods
    if _mods is None:
        if not available():
            raise RuntimeError("oracle/_ref is not built: run `python oracle/make_ref.py` in the build container")
        with open(os.path.join(REF, "MANIFEST.json")) as f:
            magic = json.load(f)["magic"]
        if magic != importlib.util.MAGIC_NUMBER.hex():
            raise RuntimeError("oracle/_ref bytecode was built by a different CPython; rebuild it")
        if "models" in sys.modules and not getattr(sys.modules["models"], "_cmgan_ref", False):
            raise RuntimeError("a foreign `models` package is already imported")
        pkg = types.ModuleType("models")
        pkg.__path__ = []
        pkg._cmgan_ref = True
        sys.modules["models"] = pkg
        conf = _load_pyc("models.conformer", "models/conformer.pyc")
        gen = _load_pyc("models.generator", "models/generator.pyc")
        pkg.conformer, pkg.generator = conf, gen
        utl = _load_pyc("_cmgan_ref_utils", "utils.pyc")
        _mods = (gen, conf, utl)
    return _mods


def tscnet(sd, num_features: int = 201):
    """The reference ``TSCNet(64, num_features)`` in eval mode with the 359-entry state_dict loaded strictly
    (src/evaluation.py:63-65)."""
    gen, _, _ = modules()
    model = gen.TSCNet(num_channel=64, num_features=num_features)
    model.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()}, strict=True)
    return model.eval()


def _stft(x, n_fft, hop):
    return torch.view_as_real(torch.stft(x, n_fft, hop, window=torch.hamming_window(n_fft), onesided=True,
                                         return_complex=True))


def _istft(spec, n_fft, hop):
    return torch.istft(torch.view_as_complex(spec.contiguous()), n_fft, hop, window=torch.hamming_window(n_fft),
                       onesided=True)


@torch.no_grad()
def enhance_rows(model, noisy: torch.Tensor, n_fft: int = 400, hop: int = 100):
    """evaluation.py:36-51 on already scaled / padded rows [B, L]: stft -> power_compress -> TSCNet ->
    power_uncompress -> istft."""
    _, _, utl = modules()
    spec = utl.power_compress(_stft(noisy, n_fft, hop)).permute(0, 1, 3, 2)
    est_real, est_imag = model(spec)
    est_real, est_imag = est_real.permute(0, 1, 3, 2), est_imag.permute(0, 1, 3, 2)
    return _istft(utl.power_uncompress(est_real, est_imag).squeeze(1), n_fft, hop)


@torch.no_grad()
def enhance(model, noisy: torch.Tensor, cut_len: int = 16000 * 16, n_fft: int = 400, hop: int = 100):
    """noisy [1, L] -> enhanced [L]: evaluation.py:21-53 (RMS scale, wrap-pad to a hop multiple, > cut_len rows)."""
    c = torch.sqrt(noisy.size(-1) / torch.sum((noisy ** 2.0), dim=-1))
    noisy = torch.transpose(torch.transpose(noisy, 0, 1) * c, 0, 1)
    length = noisy.size(-1)
    padded_len = int(math.ceil(length / 100)) * 100
    noisy = torch.cat([noisy, noisy[:, :padded_len - length]], dim=-1)
    if padded_len > cut_len:
        batch_size = int(math.ceil(padded_len / cut_len))
        while 100 % batch_size != 0:
            batch_size += 1
        noisy = torch.reshape(noisy, (batch_size, -1))
    return torch.flatten(enhance_rows(model, noisy, n_fft, hop) / c)[:length]


@torch.no_grad()
def enhance_batch(model, wav: torch.Tensor, n_fft: int = 400, hop: int = 100):
    """The benchmark's batched form: per-row RMS scale as in train.py:75-79, then the rows pipeline, un-scaled."""
    c = torch.sqrt(wav.size(-1) / torch.sum((wav ** 2.0), dim=-1))
    return enhance_rows(model, wav * c[:, None], n_fft, hop) / c[:, None]
