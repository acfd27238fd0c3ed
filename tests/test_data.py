"""cmgan_amd.data - the reference's dataset length rule, pairing, ordering and loaders (CPU)."""
import os
import random

import numpy as np
import pytest
import torch
from scipy.io import wavfile

from cmgan_amd.data import DemandDataset, fit_length, load_data, read_wav


def _make(root, split, names_lengths, seed=0):
    g = np.random.Generator(np.random.PCG64(seed))
    for sub in ("clean", "noisy"):
        os.makedirs(os.path.join(root, split, sub), exist_ok=True)
    for name, n in names_lengths:
        clean = (g.standard_normal(n) * 3000).astype(np.int16)
        noisy = (clean + g.standard_normal(n) * 500).astype(np.int16)
        wavfile.write(os.path.join(root, split, "clean", name), 16000, clean)
        wavfile.write(os.path.join(root, split, "noisy", name), 16000, noisy)


def test_short_clips_are_repeated_and_long_clips_cropped_like_the_reference(tmp_path):
    _make(str(tmp_path), "train", [("p1_10.wav", 700), ("p1_2.wav", 5000), ("p1_3.wav", 1000)])
    ds = DemandDataset(os.path.join(str(tmp_path), "train"), cut_len=1000)
    assert ds.clean_wav_name == ["p1_2.wav", "p1_3.wav", "p1_10.wav"]          # natural order (natsorted)
    # long clip: random.randint(0, length - cut_len) from the module RNG, same slice for clean and noisy
    random.seed(7)
    clean, noisy, length = ds[0]
    random.seed(7)
    start = random.randint(0, 5000 - 1000)
    full_c, sr = read_wav(os.path.join(str(tmp_path), "train", "clean", "p1_2.wav"))
    full_n, _ = read_wav(os.path.join(str(tmp_path), "train", "noisy", "p1_2.wav"))
    assert sr == 16000 and length == 5000 and clean.dtype == torch.float32
    assert torch.equal(clean, full_c[start:start + 1000]) and torch.equal(noisy, full_n[start:start + 1000])
    # exact length: returned as is (randint(0, 0))
    clean, _, length = ds[1]
    assert length == 1000 and clean.numel() == 1000
    # short clip: cut_len // length copies + the head (dataloader.py:34-44)
    clean, noisy, length = ds[2]
    full_c, _ = read_wav(os.path.join(str(tmp_path), "train", "clean", "p1_10.wav"))
    assert length == 700 and torch.equal(clean, torch.cat([full_c, full_c[:300]]))
    assert noisy.numel() == 1000


def test_fit_length_rule_and_pcm_normalisation(tmp_path):
    x = torch.arange(7, dtype=torch.float32)
    assert fit_length(x, 16).tolist() == [0, 1, 2, 3, 4, 5, 6, 0, 1, 2, 3, 4, 5, 6, 0, 1]
    assert fit_length(x, 3, start=2).tolist() == [2, 3, 4]
    p = str(tmp_path / "a.wav")
    wavfile.write(p, 16000, np.array([-32768, 0, 16384, 32767], dtype=np.int16))
    got, _ = read_wav(p)
    assert got.tolist() == [-1.0, 0.0, 0.5, 32767 / 32768]


def test_mismatched_pair_is_an_error(tmp_path):
    _make(str(tmp_path), "train", [("a.wav", 900)])
    wavfile.write(os.path.join(str(tmp_path), "train", "noisy", "a.wav"), 16000, np.zeros(800, dtype=np.int16))
    with pytest.raises(ValueError):
        DemandDataset(os.path.join(str(tmp_path), "train"), 1000)[0]


def test_load_data_batches_have_the_front_end_shape(tmp_path):
    _make(str(tmp_path), "train", [(f"s_{i}.wav", 600 + 400 * i) for i in range(5)])
    _make(str(tmp_path), "test", [(f"t_{i}.wav", 1500) for i in range(3)], seed=1)
    train, test = load_data(str(tmp_path), batch_size=2, n_cpu=0, cut_len=1200)
    batches = list(train)
    assert len(batches) == 2                                             # drop_last: 5 clips -> 2 batches of 2
    clean, noisy, length = batches[0]
    # the reference samples through DistributedSampler(shuffle=True, seed=0): epoch-0 permutation of the 5 clips
    g = torch.Generator()
    g.manual_seed(0)
    perm = torch.randperm(5, generator=g).tolist()
    assert clean.shape == (2, 1200) and noisy.shape == (2, 1200)
    assert length.tolist() == [600 + 400 * perm[0], 600 + 400 * perm[1]]
    assert perm != sorted(perm)                                          # really shuffled, like the reference
    train.sampler.set_epoch(1)
    assert [b[2].tolist() for b in train] != [b[2].tolist() for b in batches]
    assert sum(b[0].size(0) for b in test) == 3                          # test keeps the ragged last batch
