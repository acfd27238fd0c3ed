#!/usr/bin/env python3
"""Known answers for cmgan_amd.metrics from the REFERENCE's own logs (build container only).

`src/tools/Noisy_metrics_results/python_noisy_metrics.log` holds the reference tool's six scores for the 824 noisy
test tracks of VoiceBank+DEMAND; 25 of those tracks ship with the reference repo (`AudioSamples/{noisy,clean}`).  This
script stores the int16 PCM of the three shortest pairs next to their logged lines.  The log was produced with
`compute_metrics(clean.wav, noisy.wav, 16000, path=1)`, i.e. on `wavfile.read`'s int16-scale samples (the WSS term is
not scale-invariant: the same tracks scaled by 1/32768 give CSIG / CBAK / COVL 0.14 - 0.24 higher) - with that scale the
reference tool run here reproduces all six logged figures of all 25 tracks to the log's 6 decimals (PESQ given).
"""
import os
import re

import numpy as np
from scipy.io import wavfile

HERE = os.path.dirname(os.path.abspath(__file__))
SAMPLES = "/root/reference/AudioSamples"
LOG = "/root/reference/src/tools/Noisy_metrics_results/python_noisy_metrics.log"
LINE = re.compile(r"Track name: (\S+)\s+PESQ: (\S+)\s+CSIG: (\S+)\s+CBAK: (\S+)\s+COVL: (\S+)\s+SSNR: (\S+)\s+STOI: (\S+)")


def logged():
    out = {}
    for line in open(LOG):
        m = LINE.match(line)
        if m:
            out[m.group(1)] = [float(v) for v in m.groups()[1:]]
    return out


def main():
    log = logged()
    names = sorted(os.path.splitext(f)[0] for f in os.listdir(os.path.join(SAMPLES, "noisy")))
    size = {n: wavfile.read(os.path.join(SAMPLES, "noisy", n + ".wav"))[1].shape[0] for n in names}
    pick = sorted(names, key=lambda n: size[n])[:3]
    out = {"names": np.array(pick)}
    for n in pick:
        for kind in ("noisy", "clean"):
            sr, pcm = wavfile.read(os.path.join(SAMPLES, kind, n + ".wav"))
            assert sr == 16000 and pcm.dtype == np.int16
            out[f"{kind}_{n}"] = pcm
        out[f"log_{n}"] = np.array(log[n])            # PESQ CSIG CBAK COVL SSNR STOI as printed (6 decimals)
    np.savez_compressed(os.path.join(HERE, "metrics_known_answers.npz"), **out)
    print(pick, [size[n] for n in pick])


if __name__ == "__main__":
    main()
