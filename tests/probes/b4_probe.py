"""Localise the B = 4 x T = 321 whole-step gradient discrepancy: forward outputs and output gradients per clip."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cmgan_oracle as O  # noqa: E402
from oracle.weights import make_state_dict  # noqa: E402
from cmgan_amd._lib import check  # noqa: E402
from cmgan_amd.synth import kink_free_twin, synthetic_clips, synthetic_dropout_masks  # noqa: E402
from cmgan_amd.training import GeneratorTrain  # noqa: E402

DEV = torch.device("cuda")
sd = kink_free_twin(make_state_dict(seed=0))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
clean = synthetic_clips(B, 32000, seed=43)
noisy = clean + 0.3 * synthetic_clips(B, 32000, seed=44)
npm = synthetic_dropout_masks(92, B, 321, 101)
tm = lambda dev=None: [tuple({k: (torch.from_numpy(v) if dev is None else torch.from_numpy(v).to(dev))
                              for k, v in d.items()} for d in pair) for pair in npm]
want = O.generator_step_gradients(sd, clean, noisy, tm())
rel = lambda a, b: float((a.cpu() - b).abs().max() / b.abs().max())
gen = GeneratorTrain(sd, device=DEV)
eng = gen.engine
cl, nz = clean.to(DEV), noisy.to(DEV)
c = eng.rms_scale(nz)
spec, cspec = eng.stft_compress(nz, c), eng.stft_compress(cl, c)
er, ei = gen.forward(spec, tm(DEV))
for b in range(B):
    print(f"clip {b}: est_real {rel(er[b], want['est_real'][b]):.2e} est_imag {rel(ei[b], want['est_imag'][b]):.2e}", flush=True)
# loss gradient at the ORACLE's operating point
wr, wi = want["est_real"].to(DEV).contiguous(), want["est_imag"].to(DEV).contiguous()
audio = eng.uncompress_istft(wr, wi)
La = audio.shape[-1]
ccut = cl[:, :La].contiguous()
d_real, d_imag = torch.empty_like(wr), torch.empty_like(wi)
check(eng._h, eng.lib.cmgan_loss_backward(eng._h, wr.data_ptr(), wi.data_ptr(), cspec.data_ptr(), B, 321, audio.data_ptr(),
                                          ccut.data_ptr(), 0.1, 0.9, 0.2, d_real.data_ptr(), d_imag.data_ptr(), eng._stream()))
for b in range(B):
    print(f"clip {b}: d_real {rel(d_real[b], want['d_real'][b]):.2e} d_imag {rel(d_imag[b], want['d_imag'][b]):.2e}", flush=True)
# backward from the ORACLE's output gradients (HIP forward state): parameter gradients
gen.backward(want["d_real"].to(DEV).contiguous(), want["d_imag"].to(DEV).contiguous())
scale = max(float(v.abs().max()) for v in want["grads"].values())
errs = []
for k, w in want["grads"].items():
    mx = float(w.abs().max())
    if mx >= 1e-6 * scale:
        errs.append((float((gen.grads[k].cpu() - w).abs().max()) / mx, k))
errs.sort(reverse=True)
print("backward from the oracle's d_real / d_imag: " + "; ".join(f"{e:.2e} {k}" for e, k in errs[:6]), flush=True)
