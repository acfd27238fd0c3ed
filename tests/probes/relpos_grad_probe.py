"""Which part of the training attention backward loses the rel_pos_emb gradient at larger batches?  Whole generator step on
the kink-free twin at B = 2 / 4 x T = 321 against oracle autograd, fused backward vs the three cores (CMGAN_ATTN_BWD)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cmgan_oracle as O  # noqa: E402
from oracle.weights import make_state_dict  # noqa: E402
from cmgan_amd.synth import kink_free_twin, synthetic_clips, synthetic_dropout_masks  # noqa: E402
from cmgan_amd.training import AdamW, GeneratorTrain, generator_train_step  # noqa: E402

DEV = torch.device("cuda")
sd = kink_free_twin(make_state_dict(seed=0))
for B in [int(a) for a in sys.argv[1:]] or [2, 4]:
    clean = synthetic_clips(B, 32000, seed=43)
    noisy = clean + 0.3 * synthetic_clips(B, 32000, seed=44)
    npm = synthetic_dropout_masks(92, B, 321, 101)
    tm = lambda dev=None: [tuple({k: (torch.from_numpy(v) if dev is None else torch.from_numpy(v).to(dev))
                                  for k, v in d.items()} for d in pair) for pair in npm]
    want = O.generator_step_gradients(sd, clean, noisy, tm())
    scale = max(float(v.abs().max()) for v in want["grads"].values())
    for mode in ("fused", "cores"):
        os.environ["CMGAN_ATTN_BWD"] = mode
        gen = GeneratorTrain(sd, device=DEV)
        opt = AdamW(gen.engine, gen.param_bucket, gen.grad_bucket, lr=5e-4)
        generator_train_step(gen, opt, clean.to(DEV), noisy.to(DEV), masks=tm(DEV), update=False)
        errs = []
        for k, w in want["grads"].items():
            mx = float(w.abs().max())
            if mx >= 1e-6 * scale:
                errs.append((float((gen.grads[k].cpu() - w).abs().max()) / mx, k, mx / scale))
        errs.sort(reverse=True)
        print(f"B={B} {mode}: " + "; ".join(f"{e:.2e} {k} (max {m:.1e} of largest)" for e, k, m in errs[:4]), flush=True)
        rp = [(e, k) for e, k, _ in errs if "rel_pos" in k]
        print("   rel_pos_emb: " + ", ".join(f"{k.split('.')[0]}.{k.split('.')[1][:4]} {e:.1e}" for e, k in rp), flush=True)
        del gen, opt
        torch.cuda.empty_cache()
