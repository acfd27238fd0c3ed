#!/usr/bin/env python3
"""Time one generator optimisation step (cmgan_amd.training.generator_train_step: STFT of both batches, train-mode
TSCNet forward, ISTFT, loss, loss gradient, backward, gradient all-reduce (identity on one rank), AdamW) at the
reference's training shape - 2 s clips (cut_len 32000, train.py:25), batch 4 per GPU (train.py:22) - and larger
batches.  Prints one JSON line for profiles/ with the per-kernel-label split of the last step."""
import argparse, collections, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cmgan_amd.synth import make_state_dict, synthetic_clips
from cmgan_amd.training import AdamW, GeneratorTrain, generator_train_step

ap = argparse.ArgumentParser()
ap.add_argument("--batches", default="4,16")
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--cut-len", type=int, default=32000)
ap.add_argument("--attn-bwd", default=None, help="cores | fused: sets CMGAN_ATTN_BWD for the run")
ap.add_argument("--adversarial", action="store_true",
                help="time training.adversarial_train_step (generator + metric discriminator, given PESQ labels)")
args = ap.parse_args()
if args.attn_bwd:
    os.environ["CMGAN_ATTN_BWD"] = args.attn_bwd

gen = GeneratorTrain(make_state_dict(0), device="cuda:0")
opt = AdamW(gen.engine, gen.param_bucket, gen.grad_bucket, lr=5e-4)
tgen = torch.Generator(device="cuda:0").manual_seed(1)
if args.adversarial:
    from cmgan_amd.synth import discriminator_state_dict
    from cmgan_amd.training import DiscriminatorTrain, adversarial_train_step
    disc = DiscriminatorTrain(discriminator_state_dict(0), engine=gen.engine)
    opt_d = AdamW(gen.engine, disc.param_bucket, disc.grad_bucket, lr=1e-3)
res = {}
for B in [int(b) for b in args.batches.split(",")]:
    clean = synthetic_clips(B, args.cut_len, seed=5).cuda()
    noisy = (clean + 0.3 * synthetic_clips(B, args.cut_len, seed=6).cuda()).contiguous()
    if args.adversarial:
        pesq = torch.full((B,), 0.5, device="cuda:0")
        step = lambda: adversarial_train_step(gen, disc, opt, opt_d, clean, noisy, pesq, generator=tgen)[:1]
    else:
        step = lambda: generator_train_step(gen, opt, clean, noisy, generator=tgen)
    for _ in range(2):
        loss = step()[0]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()[0]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    gen.engine.set_profiling(True)
    step()
    torch.cuda.synchronize()
    split = collections.OrderedDict()
    for name, ms in gen.engine.profile():
        split[name] = split.get(name, 0.0) + ms
    gen.engine.set_profiling(False)
    top = sorted(split.items(), key=lambda kv: -kv[1])[:(24 if args.adversarial else 12)]
    res[f"batch{B}"] = {"ms_per_step": round(1e3 * dt, 2), "clips_per_s": round(B / dt, 2),
                        "frames_per_s": round(B * (args.cut_len // 100 + 1) / dt, 1), "loss": round(float(loss), 4),
                        "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 2),
                        "kernel_ms": {k: round(v, 2) for k, v in top}, "kernel_ms_total": round(sum(split.values()), 2)}
print(json.dumps({"workload": f"{'adversarial' if args.adversarial else 'generator'} train step, {args.cut_len}-sample clips, TSCNet(64,201) random-init, dropout 0.2, "
                              "split-f16 products in every conformer kernel except to_out, the dense and 1 x 3 convs and all weight gradients; fp32 tail convs / conv_1 / discriminator; 1 x MI355X", "results": res}))
