"""GraphedTrainStep: parity with the eager step (dropout off) and replay timing (dropout on)."""
import sys, time, torch
sys.path.insert(0, '/root/repo')
from cmgan_amd.synth import discriminator_state_dict, make_state_dict, synthetic_clips
from cmgan_amd.training import (AdamW, DiscriminatorTrain, GeneratorTrain, GraphedTrainStep, adversarial_train_step,
                                generator_train_step)
DEV = "cuda:0"
B, L = 4, 32000
clean = synthetic_clips(B, L, seed=5).to(DEV)
noisy = (clean + 0.3 * synthetic_clips(B, L, seed=6).to(DEV)).contiguous()
pesq = torch.full((B,), 0.5, device=DEV)
sd, dsd = make_state_dict(0), discriminator_state_dict(0)

def fresh(adv):
    gen = GeneratorTrain(sd, device=DEV)
    og = AdamW(gen.engine, gen.param_bucket, gen.grad_bucket, lr=5e-4)
    if not adv:
        return gen, og, None, None
    disc = DiscriminatorTrain(dsd, engine=gen.engine)
    return gen, og, disc, AdamW(gen.engine, disc.param_bucket, disc.grad_bucket, lr=1e-3)

for adv in (False, True):
    # parity, dropout off
    gen, og, disc, od = fresh(adv)
    for _ in range(2):
        if adv: le = adversarial_train_step(gen, disc, og, od, clean, noisy, pesq, masks=None, disc_masks=None)[0]
        else: le = generator_train_step(gen, og, clean, noisy, masks=None)[0]
    gen2, og2, disc2, od2 = fresh(adv)
    step = GraphedTrainStep(gen2, og2, B, L, disc2, od2, dropout=False)
    for _ in range(2):
        lg = step(clean, noisy, pesq if adv else None)[0]
    torch.cuda.synchronize()
    dp = float((gen.param_bucket.flat - gen2.param_bucket.flat).abs().max())
    print(f"adv={adv}: eager loss {float(le):.6f} graph loss {float(lg):.6f} max |param diff| after 2 steps {dp:.3e} "
          f"t={og2.t} bn_count={gen2.blocks[0].time.conv.num_batches_tracked}"
          + (f" disc param diff {float((disc.param_bucket.flat - disc2.param_bucket.flat).abs().max()):.3e}" if adv else ""))
    # timing, dropout on
    gen3, og3, disc3, od3 = fresh(adv)
    run_e = (lambda: adversarial_train_step(gen3, disc3, og3, od3, clean, noisy, pesq)) if adv else (lambda: generator_train_step(gen3, og3, clean, noisy))
    for _ in range(2): run_e()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): run_e()
    torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 5
    step3 = GraphedTrainStep(gen3, og3, B, L, disc3, od3)
    for _ in range(2): step3(clean, noisy, pesq if adv else None)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): out = step3(clean, noisy, pesq if adv else None)
    torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 5
    print(f"adv={adv}: eager {1e3 * te:.1f} ms/step, graph replay {1e3 * tg:.1f} ms/step, loss {float(out[0]):.4f}")
