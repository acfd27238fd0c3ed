#!/usr/bin/env python3
"""Benchmark of the CMGAN generator forward path on MI355X (BASELINE.json metric:
enhanced audio frames/sec, 16 kHz, 2 s clips, batch 32 per GPU).

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" = one pass of the whole device pipeline (RMS scale -> STFT -> power compress ->
TSCNet -> power uncompress -> ISTFT) over one batch of 32 synthetic 2 s clips per GPU that
are already resident in HBM.  For N > 1 the driver launches one rank per GPU with
torch.distributed.run; utterances are sharded by rank (weak scaling: 32 clips per GPU), the
forward needs no collective, and each step ends with ONE all-reduce of two scalars
(RCCL over xGMI), as north_star specifies.  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

CLIP_LEN = 32000          # 2 s @ 16 kHz                       (src/train.py:22)
BATCH_PER_GPU = 32        # BASELINE.json configs[1]
N_FFT, HOP = 400, 100
T_FRAMES = CLIP_LEN // HOP + 1                                 # 321 (src/train.py:53)
F_BINS, F2 = 201, 101
FP32_MFMA_PEAK_TF = 157.3                                      # MI355X_MICROARCH.md (f32 in / f32 acc MFMA)
F16_MFMA_PEAK_TF = 2500.0                                      # dense f16/bf16 MFMA (not the 2:1-sparse headline)
HBM_PEAK_GBS = 8000.0


def flops_per_clip():
    """Algorithmic FLOPs (2 x MAC) per 2 s clip by kernel family."""
    P, P2, T = T_FRAMES * F_BINS, T_FRAMES * F2, T_FRAMES
    per_tok_ffn = 2 * (64 * 256 + 256 * 64)
    fl = {
        "conv_in": 2 * 3 * 64 * P,
        "conv_dense": 491520 * P + 2 * 491520 * P2,
        "conv_1x3": 2 * 3 * 64 * 64 * P2,
        "conv_subpixel": 2 * 2 * 3 * 64 * 128 * P2,
        "tail_proj": 2 * (64 * 2 + 64 * 4) * T * 2 * F2,
        "ffn": 8 * per_tok_ffn * P2,            # ff1 of the 8 conformers
        "ffn_post": 8 * per_tok_ffn * P2,       # ff2 (+ post-norm) of the 8 conformers
        "qkv": 8 * 2 * 64 * 192 * P2,
        "outproj": 8 * 2 * 64 * 64 * P2,
        "attn": 4 * 384 * (F2 * T * T + T * F2 * F2),
        "pw1glu": 8 * 2 * 64 * 256 * P2,
        "dwconv": 8 * 2 * 31 * 128 * P2,
        "pw2": 8 * 2 * 128 * 64 * P2,
        "dwpw2": 8 * 2 * (31 * 128 + 128 * 64) * P2,      # fused depthwise + pointwise (x3 mode)
    }
    return fl


def hbm_bytes_per_clip():
    """Algorithmic (compulsory) HBM bytes per 2 s clip by kernel family at the current fusion level:
    every kernel-boundary tensor written once and read once per consumer, fp32."""
    P, P2 = T_FRAMES * F_BINS, T_FRAMES * F2
    row = 64 * 4
    return {
        # dense blocks: layer i reads i slots and writes one; encoder at F, two decoders at F'
        "conv_dense": (10 + 4) * P * row + 2 * (10 + 4) * P2 * row,
        "attn": 8 * (3 * P2 * row + P2 * row),          # q, k, v images in, o out
        "ffn": 8 * 2 * P2 * row, "ffn_post": 8 * 3 * P2 * row,
        "qkv": 8 * 4 * P2 * row, "outproj": 8 * 3 * P2 * row,
        "pw1glu": 8 * 3 * P2 * row, "dwconv": 8 * 4 * P2 * row, "pw2": 8 * 4 * P2 * row,
        "dwpw2": 8 * 4 * P2 * row,                          # u in (2 rows of 64), x in, x out
        "stft_compress": 4 * CLIP_LEN + 8 * F_BINS * T_FRAMES,
    }


def cpu_baseline(sd, seconds_budget=20.0):
    """The oracle (CPU restatement of the reference path, torch fp32 on the host cores) timed on a
    bounded sample of the same workload: B=1 clips of the same shape, repeated.  torch's intra-op
    pool scales badly past a few dozen threads on these small ops, so a short sweep picks the
    thread count with the best throughput and the figure is quoted at that count."""
    from oracle import cmgan_oracle as O
    from oracle.weights import synthetic_clips
    wav = synthetic_clips(1, CLIP_LEN, seed=0)
    ncpu = os.cpu_count() or 1
    cands = sorted({c for c in (8, 16, 32, 64) if c <= ncpu} | ({ncpu} if ncpu <= 64 else set()))
    best, best_t = cands[0], float("inf")
    t_start = time.perf_counter()
    for c in cands:
        torch.set_num_threads(c)
        O.enhance_batch(sd, wav)                               # warm-up at this thread count
        t0 = time.perf_counter()
        O.enhance_batch(sd, wav)
        dt = time.perf_counter() - t0
        if dt < best_t:
            best, best_t = c, dt
        if time.perf_counter() - t_start > seconds_budget:
            break
    torch.set_num_threads(best)
    O.enhance_batch(sd, wav)
    n, t0 = 0, time.perf_counter()
    while True:
        O.enhance_batch(sd, wav)
        n += 1
        dt = time.perf_counter() - t0
        if dt >= seconds_budget * 0.5 or n >= 12:
            break
    return {"value": n * T_FRAMES / dt, "unit": "frames/s", "cores": best, "host_cpus": ncpu, "kind": "port",
            "sample": f"{n} x (B=1, 2 s clip, full pipeline wav->wav) at {best} threads (best of {cands}), "
                      f"{dt:.1f} s timed"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--mfma-mode", choices=["f16x3", "f32"], default="f16x3",
                    help="f16x3: fp32-accurate 3-term split products on the f16 matrix pipe (default); "
                         "f32: bit-exact fp32 MFMA")
    args = ap.parse_args()

    from cmgan_amd import TSCNet, dist as cdist
    from oracle.weights import make_state_dict, synthetic_clips

    rank, local, world = cdist.init_from_env()
    if world != args.gpus:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")

    sd = make_state_dict(seed=0)                               # random-init weights of the architecture
    model = TSCNet(64, F_BINS, device=dev, mfma_mode=args.mfma_mode).load_state_dict(sd).eval()
    eng = model.engine
    wav = synthetic_clips(BATCH_PER_GPU, CLIP_LEN, seed=rank).to(dev)    # resident in HBM before timing
    scal = torch.zeros(2, device=dev)

    run = eng.enhance if args.no_graph else eng.enhance_graphed

    def step():
        out = run(wav)
        # two per-step "loss" scalars (time-domain L1 / L2 against the input) and their single all-reduce
        d = out - wav
        scal[0] = d.abs().mean()
        scal[1] = (d * d).mean()
        cdist.allreduce_scalars(scal)
        return out

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    et = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        torch.distributed.all_reduce(et, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(et.item())

    # ---- per-kernel durations, measured live with HIP events on the launch stream ----------
    eng.set_profiling(True)
    agg = {}
    reps = 3
    for _ in range(reps):
        eng.enhance(wav)
        torch.cuda.synchronize()
        for name, ms in eng.profile():
            a = agg.setdefault(name, [0.0, 0])
            a[0] += ms
            a[1] += 1
    eng.set_profiling(False)

    if rank == 0:
        frames_per_step = world * BATCH_PER_GPU * T_FRAMES
        ms_per_step = 1e3 * elapsed / args.steps
        fl = flops_per_clip()
        total_flop = sum(v for k, v in fl.items() if k != "dwpw2") * BATCH_PER_GPU   # algorithmic, fusion-independent
        kern = {k: {"ms_per_step": v[0] / reps, "launches_per_step": v[1] // reps} for k, v in agg.items()}
        dom = max((k for k in kern if k in fl), key=lambda k: kern[k]["ms_per_step"])
        x3 = args.mfma_mode == "f16x3"
        peak_tf = F16_MFMA_PEAK_TF if x3 else FP32_MFMA_PEAK_TF
        dom_s = kern[dom]["ms_per_step"] * 1e-3
        dom_tf = fl[dom] * BATCH_PER_GPU / dom_s / 1e12
        roof = {"bound": "mfma", "kernel": dom, "achieved": round(dom_tf, 2), "peak": peak_tf,
                "unit": "TFLOP/s", "frac": round(dom_tf / peak_tf, 4), "traffic": None,
                "launches_per_step": kern[dom]["launches_per_step"],
                "avg_launch_ms": round(kern[dom]["ms_per_step"] / max(1, kern[dom]["launches_per_step"]), 4),
                "note": ("algorithmic FLOPs; the f16x3 mode issues 3 MFMA products per algorithmic product, so the "
                         "matrix pipe is doing 3x this" if x3 else "exact fp32 MFMA")}
        # HBM bytes per launch from the committed PMC passes of this same command (profiles/README.md);
        # the counters cannot be read from inside the process, so this is the one field not measured live
        tsrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles",
                            "r01_x3_hbm_traffic.json" if x3 else "r01_fp32_hbm_traffic.json")
        pmc_traffic = {}
        if os.path.exists(tsrc):
            with open(tsrc) as f:
                pmc_traffic = json.load(f).get("per_launch", {})
        if dom in pmc_traffic and BATCH_PER_GPU == 32:
            roof["traffic"] = pmc_traffic[dom]["hbm_bytes"]
            roof["traffic_note"] = ("HBM bytes per launch, rocprofv3 --pmc FETCH_SIZE (x2 gfx950 correction) + WRITE_SIZE, "
                                    "from profiles/" + os.path.basename(tsrc))
            roof["algorithmic_hbm_bytes_per_launch"] = round(
                hbm_bytes_per_clip().get(dom, 0) * BATCH_PER_GPU / max(1, kern[dom]["launches_per_step"]))
        hb = hbm_bytes_per_clip()
        extra = {}
        # the same figure for every MFMA kernel family (the dominant one changes with a few % of run-to-run noise:
        # attention and the dense convs are within 3 % of each other)
        extra["roofline_by_kernel"] = {
            k: {"ms_per_step": round(kern[k]["ms_per_step"], 4),
                "achieved_tflops": round(fl[k] * BATCH_PER_GPU / (kern[k]["ms_per_step"] * 1e-3) / 1e12, 2),
                "frac_of_mfma_peak": round(fl[k] * BATCH_PER_GPU / (kern[k]["ms_per_step"] * 1e-3) / 1e12 / peak_tf, 4),
                "hbm_bytes_per_launch_pmc": pmc_traffic.get(k, {}).get("hbm_bytes")}
            for k in sorted((k for k in kern if k in fl and fl[k] > 0), key=lambda k: -kern[k]["ms_per_step"])[:5]}
        if dom in hb:
            gbs = hb[dom] * BATCH_PER_GPU / dom_s / 1e9
            extra["roofline_hbm"] = {"bound": "hbm", "kernel": dom, "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
                                     "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
                                     "note": "algorithmic bytes (each boundary tensor written once, read once per consumer)"}
        if "stft_compress" in kern:
            gbs = hb["stft_compress"] * BATCH_PER_GPU / (kern["stft_compress"]["ms_per_step"] * 1e-3) / 1e9
            extra["stft_hbm"] = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": round(gbs / HBM_PEAK_GBS, 4)}
        line = {
            "metric": "enhanced audio frames/sec (16 kHz, 2 s clips, batch 32 per GPU)",
            "value": frames_per_step * args.steps / elapsed,
            "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 storage/accumulate; products as 3 x f16 split MFMA (fp32-class accuracy)" if x3 else "f32",
            "data": "synthetic",
            "config": {"workload": "configs[1]: batch=32 x 2 s synthetic 16 kHz noisy clips per GPU, n_fft=400 "
                                   "hop=100, TSCNet(64,201) random-init, full pipeline wav->wav",
                       "mfma_mode": args.mfma_mode, "launch": "eager" if args.no_graph else "hipGraph replay",
                       "batch_per_gpu": BATCH_PER_GPU, "global_batch": world * BATCH_PER_GPU,
                       "frames_per_clip": T_FRAMES, "parallelism": f"dp{world}"},
            "path_tflops": round(total_flop / (ms_per_step * 1e-3) / 1e12, 2),
            "path_frac_of_mfma_peak": round(total_flop / (ms_per_step * 1e-3) / 1e12 / peak_tf, 4),
            "roofline": roof,
            "kernels_ms_per_step": {k: round(v["ms_per_step"], 4) for k, v in sorted(kern.items(),
                                                                                   key=lambda kv: -kv[1]["ms_per_step"])},
        }
        line.update(extra)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(sd)
        print(json.dumps(line), flush=True)

    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
