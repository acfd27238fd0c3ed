#!/usr/bin/env python3
"""Benchmark of the CMGAN generator forward path on MI355X (BASELINE.json metric:
enhanced audio frames/sec, 16 kHz, 2 s clips, batch 32 per GPU).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload 16k|48k]

A "step" = one pass of the whole device pipeline (RMS scale -> STFT -> power compress ->
TSCNet -> power uncompress -> ISTFT) over one batch of synthetic 2 s clips per GPU that are
already resident in HBM, followed by the validation-step loss scalars (a HIP reduction from
the library, src/train.py:139-141) and ONE all-reduce of them (RCCL over xGMI), as north_star
specifies.  Utterances are sharded by rank (weak scaling: 32 clips per GPU); the forward
needs no collective.

`--gpus N` launches the N ranks itself (one process per GPU under torch.distributed.run, like
the reference's mp.spawn in src/train.py:294-297) unless it is already running as a rank of
such a launch (WORLD_SIZE set by the driver); a world size that differs from --gpus, or fewer
visible GPUs than requested, is a hard error - never a silent 1-GPU run.  Rank 0 prints one
JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

FP32_MFMA_PEAK_TF = 157.3                                      # MI355X_MICROARCH.md (f32 in / f32 acc MFMA)
F16_MFMA_PEAK_TF = 2500.0                                      # dense f16/bf16 MFMA (not the 2:1-sparse headline)
HBM_PEAK_GBS = 8000.0

# BASELINE.json configs: [1] is the metric's workload; [3] (48 kHz) is measured with --workload 48k
WORKLOADS = {
    "16k": dict(n_fft=400, hop=100, clip_len=32000, batch=32,            # src/train.py:22,47-48,53
                name="configs[1]: batch=32 x 2 s synthetic 16 kHz noisy clips per GPU, n_fft=400 hop=100, "
                     "TSCNet(64,201) random-init, full pipeline wav->wav",
                metric="enhanced audio frames/sec (16 kHz, 2 s clips, batch 32 per GPU)"),
    "48k": dict(n_fft=1200, hop=300, clip_len=96000, batch=8,
                name="configs[3]: batch=8 x 2 s synthetic 48 kHz clips per GPU, n_fft=1200 hop=300, "
                     "TSCNet(64,601) random-init, full pipeline wav->wav",
                metric="enhanced audio frames/sec (48 kHz super-wideband variant, 2 s clips, batch 8 per GPU)"),
}


class Shape:
    def __init__(self, wl):
        self.n_fft, self.hop, self.L, self.B = wl["n_fft"], wl["hop"], wl["clip_len"], wl["batch"]
        self.F = self.n_fft // 2 + 1
        self.F2 = (self.F + 1) // 2
        self.T = self.L // self.hop + 1                                    # 321 (src/train.py:53)


def flops_per_clip(sh: Shape):
    """Algorithmic FLOPs (2 x MAC) per clip by kernel family."""
    T, P, P2, F2 = sh.T, sh.T * sh.F, sh.T * sh.F2, sh.F2
    per_tok_ffn = 2 * (64 * 256 + 256 * 64)
    return {
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
        "attn_out": 4 * 384 * (F2 * T * T + T * F2 * F2) + 8 * 2 * 64 * 64 * P2,   # attention + to_out fused
        "pw1glu": 8 * 2 * 64 * 256 * P2,
        "dwconv": 8 * 2 * 31 * 128 * P2,
        "pw2": 8 * 2 * 128 * 64 * P2,
        "dwpw2": 8 * 2 * (31 * 128 + 128 * 64) * P2,      # fused depthwise + pointwise (x3 mode)
    }


# families that are the same arithmetic under a different fusion: counted once in the path total
_FUSED_ALIASES = {"dwpw2": ("dwconv", "pw2"), "attn_out": ("attn", "outproj")}


def path_flops_per_clip(sh: Shape) -> float:
    fl = flops_per_clip(sh)
    return float(sum(v for k, v in fl.items() if k not in _FUSED_ALIASES))


def hbm_bytes_per_clip(sh: Shape):
    """Algorithmic (compulsory) HBM bytes per clip by kernel family at the current fusion level:
    every kernel-boundary tensor written once and read once per consumer, fp32."""
    P, P2 = sh.T * sh.F, sh.T * sh.F2
    row = 64 * 4
    return {
        # dense blocks: layer i reads i slots and writes one; encoder at F, two decoders at F'
        "conv_dense": (10 + 4) * P * row + 2 * (10 + 4) * P2 * row,
        "attn": 8 * (3 * P2 * row + P2 * row),          # q, k, v images in, o out
        "attn_out": 8 * (3 * P2 * row + 2 * P2 * row),  # q, k, v in; residual read + write (o stays on chip)
        "ffn": 8 * 2 * P2 * row, "ffn_post": 8 * 3 * P2 * row,
        "qkv": 8 * 4 * P2 * row, "outproj": 8 * 3 * P2 * row,
        "pw1glu": 8 * 3 * P2 * row, "dwconv": 8 * 4 * P2 * row, "pw2": 8 * 4 * P2 * row,
        "dwpw2": 8 * 4 * P2 * row,                          # u in (2 rows of 64), x in, x out
        "stft_compress": 4 * sh.L + 8 * sh.F * sh.T,
    }


def csrc_digest() -> str:
    from cmgan_amd import build as _build
    return _build.inference_digest()[:16]


def cpu_baseline(sd, sh: Shape, seconds_budget=45.0):
    """CPU baseline on the GPU box's host cores, on a bounded sample of the same workload (B = 1 clips of the same
    shape: thread-count sweep over ALL host cores, then repeats at the best count; plus one B = 4 point).
    kind = "reference": the reference's OWN modules (oracle/_ref: TSCNet / power_compress / power_uncompress
    byte-compiled from /root/reference by oracle/make_ref.py, driven through the evaluation.py glue by
    oracle/ref_runner.py), with the oracle port's figure beside it.  kind = "port" (the oracle,
    oracle/cmgan_oracle.py) only when oracle/_ref has not been built."""
    from oracle import cmgan_oracle as O
    from oracle import ref_runner as R
    from cmgan_amd.synth import synthetic_clips
    wav = synthetic_clips(1, sh.L, seed=0)
    wav4 = synthetic_clips(4, sh.L, seed=1)
    ncpu = os.cpu_count() or 1
    use_ref = R.available()
    if use_ref:
        model = R.tscnet(sd, sh.F)
        run = lambda w: R.enhance_batch(model, w, sh.n_fft, sh.hop)
    else:
        run = lambda w: O.enhance_batch(sd, w, sh.n_fft, sh.hop)
    cands = sorted({c for c in (8, 16, 32, 64, 128, 256) if c < ncpu} | {ncpu})
    sweep, best, best_t = {}, cands[0], float("inf")
    t_start = time.perf_counter()
    for c in cands:
        torch.set_num_threads(c)
        run(wav)                                               # warm-up at this thread count
        t0 = time.perf_counter()
        run(wav)
        dt = time.perf_counter() - t0
        sweep[c] = round(sh.T / dt, 1)
        if dt < best_t:
            best, best_t = c, dt
        if time.perf_counter() - t_start > seconds_budget * 0.6:      # (every candidate fits on the 256-CPU box: ~4 s each)
            break
    torch.set_num_threads(best)
    run(wav)
    n, t0 = 0, time.perf_counter()
    while True:
        run(wav)
        n += 1
        dt = time.perf_counter() - t0
        if dt >= seconds_budget * 0.2 or n >= 12:
            break
    t4 = time.perf_counter()
    run(wav4)
    dt4 = time.perf_counter() - t4
    out = {"value": n * sh.T / dt, "unit": "frames/s", "cores": best, "host_cpus": ncpu,
           "kind": "reference" if use_ref else "port",
           "sample": f"{n} x (B=1, 2 s clip, full pipeline wav->wav) at {best} threads (best of the sweep), "
                     f"{dt:.1f} s timed",
           "thread_sweep_frames_per_s": sweep,
           "b4": {"value": 4 * sh.T / dt4, "unit": "frames/s", "cores": best,
                  "sample": f"1 x (B=4, 2 s clips) at {best} threads, {dt4:.1f} s"}}
    if use_ref:
        O.enhance_batch(sd, wav, sh.n_fft, sh.hop)
        tp = time.perf_counter()
        got = O.enhance_batch(sd, wav, sh.n_fft, sh.hop)
        dtp = time.perf_counter() - tp
        want = run(wav)
        out["port"] = {"value": sh.T / dtp, "unit": "frames/s", "cores": best,
                       "sample": f"1 x (B=1) of oracle/cmgan_oracle.py at {best} threads, {dtp:.1f} s",
                       "rel_err_vs_reference": float((got - want).abs().max() / want.abs().max())}
        out["kind_note"] = ("reference = the reference repo's own models/generator.py + models/conformer.py + utils.py "
                            "(bytecode in oracle/_ref, built from /root/reference by oracle/make_ref.py) behind the "
                            "src/evaluation.py:21-53 glue; port = the oracle restatement, same inputs, same run")
    else:
        out["kind_note"] = "port = oracle/cmgan_oracle.py; oracle/_ref (the reference's own modules) was not built"
    return out


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args) -> int:
    """One process per GPU on this node (the reference: mp.spawn(main, nprocs=#GPUs), src/train.py:294-297)."""
    if not args.stub_cpu:
        have = torch.cuda.device_count()
        if have < args.gpus:
            print(f"bench.py: --gpus {args.gpus} requested but only {have} GPU(s) are visible; refusing to fall "
                  "back to fewer ranks", file=sys.stderr)
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC for RCCL on this host driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.call(cmd, env=env)


def kernel_table(eng, wav, reps=3):
    """Per-kernel durations of one forward, measured live with HIP events on the launch stream."""
    eng.set_profiling(True)
    agg = {}
    for _ in range(reps):
        eng.enhance(wav)
        torch.cuda.synchronize()
        for name, ms in eng.profile():
            a = agg.setdefault(name, [0.0, 0])
            a[0] += ms
            a[1] += 1
    eng.set_profiling(False)
    return {k: {"ms_per_step": v[0] / reps, "launches_per_step": v[1] // reps} for k, v in agg.items()}


def roofline_of(kern, sh: Shape, x3: bool, pmc_traffic=None, traffic_src=None):
    fl = flops_per_clip(sh)
    hb = hbm_bytes_per_clip(sh)
    peak_tf = F16_MFMA_PEAK_TF if x3 else FP32_MFMA_PEAK_TF
    dom = max((k for k in kern if k in fl), key=lambda k: kern[k]["ms_per_step"])
    dom_s = kern[dom]["ms_per_step"] * 1e-3
    dom_tf = fl[dom] * sh.B / dom_s / 1e12
    n_launch = max(1, kern[dom]["launches_per_step"])
    roof = {"bound": "mfma", "kernel": dom, "achieved": round(dom_tf, 2), "peak": peak_tf,
            "unit": "TFLOP/s", "frac": round(dom_tf / peak_tf, 4), "traffic": None,
            "launches_per_step": kern[dom]["launches_per_step"],
            "avg_launch_ms": round(kern[dom]["ms_per_step"] / n_launch, 4),
            "note": ("algorithmic FLOPs; the f16x3 mode issues 3 MFMA products per algorithmic product, so the "
                     "matrix pipe is doing 3x this" if x3 else "exact fp32 MFMA")}
    if dom in hb:
        roof["algorithmic_hbm_bytes_per_launch"] = round(hb[dom] * sh.B / n_launch)
    if pmc_traffic is not None and dom in pmc_traffic:
        roof["traffic"] = pmc_traffic[dom]["hbm_bytes"]
        roof["traffic_note"] = ("HBM bytes per launch, rocprofv3 --pmc FETCH_SIZE (x2 gfx950 correction) + WRITE_SIZE, "
                                "from " + traffic_src)
    return roof, dom


def load_pmc_traffic(x3: bool, sh: Shape, workload: str):
    """HBM bytes per launch from the committed PMC passes of this same command (the counters cannot be read from
    inside the process).  Only used when the file was produced from EXACTLY the kernel sources now built
    (csrc digest match) at the same workload: a kernel edit without a profile refresh publishes null, not
    stale counters."""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{'x3' if x3 else 'fp32'}_hbm_traffic.json")))
    for path in reversed(cands):
        with open(path) as f:
            d = json.load(f)
        if d.get("csrc_digest") == csrc_digest() and d.get("workload", "16k") == workload and \
                d.get("batch", 32) == sh.B:
            return d.get("per_launch", {}), "profiles/" + os.path.basename(path), None
    why = "no profiles/*_hbm_traffic.json matches the built kernel sources (digest %s)" % csrc_digest()
    return None, None, why


def train_leg(eng, sd, dev, rank, world, barrier, batch=32, cut_len=32000, steps=5):
    """BASELINE configs[2] at its own size: "batch=256 x 2 s clips data-parallel across 8 GPUs" = 32 clips per GPU and
    step.  One data-parallel adversarial training step per rank (cmgan_amd.training.adversarial_train_step: generator
    on RI + magnitude + time + GAN loss, metric discriminator on given PESQ labels; per step TWO gradient all-reduces
    over the flat buckets - 7.3 MB generator, 0.7 MB discriminator - and two AdamW launches).  Timed like the main leg:
    barrier + synchronize on both sides, MAX over ranks.  Any failure is reported in the line instead of losing it."""
    # phase 1 - no collectives: build both networks on every rank, then agree (one MIN all-reduce) that all of them
    # succeeded before any rank enters a step with gradient all-reduces, so a rank-local failure cannot strand the others
    err = None
    try:
        from cmgan_amd.synth import discriminator_state_dict, synthetic_clips
        from cmgan_amd.training import AdamW, DiscriminatorTrain, GeneratorTrain, adversarial_train_step
        gen = GeneratorTrain(sd, engine=eng)
        disc = DiscriminatorTrain(discriminator_state_dict(0), engine=eng)
        opt_g = AdamW(eng, gen.param_bucket, gen.grad_bucket, lr=5e-4)
        opt_d = AdamW(eng, disc.param_bucket, disc.grad_bucket, lr=1e-3)
        clean = synthetic_clips(batch, cut_len, seed=2000 + rank).to(dev)
        noisy = (clean + 0.3 * synthetic_clips(batch, cut_len, seed=3000 + rank).to(dev)).contiguous()
        pesq = torch.full((batch,), 0.5, device=dev)
        tgen = torch.Generator(device=dev).manual_seed(rank)
        run = lambda: adversarial_train_step(gen, disc, opt_g, opt_d, clean, noisy, pesq, generator=tgen)
    except Exception as e:                                      # noqa: BLE001 - the headline line must survive
        err = f"{type(e).__name__}: {e}"[:300]
    ready = torch.tensor([0.0 if err else 1.0], device=dev)
    if world > 1:
        torch.distributed.all_reduce(ready, op=torch.distributed.ReduceOp.MIN)
    if float(ready) < 1.0:
        return {"error": err or "another rank failed to set the training leg up"}
    try:
        torch.cuda.reset_peak_memory_stats(dev)
        for _ in range(2):                                      # untimed: the first steps still grow the allocator's pools and
            run()                                               # load kernels (step 2 measured 1.8 x a steady-state step)
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss, _, gan, loss_d = run()
        torch.cuda.synchronize()
        barrier()
        dt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        if world > 1:
            torch.distributed.all_reduce(dt, op=torch.distributed.ReduceOp.MAX)
        ms = 1e3 * float(dt) / steps
        return {"workload": f"configs[2]: adversarial train step, {batch} x {cut_len / 16000:g} s clips per GPU, dropout on, "
                            "TSCNet(64,201) + Discriminator(16) random-init, synthetic PESQ labels",
                "batch_per_gpu": batch, "global_batch": batch * world, "steps": steps, "warmup": 2, "ms_per_step": round(ms, 2),
                "clips_per_s": round(batch * world / (ms * 1e-3), 2), "dtype": "f32 storage / accumulate; every conformer kernel except to_out, the dense convs and all weight gradients as split-f16 MFMA products (fp32-class, exact power-of-two gradient scaling); the encoder's / decoders' 1 x 3 convs likewise since round 6; tail convs, conv_1 and the discriminator fp32",
                "peak_device_GB": round(torch.cuda.max_memory_allocated(dev) / 2**30, 1),
                "collectives_per_step": "2 all-reduces over flat buckets (generator %.1f MB, discriminator %.1f MB)"
                                        % (gen.grad_bucket.numel * 4 / 2**20, disc.grad_bucket.numel * 4 / 2**20),
                "loss": round(float(loss), 4), "gen_loss_GAN": round(float(gan), 4), "disc_loss": round(float(loss_d), 4)}
    except Exception as e:                                      # noqa: BLE001 - the headline line must survive
        return {"error": f"{type(e).__name__}: {e}"[:300]}


def stream_leg(model, dev, reps=5):
    """BASELINE configs[4]: one 10 s 16 kHz clip in 400-frame windows, each step replayed from a captured hipGraph.
    `carried_state_*` = cmgan_amd.streaming.enhance_stream (by default the decoders of step k run on a second
    stream beside the encoder / TSCBs of step k + 1, `_not_pipelined` = one graph per step): frozen InstanceNorm statistics make the dense
    encoder and both decoders exactly causal, so their state is CARRIED (15 frames of input history, every frame
    computed once) and only the four TSCBs - bidirectional attention - run on [40 cached context | 400 | 40 look-ahead]
    frames.  `windows_*` = the stateless per-window contract (enhance_windows: 40 frames of context recomputed on each
    side by every stage)."""
    from cmgan_amd.streaming import enhance_stream, enhance_windows
    from cmgan_amd.synth import synthetic_clips
    noisy = synthetic_clips(1, 160000, seed=3).to(dev)
    stats = model.engine.tscnet_forward_stats(model.engine.stft_compress(noisy[:, :44000], model.engine.rms_scale(noisy)))[2]
    out = {}
    legs = (("carried_state_1_window_per_replay", lambda: enhance_stream(model, noisy, 400, 40, 40, stats=stats, graph=True)),
            ("carried_state_1_window_per_replay_not_pipelined",
             lambda: enhance_stream(model, noisy, 400, 40, 40, stats=stats, graph=True, pipeline=False)),
            ("carried_state_no_lookahead", lambda: enhance_stream(model, noisy, 400, 40, 0, stats=stats, graph=True)),
            ("windows_per_replay_1", lambda: enhance_windows(model, noisy, 40000, 4000, graph=True, batch=1)),
            ("windows_per_replay_4", lambda: enhance_windows(model, noisy, 40000, 4000, graph=True, batch=4)),
            ("windows_per_replay_1_no_lookahead", lambda: enhance_windows(model, noisy, 40000, 4000, graph=True, batch=1, lookahead=0)))
    for name, fn in legs:
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        out[name] = {"ms_per_10s_clip": round(1e3 * dt, 3), "frames_per_s": round(1601 / dt, 1),
                     "real_time_factor": round(10.0 / dt, 1)}
    # What the carried-state contract costs in accuracy: frozen statistics + windowed attention is a NEW numerical contract
    # (the reference has none: src/evaluation.py:30-34 enhances whole rows), so the streamed output is compared with the
    # WHOLE-CLIP forward of the same library on the same clip (cmgan_enhance, which equals the reference's own modules
    # within 3e-6: tests/test_gpu_parity.py; the GPU suite repeats the comparison against oracle/_ref directly).
    # SSNR / STOI use the whole-clip output as the reference signal (cmgan_amd.metrics, the reference tool's arithmetic).
    import numpy as np
    from cmgan_amd import metrics as M

    def cost(got, whole):
        a, ref = got.double().cpu().numpy(), whole.double().cpu().numpy()
        d = a - ref
        ss = M.segmental_snr(ref, a, 16000)
        ss = ss[1] if isinstance(ss, tuple) else ss
        return {"rel_err_max": float(f"{np.abs(d).max() / np.abs(ref).max():.3e}"),
                "rel_err_rms": float(f"{np.sqrt((d * d).mean() / (ref * ref).mean()):.3e}"),
                "ssnr_db_vs_whole_clip": round(float(np.mean(ss)), 2), "stoi_vs_whole_clip": round(float(M.stoi(ref, a, 16000)), 5)}

    try:
        acc = {}
        clips = [("synthetic_10s", noisy, 400)]
        tr = os.path.join(ROOT, "tests", "golden", "tracks.npz")
        if os.path.exists(tr):
            g = np.load(tr)
            for name in ("a", "b", "silence"):                     # the reference repo's AudioSamples recordings (2.1 s each:
                pcm = torch.from_numpy(g["pcm_" + name].astype(np.float32) / 32768.0)[None]     # one 400-frame window would
                clips.append(("track_" + name, pcm[:, :pcm.size(1) // 100 * 100].contiguous().to(dev), 100))   # be the whole clip)
        for name, clip, w in clips:
            whole = model.engine.enhance(clip)[0]
            for ca, la in ((40, 40), (40, 0)):
                got = enhance_stream(model, clip, w, ca, la, graph=True)
                acc[f"{name}_window{w}_context{ca}_lookahead{la}"] = cost(got, whole)
        approx = {"reference": "whole-clip cmgan_enhance of the same clip (= the reference's TSCNet on the whole clip, 3e-6)",
                  "statistics": "calibrated on the clip's first window + look-ahead frames, then frozen", "cases": acc}
    except Exception as e:                                      # noqa: BLE001 - the headline line must survive
        approx = {"error": f"{type(e).__name__}: {e}"[:300]}
    return {"approximation_cost_vs_whole_clip": approx,
            "workload": "configs[4]: 10 s 16 kHz clip, 400-frame windows, 40 frames of attention context + 40 of "
                        "look-ahead; carried_state = encoder / decoder state carried under frozen InstanceNorm statistics "
                        "(exact), TSCB context from cached encoder outputs; windows = context recomputed by every stage",
            "results": out}


def workload_48k_leg(dev, mfma_mode, steps=5):
    """BASELINE configs[3]: the 48 kHz super-wideband variant (n_fft 1200, F = 601) at batch 8, wav -> wav."""
    from cmgan_amd import TSCNet
    from cmgan_amd.synth import make_state_dict, synthetic_clips
    sh = Shape(WORKLOADS["48k"])
    m = TSCNet(64, sh.F, n_fft=sh.n_fft, hop=sh.hop, device=dev, mfma_mode=mfma_mode)
    m.load_state_dict(make_state_dict(seed=0, num_features=sh.F)).eval()
    wav = synthetic_clips(sh.B, sh.L, seed=7).to(dev)
    for _ in range(2):
        m.engine.enhance_graphed(wav)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        m.engine.enhance_graphed(wav)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"workload": WORKLOADS["48k"]["name"], "ms_per_step": round(1e3 * dt, 3), "steps": steps,
            "value": round(sh.B * sh.T / dt, 1), "unit": "frames/s", "frames_per_clip": sh.T, "launch": "hipGraph replay"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="16k")
    ap.add_argument("--batch", type=int, default=None, help="clips per GPU (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-f16x1", action="store_true",
                    help="skip the reduced-precision (single fp16 product) mode leg of the line")
    ap.add_argument("--no-f32", action="store_true", help="skip the bit-exact fp32-MFMA mode leg of the line")
    ap.add_argument("--no-train", action="store_true",
                    help="skip the training-step leg (BASELINE configs[2]: adversarial train steps at 32 clips per GPU "
                         "with the gradient all-reduces over the flat buckets)")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the configs[3] (48 kHz) and configs[4] (10 s clip in 400-frame windows) legs of the line")
    ap.add_argument("--train-batch", type=int, default=32, help="clips per GPU of the training-step leg (configs[2]: 32)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--mfma-mode", choices=["f16x3", "f32"], default="f16x3",
                    help="f16x3: fp32-accurate 3-term split products on the f16 matrix pipe (default); "
                         "f32: bit-exact fp32 MFMA")
    ap.add_argument("--stub-cpu", action="store_true",
                    help="launcher / collective self-test without a GPU: gloo backend, the forward replaced by a "
                         "trivial CPU op (tests/test_bench_launcher.py); prints a line marked \"stub\": true")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))

    from cmgan_amd import dist as cdist
    from cmgan_amd.synth import make_state_dict, synthetic_clips

    rank, local, world = cdist.init_from_env("gloo" if args.stub_cpu else None)
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launch has WORLD_SIZE={world}; refusing to mislabel the run",
              file=sys.stderr)
        sys.exit(2)

    wl = dict(WORKLOADS[args.workload])
    if args.batch:
        wl["batch"] = args.batch
    sh = Shape(wl)
    x3 = args.mfma_mode == "f16x3"

    if args.stub_cpu:
        dev = torch.device("cpu")
        wav = synthetic_clips(2, 1600, seed=rank)
        clean = synthetic_clips(2, 1600, seed=100 + rank)
        scal = torch.zeros(4)

        def step():
            out = wav * 0.5
            d = out - clean
            scal[2], scal[3] = d.abs().mean(), (d * d).mean()
            cdist.allreduce_scalars(scal[2:4])
            return out

        def sync():
            pass
    else:
        from cmgan_amd import TSCNet
        if torch.cuda.device_count() <= local:
            print(f"bench.py: rank {rank} needs cuda:{local} but only {torch.cuda.device_count()} GPU(s) are visible",
                  file=sys.stderr)
            sys.exit(2)
        torch.cuda.set_device(local)
        dev = torch.device(f"cuda:{local}")
        sd = make_state_dict(seed=0, num_features=sh.F)        # random-init weights of the architecture
        model = TSCNet(64, sh.F, n_fft=sh.n_fft, hop=sh.hop, device=dev, mfma_mode=args.mfma_mode)
        model.load_state_dict(sd).eval()
        eng = model.engine
        wav = synthetic_clips(sh.B, sh.L, seed=rank).to(dev)            # resident in HBM before timing
        clean = synthetic_clips(sh.B, sh.L, seed=1000 + rank).to(dev)   # validation target (src/train.py:207-220)
        scal = torch.zeros(4, device=dev)
        run = eng.enhance if args.no_graph else eng.enhance_graphed

        def step():
            out = run(wav)
            # the validation step's time-domain loss scalars (train.py:139-141): one HIP reduction from the
            # library, then the path's single collective
            eng.loss_terms(est_audio=out, clean_audio=clean, out=scal)
            cdist.allreduce_scalars(scal[2:4])
            return out

        def sync():
            torch.cuda.synchronize()

    def barrier():
        if world > 1 or cdist._active():
            torch.distributed.barrier()

    # did the collective really span `world` ranks?
    ones = torch.ones(1, device=dev)
    cdist.allreduce_scalars(ones)
    ranks_seen = int(round(float(ones.item())))
    backend = torch.distributed.get_backend() if cdist._active() else "none (single process)"

    for _ in range(args.warmup):
        step()
    sync()
    barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    own = time.perf_counter() - t0                            # this rank's K steps
    barrier()
    sync()
    elapsed = time.perf_counter() - t0
    et = torch.tensor([elapsed, own], device=dev, dtype=torch.float64)
    per_rank = [et.clone() for _ in range(world)]
    if world > 1 or cdist._active():
        torch.distributed.all_gather(per_rank, et)
    elapsed = max(float(t[0]) for t in per_rank)              # MAX over ranks
    unit_frames = sh.B * sh.T if not args.stub_cpu else 2 * 17
    per_rank_fps = [unit_frames * args.steps / float(t[1]) for t in per_rank]
    loss_scalars = [float(v) / world for v in scal[2:4].tolist()]

    line = None
    if rank == 0:
        frames_per_step = world * unit_frames
        ms_per_step = 1e3 * elapsed / args.steps
        value = frames_per_step * args.steps / elapsed
        line = {
            "metric": wl["metric"],
            "value": value,
            "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 storage/accumulate; products as 3 x f16 split MFMA (fp32-class accuracy)" if x3 else "f32",
            "data": "synthetic",
            "config": {"workload": wl["name"], "mfma_mode": args.mfma_mode,
                       "launch": "eager" if args.no_graph else "hipGraph replay",
                       "batch_per_gpu": sh.B, "global_batch": world * sh.B,
                       "frames_per_clip": sh.T, "parallelism": f"dp{world}"},
            "per_rank_frames_per_s": [round(v, 1) for v in per_rank_fps],
            "scaling_efficiency_vs_rank_sum": round(value / sum(per_rank_fps), 4),
            "collective": {"backend": ("nccl = RCCL" if backend == "nccl" else backend), "ranks_seen": ranks_seen,
                           "per_step": "1 all-reduce of 2 fp32 loss scalars", "loss_scalars_mean": loss_scalars},
        }
        if args.stub_cpu:
            line["stub"] = True
            line["config"]["workload"] = "launcher self-test (CPU stub, gloo)"

    if not args.stub_cpu:
        kern = kernel_table(eng, wav)
        if rank == 0:
            total_flop = path_flops_per_clip(sh) * sh.B                  # algorithmic, fusion-independent
            peak_tf = F16_MFMA_PEAK_TF if x3 else FP32_MFMA_PEAK_TF
            pmc, src, why = load_pmc_traffic(x3, sh, args.workload)
            roof, dom = roofline_of(kern, sh, x3, pmc, src)
            if pmc is None:
                roof["traffic_note"] = why
            fl, hb = flops_per_clip(sh), hbm_bytes_per_clip(sh)
            line["csrc_digest"] = csrc_digest()
            line["path_tflops"] = round(total_flop / (ms_per_step * 1e-3) / 1e12, 2)
            line["path_frac_of_mfma_peak"] = round(total_flop / (ms_per_step * 1e-3) / 1e12 / peak_tf, 4)
            line["roofline"] = roof
            line["kernels_ms_per_step"] = {k: round(v["ms_per_step"], 4)
                                           for k, v in sorted(kern.items(), key=lambda kv: -kv[1]["ms_per_step"])}
            # the same figure for every MFMA kernel family (the top two are within a few % of each other)
            line["roofline_by_kernel"] = {
                k: {"ms_per_step": round(kern[k]["ms_per_step"], 4),
                    "achieved_tflops": round(fl[k] * sh.B / (kern[k]["ms_per_step"] * 1e-3) / 1e12, 2),
                    "frac_of_mfma_peak": round(fl[k] * sh.B / (kern[k]["ms_per_step"] * 1e-3) / 1e12 / peak_tf, 4),
                    "hbm_bytes_per_launch_pmc": (pmc or {}).get(k, {}).get("hbm_bytes")}
                for k in sorted((k for k in kern if k in fl and fl[k] > 0), key=lambda k: -kern[k]["ms_per_step"])[:5]}
            if dom in hb:
                gbs = hb[dom] * sh.B / (kern[dom]["ms_per_step"] * 1e-3) / 1e9
                line["roofline_hbm"] = {"bound": "hbm", "kernel": dom, "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
                                        "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
                                        "note": "algorithmic bytes (each boundary tensor written once, read once per consumer)"}
            if "stft_compress" in kern:
                gbs = hb["stft_compress"] * sh.B / (kern["stft_compress"]["ms_per_step"] * 1e-3) / 1e9
                line["stft_hbm"] = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": round(gbs / HBM_PEAK_GBS, 4)}

        # ---- the bit-exact fp32-MFMA mode of the same library, in the same line (N = 1 only) -------------
        if world == 1 and x3 and not args.no_f32:
            eng._graphs.clear()
            m32 = TSCNet(64, sh.F, n_fft=sh.n_fft, hop=sh.hop, device=dev, mfma_mode="f32").load_state_dict(sd).eval()
            e32 = m32.engine
            k32 = max(2, min(args.steps, 5))
            e32.enhance(wav)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(k32):
                e32.enhance(wav)
            torch.cuda.synchronize()
            dt32 = (time.perf_counter() - t0) / k32
            kern32 = kernel_table(e32, wav, reps=1)
            roof32, _ = roofline_of(kern32, sh, False)
            line["f32_mode"] = {"ms_per_step": round(1e3 * dt32, 3), "value": round(sh.B * sh.T / dt32, 1),
                                "unit": "frames/s", "steps": k32, "launch": "eager", "dtype": "f32 (exact fp32 MFMA products)",
                                "path_frac_of_fp32_mfma_peak": round(path_flops_per_clip(sh) * sh.B / dt32 / 1e12 /
                                                                     FP32_MFMA_PEAK_TF, 4),
                                "roofline": roof32}
            del m32, e32
        # ---- the opt-in REDUCED-precision mode (BASELINE configs[1] "bf16": one fp16 product per contraction in the
        # TSCNet body), same workload, same line; never the headline: its error is reported next to its time ----
        if world == 1 and x3 and not args.no_f16x1:
            ref_out = run(wav).clone()
            m1 = TSCNet(64, sh.F, n_fft=sh.n_fft, hop=sh.hop, device=dev, mfma_mode="f16x1").load_state_dict(sd).eval()
            e1 = m1.engine
            run1 = e1.enhance if args.no_graph else e1.enhance_graphed
            out1 = run1(wav)
            torch.cuda.synchronize()
            k1 = max(2, min(args.steps, 10))
            t0 = time.perf_counter()
            for _ in range(k1):
                run1(wav)
            torch.cuda.synchronize()
            dt1 = (time.perf_counter() - t0) / k1
            err = float((out1 - ref_out).abs().max() / ref_out.abs().max())
            line["f16x1_mode"] = {"ms_per_step": round(1e3 * dt1, 3), "value": round(sh.B * sh.T / dt1, 1),
                                  "unit": "frames/s", "steps": k1, "dtype": "f16 (single fp16 product, fp32 accumulate)",
                                  "rel_err_vs_f16x3": float(f"{err:.3e}"),
                                  "note": "opt-in reduced precision (TSCNet body only): 6e-4..9e-4 vs the reference on synthetic "
                                          "and real clips (tests/test_gpu_parity.py::test_f16x1_mode_error_bands), "
                                          "200x the default mode's error"}
            e1._graphs.clear()
            del m1, e1, out1, ref_out
        # ---- the reduced-precision mode WITH margin: every conformer kernel family on one fp16 product, the dense convs
        # (where all of F16X1's error comes from: profiles/r06_mix_ablation.json) on three; secondary leg, never the headline ----
        if world == 1 and x3 and not args.no_f16x1:
            ref_out = run(wav).clone()
            mm = TSCNet(64, sh.F, n_fft=sh.n_fft, hop=sh.hop, device=dev, mfma_mode="f16mix").load_state_dict(sd).eval()
            em_ = mm.engine
            runm = em_.enhance if args.no_graph else em_.enhance_graphed
            outm = runm(wav)
            torch.cuda.synchronize()
            km = max(2, min(args.steps, 10))
            t0 = time.perf_counter()
            for _ in range(km):
                runm(wav)
            torch.cuda.synchronize()
            dtm = (time.perf_counter() - t0) / km
            errm = float((outm - ref_out).abs().max() / ref_out.abs().max())
            line["f16mix_mode"] = {"ms_per_step": round(1e3 * dtm, 3), "value": round(sh.B * sh.T / dtm, 1), "unit": "frames/s",
                                   "steps": km, "single_product_families": list(em_.mix_single),
                                   "dtype": "f16: one fp16 product in the conformer kernels, three split products in the dense / sub-pixel convs; fp32 accumulate",
                                   "rel_err_vs_f16x3": float(f"{errm:.3e}"),
                                   "note": "opt-in reduced precision with margin: <= 2e-4 vs the reference on synthetic and real clips, "
                                           "asserted two-sidedly by tests/test_gpu_parity.py::test_f16mix_mode_error_band; "
                                           "per-family ablation in profiles/r06_mix_ablation.json"}
            em_._graphs.clear()
            del mm, em_, outm, ref_out
        # ---- BASELINE configs[4] and [3] in the same line (N = 1 only; each takes about a second) ----------------
        if world == 1 and x3 and args.workload == "16k" and not args.no_extra:
            for key, fn in (("stream_config5", lambda: stream_leg(model, dev)),
                            ("workload_48k", lambda: workload_48k_leg(dev, args.mfma_mode))):
                try:
                    line[key] = fn()
                except Exception as e:                          # noqa: BLE001 - the headline line must survive
                    line[key] = {"error": f"{type(e).__name__}: {e}"[:300]}
            torch.cuda.empty_cache()
        # ---- BASELINE configs[2]: the reference's training step (train.py:173-205), 32 clips per GPU, data parallel ----
        if not args.no_train and args.workload == "16k" and x3:
            eng.release_workspaces()                            # the inference graphs and workspaces are not needed any more
            torch.cuda.empty_cache()
            tr = train_leg(eng, sd, dev, rank, world, barrier, batch=args.train_batch)
            if rank == 0:
                line["train_step"] = tr
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(sd, sh)
    elif not args.no_train:
        # launcher self-test of the same collective shape: a flat gradient bucket averaged over the gloo ranks
        b = cdist.FlatBucket({"w": (1000,), "b": (7,)})
        b.flat.fill_(float(rank + 1))
        cdist.allreduce_mean(b.flat)
        want = sum(range(1, world + 1)) / world
        if rank == 0:
            line["train_step"] = {"stub": True, "grad_mean_ok": bool(abs(float(b["w"][0]) - want) < 1e-6),
                                  "bucket_floats": int(b.numel)}

    # The JSON line is the LAST thing on stdout: RCCL prints a version banner through C stdio, which sits in libc's
    # buffer until exit and would otherwise land after the line - so the process group is shut down and libc's
    # buffers are flushed first.
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:                                           # noqa: BLE001 - cosmetic
        pass
    if rank == 0:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
