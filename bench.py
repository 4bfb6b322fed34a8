#!/usr/bin/env python3
"""Benchmark of the wakeword training inner loop on MI355X.

    python bench.py --gpus N --steps K --warmup W

N>1 without a launcher: this process starts the N ranks itself (``python -m torch.distributed.run``) BEFORE anything
touches the GPU and relays rank 0's JSON line; under ``torch.distributed.run`` (the driver's command) it is a rank.

One "step" = the whole hot path on one batch of synthetic 16 kHz x 1.5 s clips already resident in HBM:
fused log-mel(40)+SpecAugment -> cnn_small forward -> 2-class loss -> backward -> [RCCL all-reduce of the flat
gradient bucket] -> grad-norm clip -> AdamW step, i.e. wakeword_trainer_home_amd.training.Trainer._step_native
(BASELINE.json config 2: cnn_small + log-mel 40, batch 512 per GPU).  Rank 0 prints ONE JSON line with the
metric, the roofline of the dominant kernel (HIP-event timed inside the timed region) and the CPU baseline
(the oracle's PyTorch CPU step on the host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")     # before HIP initialises; see wakeword_trainer_home_amd/__init__.py

N_SAMPLES = 24000            # 16 kHz x 1.5 s
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is the achievable copy rate
A_ELEMS = 64 * 20 * 76       # one conv activation tensor per sample
# ALGORITHMIC fp32 bytes per sample and launch, by kernel class (DESIGN.md "Measurement"): the per-layer
# accounting of SURVEY.md §8d (write y, read y fwd, read y bwd, write dy, read dy) split over the kernels.
ALGO_BYTES = {
    "logmel_specaug": 96000 + 24160,
    "conv_stem_fwd": 24160 + A_ELEMS * 4,
    "dwconv3x3_fwd": 2 * A_ELEMS * 4,
    "pwconv1x1_fwd": 2 * A_ELEMS * 4,
    "gap_fwd": A_ELEMS * 4,
    "pwconv1x1_bwd": 3 * A_ELEMS * 4,
    "dwconv3x3_bwd": 3 * A_ELEMS * 4,
    "conv_stem_bwd": 2 * A_ELEMS * 4,
    "audio_augment": int(4 * N_SAMPLES * 2.5),      # read x, write out, read a noise segment for half of the clips
}


# HBM bytes per launch come from the PMC passes of THIS command (separate rocprofv3 --pmc runs: FETCH_SIZE x2 on gfx950 +
# WRITE_SIZE, aggregated by tools/pmc_traffic.py into profiles/pmc_traffic.json together with the batch / dtype / source
# digest they were measured on).  bench.py cannot collect counters itself; it reports the file's figure only when the file
# describes the running configuration AND the kernel sources it was measured on, and null otherwise.
TRAFFIC_FILE = ROOT / "profiles" / "pmc_traffic.json"
LOGMEL_FLOPS = 4.41e6        # per clip: 151 frames x (rFFT-1024 + power + sparse mel), SURVEY.md §8d
VALU_PEAK_TFLOPS = 157.3     # fp32 vector peak, MI355X_MICROARCH.md


BENCH_KERNEL_SOURCES = ("ww_frontend.hip", "ww_conv_fwd.hip", "ww_conv_bwd.hip", "ww_model.hip", "ww_act.h", "ww_fft.h",
                        "ww_internal.h")


def kernel_source_digest():
    """sha1 over the sources of the kernels this benchmark runs (front end + conv stack): ties a PMC traffic figure to the
    kernels it was measured on."""
    import hashlib
    h = hashlib.sha1()
    for name in BENCH_KERNEL_SOURCES:
        h.update(name.encode())
        h.update((ROOT / "wakeword_trainer_home_amd" / "csrc" / name).read_bytes())
    return h.hexdigest()[:12]


def measured_traffic(kernel, batch, dtype):
    """-> (bytes per launch | None, provenance string)."""
    try:
        t = json.loads(TRAFFIC_FILE.read_text())
    except Exception:
        return None, "no profiles/pmc_traffic.json"
    if t.get("batch") != batch or t.get("dtype") != dtype:
        return None, f"{TRAFFIC_FILE.name} was measured at batch {t.get('batch')} / {t.get('dtype')}"
    src = f"{TRAFFIC_FILE.name} ({t.get('label')}, csrc digest {t.get('csrc_digest')})"
    if t.get("csrc_digest") != kernel_source_digest():
        return None, src + " is stale: the kernel sources changed since that PMC pass"
    v = t.get("bytes_per_launch", {}).get(kernel)
    return (None if v is None else float(v)), src


def algo_bytes(kernel, esz):
    """fp32 figures above scale with the activation element size (features stay fp32)."""
    b = ALGO_BYTES.get(kernel, 0)
    if kernel in ("logmel_specaug", "audio_augment"):
        return b
    if kernel == "conv_stem_fwd":
        return 24160 + A_ELEMS * esz
    return b * esz // 4


def step_algo_bytes(esz):
    return 120160 + 45 * A_ELEMS * esz               # 17.63 MB (fp32) / 8.88 MB (bf16) per sample


def cpu_baseline(batch, seconds_budget=20.0):
    """Reference-style PyTorch CPU path (oracle/train_step.py: torch.stft log-mel -> SpecAugment -> cnn_small in
    torch.nn -> label-smoothing CE -> backward -> clip -> AdamW), timed on this host's cores."""
    import torch
    from oracle.cnn_small import CNNSmallOracle
    from oracle.train_step import TorchLoss, frontend, train_step
    from wakeword_trainer_home_amd.data import make_synthetic_batch
    # the 1-GPU box's CPU share is 16 cores; asking torch for all 256 hardware threads oversubscribes them
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    torch.set_num_threads(max(1, min(avail, 16)))
    cores = torch.get_num_threads()
    torch.manual_seed(0)
    model = CNNSmallOracle(dropout=0.3).to(memory_format=torch.channels_last)
    model.train()
    crit = TorchLoss("cross_entropy", eps=0.05)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4)
    wave, y = make_synthetic_batch(batch, N_SAMPLES, seed=1234)
    spec = dict(freq_mask_param=15, time_mask_param=35, n_freq_masks=2, n_time_masks=2, freq_mask_prob=0.5,
                time_mask_prob=0.5)

    def one(i):
        x, _ = frontend(wave.numpy(), spec, seed=2024, step=i)
        train_step(model, crit, opt, x.to(memory_format=torch.channels_last), y, 1.0)
    one(0)
    t0 = time.perf_counter()
    one(1)
    per = time.perf_counter() - t0
    steps = max(2, min(20, int(seconds_budget / max(per, 1e-3))))
    t0 = time.perf_counter()
    for i in range(steps):
        one(2 + i)
    dt = time.perf_counter() - t0
    return {"value": round(batch * steps / dt, 1), "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"{steps} steps x batch {batch} of the same workload (oracle PyTorch CPU step, fp32), "
                      f"{dt:.1f} s wall"}


class _fd1_to_stderr:
    """RCCL prints its version banner on fd 1 when the communicator is created; stdout must carry ONE JSON line."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=512, help="clips per GPU per step")
    ap.add_argument("--dtype", choices=("bf16", "f32", "f16"), default="bf16",
                    help="storage of the conv-stack activations/gradients (arithmetic is fp32 either way); "
                         "BASELINE config 2 names bf16, f32 is the 1e-3 parity mode")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=128)
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the RCCL process group even for one rank (plumbing test on a 1-GPU box)")
    ap.add_argument("--augment", action="store_true",
                    help="BASELINE config 4's input stage: on-GPU RIR convolution + background mix ahead of the log-mel "
                         "(rir_prob 0.25, noise prob 0.5, SNR 5-20 dB: src/config/defaults.py:82-87); not the headline line")
    ap.add_argument("--rir-len", type=int, default=4000, help="RIR taps for --augment (0.25 s at 16 kHz)")
    ap.add_argument("--dp-overlap", action="store_true", help="two gradient buckets (Trainer.dp_overlap)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step as a captured HIP graph (Trainer.enable_hip_graph; not the headline line)")
    ap.add_argument("--passes", type=int, default=3,
                    help="the timed region (--steps steps) is run this many times; the MEDIAN is reported (SURVEY §8d)")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="process-group backend; gloo exists for tests/ only (the spawner path on a one-GPU box)")
    ap.add_argument("--share-device0", action="store_true",
                    help="tests/ only: every rank uses cuda:0 (a one-GPU box rehearsing the N-rank command path)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: start the N ranks from here.  Nothing in this process has touched HIP (torch is not even imported),
        # so the children are fresh processes; rank 0's JSON line is relayed as this process's only stdout line.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
        proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
        lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
        for ln in proc.stdout.splitlines():
            if ln not in lines[-1:]:
                print(ln, file=sys.stderr)
        if lines:
            print(lines[-1], flush=True)
        raise SystemExit(proc.returncode if proc.returncode else (0 if lines else 1))

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP hot path has no CPU fallback")
    if args.share_device0:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        with _fd1_to_stderr():
            if args.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device(dev))
            else:
                dist.init_process_group("gloo")
            warm = torch.zeros(1, device=dev)
            dist.all_reduce(warm)                      # communicator (and its banner) are created here
            torch.cuda.synchronize()

    from wakeword_trainer_home_amd import _native as nat
    from wakeword_trainer_home_amd.config import get_preset
    from wakeword_trainer_home_amd.data import make_synthetic_batch
    from wakeword_trainer_home_amd.models import create_model
    from wakeword_trainer_home_amd.training import Trainer

    cfg = get_preset("cnn_small_logmel40")
    cfg.training.batch_size = args.batch
    cfg.training.dp_overlap = args.dp_overlap
    torch.manual_seed(1234)                                   # same initial weights on every rank
    model = create_model("cnn_small", num_classes=2, pretrained=False, dropout=cfg.model.dropout,
                         act_dtype={"bf16": "bf16", "f16": "fp16"}.get(args.dtype, "fp32"))
    esz = 4 if args.dtype == "f32" else 2
    import contextlib
    import tempfile
    with contextlib.redirect_stdout(sys.stderr):          # stdout carries exactly ONE JSON line
        trainer = Trainer(model, [], [], cfg, checkpoint_dir=Path(tempfile.mkdtemp(prefix="wwbench_")), device=dev)
    # a few distinct synthetic batches, generated on the device, rank-specific seeds (weak scaling)
    pool = [make_synthetic_batch(args.batch, N_SAMPLES, seed=1234 + 97 * rank + i, device=dev) for i in range(4)]
    trainer.model.train()
    if args.augment:
        from wakeword_trainer_home_amd.data import AudioAugmentation
        g = torch.Generator(device=dev).manual_seed(77)
        decay = torch.exp(-torch.arange(args.rir_len, device=dev) / (args.rir_len / 6.0))
        rirs = torch.randn(64, args.rir_len, device=dev, generator=g) * decay
        noises = 0.1 * torch.randn(64, 10 * 16000, device=dev, generator=g)
        a = cfg.augmentation
        trainer.audio_augmentation = AudioAugmentation(
            sample_rate=16000, device=dev, background_noise_prob=a.background_noise_prob,
            noise_snr_range=(a.noise_snr_min, a.noise_snr_max), rir_prob=a.rir_prob, rirs=rirs, noises=noises, seed=a.seed)

    last_done = [None]

    staged = {}

    def prepare(i):                                       # input stage of step i on the side stream (Trainer's pipeline)
        wave, y = pool[i % len(pool)]
        staged[i] = trainer._prepare_native(wave, y, i)

    gate = os.environ.get("WW_INPUT_GATE", "")           # experiment: fwd | bwd | mid (profiles/EXPERIMENTS.md), default none
    if gate:
        model.input_gate, model.gate_event = gate, torch.cuda.Event()

    def step(i, lookahead=True):
        if i not in staged:
            prepare(i)
        prep = staged.pop(i)
        if lookahead and not gate:
            prepare(i + 1)                                # one batch of lookahead, as Trainer.train_epoch does
        for done in trainer._step_native(None, None, i, prepared=prep):   # results arrive one step late
            trainer.state.global_step += 1
            last_done[0] = done
        if lookahead and gate:                            # the look-ahead input stage waits for this step's gate event
            if trainer._in_stream is not None:
                trainer._in_stream.wait_event(model.gate_event)
            prepare(i + 1)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def run_steps(first, n):
        """n whole steps: n input stages (the first one cannot overlap anything, the rest run one batch ahead on the side
        stream) and n model steps; nothing is launched for a batch that is not consumed."""
        for j in range(n):
            step(first + j, lookahead=j < n - 1)
        for done in trainer._flush_pending():
            last_done[0] = done

    # the cyclic garbage collector stays out of the timed region, as in timeit: one generation-2 pass over this process's heap
    # (torch + its imports) is a 50-80 ms host pause -- 40-60 steps of this benchmark (seen in tools/bench_models.py: one 76 ms
    # step among thirty 2.9 ms ones).  Collected HERE, ahead of the warm-up: between the warm-up and the first timed pass it left
    # the GPU idle long enough to drop its clocks, and that pass read 6-8 % slower than the next two.
    import gc
    gc.collect()
    gc.disable()
    # warm-up with HIP events on every kernel class; the first steps also tell which class dominates
    nat.prof_enable(dev, None)
    run_steps(0, args.warmup)
    fence()
    warm = nat.prof_collect(dev)
    # the roofline kernel is the largest class on the step's critical path (the main stream).  The input stage (log-mel,
    # waveform augmentation) runs a batch ahead on the side stream and is reported separately as roofline.side_stream:
    # its event spans include the time it shares CUs with the conv kernels, and it is timed alone below as well.
    side = ("logmel_specaug", "audio_augment")
    main_classes = {k: v for k, v in warm.items() if k not in side}
    dominant = max(main_classes, key=lambda k: main_classes[k][0]) if main_classes else "dwconv3x3_bwd"

    # the input-stage kernel alone on an idle GPU (north_star: "achieved HBM GB/s for STFT/mel")
    nat.prof_enable(dev, ["logmel_specaug"])
    for i in range(5):
        trainer._features(pool[i % len(pool)][0], training=True, step=i)
    fence()
    alone = nat.prof_collect(dev).get("logmel_specaug", (0.0, 0))
    nat.prof_enable(dev, [])
    nxt = args.warmup
    if args.graph:
        # the step as ONE replayed HIP graph (Trainer._graph_capture): captured after the next eager step, then two replays
        # to settle.  Graph nodes cannot carry HIP events, so the per-kernel figures below come from the eager warm-up.
        trainer.use_hip_graph = True
        run_steps(nxt, 4)
        nxt += 4
        assert trainer._graph is not None, "the HIP graph was not captured"

    # SURVEY §8d protocol: the region of EXACTLY --steps steps is timed --passes times (barrier + synchronize on both
    # sides of each), no HIP events inside; the median pass is the reported one.
    pass_dt = []
    for _ in range(max(1, args.passes)):
        fence()
        t0 = time.perf_counter()
        run_steps(nxt, args.steps)
        fence()
        pass_dt.append(time.perf_counter() - t0)
        nxt += args.steps
    # per-kernel HIP events: one more pass of the same region with events around the dominant class and the input stage
    # (on the streams they are launched on), so the event records cost the timed passes nothing
    ev_dt = None
    if not args.graph:
        nat.prof_enable(dev, [dominant, "logmel_specaug"])
        fence()
        t0 = time.perf_counter()
        run_steps(nxt, args.steps)
        fence()
        ev_dt = time.perf_counter() - t0
        nxt += args.steps
    last = None if last_done[0] is None else (last_done[0][1], last_done[0][2])
    gc.enable()
    prof = nat.prof_collect(dev) if not args.graph else {k: v for k, v in warm.items()}
    nat.prof_enable(dev, [])
    devices = [torch.cuda.get_device_name(local_rank) + f" (cuda:{local_rank})"]
    if use_dist:
        t = torch.tensor(pass_dt, dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)             # per pass: the slowest rank's time
        pass_dt = [float(v) for v in t.tolist()]
        gathered = [None] * dist.get_world_size()
        dist.all_gather_object(gathered, devices[0])
        devices = gathered

    if rank == 0:
        dt = sorted(pass_dt)[len(pass_dt) // 2]               # median pass
        total = args.batch * world * args.steps
        value = total / dt
        ms, launches = prof.get(dominant, (0.0, 0))
        per_launch_s = (ms / launches) * 1e-3 if launches else float("nan")
        algo = algo_bytes(dominant, esz) * args.batch
        achieved = algo / per_launch_s / 1e9 if launches else float("nan")
        traffic, traffic_src = measured_traffic(dominant, args.batch, args.dtype)

        def logmel_leg(ms_n, where):
            if not ms_n[1]:
                return None
            sec = ms_n[0] / ms_n[1] * 1e-3
            gbs = ALGO_BYTES["logmel_specaug"] * args.batch / sec / 1e9
            tf = LOGMEL_FLOPS * args.batch / sec / 1e12
            return {"where": where, "launch_us": round(sec * 1e6, 2), "launches_timed": ms_n[1],
                    "achieved_GBps": round(gbs, 1), "frac_hbm": round(gbs / HBM_PEAK_GBS, 4),
                    "achieved_TFLOPs": round(tf, 2), "frac_valu_f32": round(tf / VALU_PEAK_TFLOPS, 4)}
        out = {
            "metric": "training samples/sec (16kHz x 1.5s clips)", "value": round(value, 1), "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "passes_ms_per_step": [round(v / args.steps * 1e3, 4) for v in pass_dt], "timing": "median of the passes; each "
            "pass = barrier + synchronize, exactly --steps steps (input stages included), barrier + synchronize",
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "BASELINE config 2: cnn_small + log-mel(40) + SpecAugment, fwd/bwd + clip + AdamW, "
                                   "16 kHz x 1.5 s clips resident in HBM" + (f" + on-GPU RIR({args.rir_len} taps)/noise-mix augmentation "
                                   "(config 4's input stage)" if args.augment else ""), "batch_per_gpu": args.batch,
                       "activation_storage": args.dtype, "arithmetic": "f32",
                       "global_batch": args.batch * world, "n_samples": N_SAMPLES,
                       "parallelism": f"dp{world}" if world > 1 else "single",
                       "ranks_seen": dist.get_world_size() if use_dist else 1, "devices": devices,
                       "collective": (f"{dist.get_backend()} all-reduce (AVG) of the flat gradient bucket, "
                                      + ("2 buckets, the first under the early layers' backward" if trainer.dp_overlap
                                         else "in-stream after the backward")) if use_dist else None,
                       "hip_graph": bool(args.graph),
                       "last_loss": None if last is None else round(last[0], 6)},
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "traffic_source": traffic_src,
                         "launch_us": round(per_launch_s * 1e6, 2), "launches_timed": launches,
                         "timed_in": "eager warm-up steps (nodes of the replayed graph cannot carry HIP events)" if args.graph
                         else f"one more pass of the same {args.steps}-step region right after the timed passes, events on "
                              f"this class + the input stage only ({ev_dt / args.steps * 1e3:.4f} ms/step with them)",
                         "algorithmic_bytes_per_launch": algo,
                         "step_frac_of_hbm_roofline": round(value / world * step_algo_bytes(esz) / (HBM_PEAK_GBS * 1e9), 4),
                         # the input stage: NOT on the critical path (side stream, one batch ahead), far from both of its
                         # rooflines -- latency/LDS-bound FFT; algorithmic 120 160 B and 4.41 MFLOP per clip
                         "side_stream": {"kernel": "logmel_specaug",
                                         "algorithmic_bytes_per_launch": ALGO_BYTES["logmel_specaug"] * args.batch,
                                         "alone": logmel_leg(alone, "idle GPU, before the timed region, full-device grid"),
                                         "in_step": logmel_leg(prof.get("logmel_specaug", (0.0, 0)),
                                                               "timed region, sharing the CUs with the conv kernels, "
                                                               f"{trainer.input_stage_workgroups} persistent workgroups "
                                                               "(Trainer.input_stage_workgroups: one per CU)")}},
            # HIP-event SPANS around each class's launches during the warm-up steps (all classes on): they include queue
            # gaps and the stretch of running beside the other stream, so they do not add up to ms_per_step
            "event_span_ms_per_step_warmup": {k: round(v[0] / max(args.warmup, 1), 4) for k, v in sorted(warm.items())},
        }
        if not args.no_cpu_baseline and world == 1:       # reported at N=1 only (bounded sample, rank 0)
            out["cpu_baseline"] = cpu_baseline(args.cpu_batch)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
