"""GPU: the native data-parallel step (flat gradient bucket all-reduce before the clip, per-rank BatchNorm statistics,
global-sample-index SpecAugment/dropout streams) with 2 ranks sharing the one GPU of the test box over gloo.
(The driver's 2/4/8-GPU runs use the same code with backend nccl = RCCL.)"""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _cfg():
    from wakeword_trainer_home_amd.config import get_preset
    cfg = get_preset("cnn_small_logmel40")
    cfg.training.epochs, cfg.optimizer.warmup_epochs, cfg.training.batch_size = 1, 0, 8
    cfg.model.dropout = 0.3
    return cfg


def _data():
    from wakeword_trainer_home_amd.data import make_synthetic_batch
    wave, y = make_synthetic_batch(32, 24000, seed=9)
    y[::3] = 1
    return wave, y


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, str(REPO))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from wakeword_trainer_home_amd.models import create_model
    from wakeword_trainer_home_amd.training import Trainer
    cfg = _cfg()
    torch.manual_seed(50 + rank)
    model = create_model("cnn_small", dropout=cfg.model.dropout, dropout_seed=5)
    wave, y = _data()
    # global batch of 16 per step = rank 0's 8 clips followed by rank 1's 8 clips
    batches = [(wave[16 * s + 8 * rank:16 * s + 8 * rank + 8], y[16 * s + 8 * rank:16 * s + 8 * rank + 8]) for s in range(2)]
    t = Trainer(model, batches, batches[:1], cfg, checkpoint_dir=Path(out_dir) / f"ck{rank}", device="cuda:0")
    assert t.world_size == world and t.native
    losses = []
    t.add_callback(type("R", (), {"on_batch_end": lambda self, i, l, a: losses.append(l)})())
    t.train_epoch(0)
    torch.save({"sd": {k: v.cpu() for k, v in model.state_dict().items()}, "loss": losses}, Path(out_dir) / f"r{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_native_two_rank_step_equals_sharded_emulation(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    port = _free_port()
    mp.start_processes(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r0 = torch.load(tmp_path / "r0.pt", weights_only=False)
    r1 = torch.load(tmp_path / "r1.pt", weights_only=False)
    for k in r0["sd"]:
        if "running" in k or "num_batches" in k:
            continue
        assert torch.equal(r0["sd"][k], r1["sd"][k]), k          # replicas stay in lock-step

    # single-process emulation on the CPU oracle: per-shard forward/backward (own BatchNorm statistics, dropout and
    # SpecAugment drawn by GLOBAL sample index), averaged gradients, clip, AdamW
    from oracle.cnn_small import CNNSmallOracle, dropout_keep_mask
    from oracle.train_step import TorchLoss, frontend
    cfg = _cfg()
    a = cfg.augmentation
    spec = dict(freq_mask_param=a.freq_mask_param, time_mask_param=a.time_mask_param, n_freq_masks=a.n_freq_masks,
                n_time_masks=a.n_time_masks, freq_mask_prob=a.freq_mask_prob, time_mask_prob=a.time_mask_prob)
    torch.manual_seed(50)
    from wakeword_trainer_home_amd.models import create_model
    init = create_model("cnn_small", dropout=0.3, dropout_seed=5).state_dict()     # rank 0's init is broadcast
    reps = [CNNSmallOracle(dropout=0.3, dropout_seed=5) for _ in range(2)]
    for m in reps:
        m.load_state_dict(init)
        m.train()
    opts = [torch.optim.AdamW(m.parameters(), lr=cfg.training.learning_rate, weight_decay=cfg.optimizer.weight_decay)
            for m in reps]
    crit = TorchLoss("cross_entropy", eps=cfg.loss.label_smoothing)
    wave, y = _data()
    from oracle.specaugment import specaug_indices, specaug_apply
    from oracle import features as OF
    for s in range(2):
        for r, m in enumerate(reps):
            sl = slice(16 * s + 8 * r, 16 * s + 8 * r + 8)
            lm = OF.logmel_torch(wave[sl].numpy()).numpy()
            idx = specaug_indices(8, 40, 151, seed=a.seed, step=s, sample_offset=8 * r, **spec)
            x = torch.from_numpy(specaug_apply(lm, idx, a.n_freq_masks))
            opts[r].zero_grad(set_to_none=True)
            feats = m.features(x)
            keep = dropout_keep_mask(8, 64, 0.3, 5, s, sample_offset=8 * r)
            pooled = feats * torch.from_numpy(keep.astype(np.float32) / np.float32(1.0 - float(np.float32(0.3))))
            loss = crit(m.classifier(pooled), y[sl])
            loss.backward()
            assert abs(loss.item() - (r0 if r == 0 else r1)["loss"][s]) < 1e-3, (s, r)
        for p0, p1 in zip(reps[0].parameters(), reps[1].parameters()):
            avg = (p0.grad + p1.grad) / 2
            p0.grad, p1.grad = avg.clone(), avg.clone()
        for r, m in enumerate(reps):
            torch.nn.utils.clip_grad_norm_(m.parameters(), cfg.optimizer.gradient_clip)
            opts[r].step()
    for (k, v) in reps[0].state_dict().items():
        if "running" in k or "num_batches" in k:
            continue
        # Adam amplifies round-off-level gradient differences into lr-sized steps: 2 steps x lr 1e-3
        np.testing.assert_allclose(r0["sd"][k].numpy(), v.numpy(), atol=2.5e-3, err_msg=k)


def _worker_generic(rank, world, port, out_dir, arch, overlap="auto"):
    sys.path.insert(0, str(REPO))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from wakeword_trainer_home_amd.models import create_model
    from wakeword_trainer_home_amd.training import Trainer
    cfg = _cfg()
    cfg.training.dp_overlap = overlap
    torch.manual_seed(70 + rank)                      # different initial weights per rank: the Trainer broadcasts rank 0's
    model = create_model(arch, dropout=0.3, dropout_seed=5)
    wave, y = _data()
    batches = [(wave[16 * s + 8 * rank:16 * s + 8 * rank + 8], y[16 * s + 8 * rank:16 * s + 8 * rank + 8]) for s in range(2)]
    t = Trainer(model, batches, batches[:1], cfg, checkpoint_dir=Path(out_dir) / f"ck{rank}", device="cuda:0")
    assert t.world_size == world and not t.native
    losses = []
    t.add_callback(type("R", (), {"on_batch_end": lambda self, i, l, a: losses.append(l)})())
    t.train_epoch(0)
    torch.save({"sd": {k: v.cpu() for k, v in model.state_dict().items()}, "loss": losses, "collective": t.last_collective},
               Path(out_dir) / f"r{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_overlapped_bucket_allreduce_equals_the_in_stream_form(tmp_path):
    """MobileNetV3's 6 MB gradient bucket, 2 ranks: the tail of the bucket (late layers) all-reduced from a
    post-accumulate-grad hook while the early layers' backward is still running + the head in-stream afterwards, against ONE
    in-stream all-reduce after the backward.  Same sums in the same order per element: losses and parameters are bit-identical,
    on both ranks.  "auto" picks the overlapped form for this bucket (>= Trainer.DP_OVERLAP_MIN_BYTES) on eager steps."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    out = {}
    for name, ov in (("split", True), ("one", False), ("auto", "auto")):
        d = tmp_path / name
        d.mkdir()
        mp.start_processes(_worker_generic, args=(2, _free_port(), str(d), "mobilenetv3", ov), nprocs=2, join=True,
                           start_method="spawn")
        out[name] = [torch.load(d / f"r{r}.pt", weights_only=False) for r in range(2)]
    assert out["split"][0]["collective"] == "overlapped" and out["one"][0]["collective"] == "in-stream"
    assert out["auto"][0]["collective"] == "overlapped"
    for r in range(2):
        assert out["split"][r]["loss"] == out["one"][r]["loss"] == out["auto"][r]["loss"]
    for k, v in out["one"][0]["sd"].items():
        if "running" in k or "num_batches" in k:
            continue
        assert torch.equal(v, out["split"][0]["sd"][k]) and torch.equal(v, out["split"][1]["sd"][k]), k
        assert torch.equal(v, out["auto"][0]["sd"][k]), k


def _worker_custom_criterion(rank, world, port, out_dir):
    sys.path.insert(0, str(REPO))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.train_step import TorchLoss
    from wakeword_trainer_home_amd.models import create_model
    from wakeword_trainer_home_amd.training import Trainer
    cfg = _cfg()
    torch.manual_seed(70)
    model = create_model("crnn", dropout=0.0)
    wave, y = _data()
    batches = [(wave[16 * s + 8 * rank:16 * s + 8 * rank + 8], y[16 * s + 8 * rank:16 * s + 8 * rank + 8]) for s in range(2)]
    t = Trainer(model, batches, batches[:1], cfg, checkpoint_dir=Path(out_dir) / f"ck{rank}", device="cuda:0",
                criterion=TorchLoss("cross_entropy", eps=0.05))
    assert not t.native and not t._async_autograd                      # a foreign criterion: the reference-style step
    t.train_epoch(0)
    torch.save({"sd": {k: v.cpu() for k, v in model.state_dict().items()}}, Path(out_dir) / f"r{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_crnn_with_a_custom_criterion_stays_in_lock_step(tmp_path):
    """crnn is bucketed, but its recurrent layers' gradients are born outside the bucket.  On the reference-style step (user
    criterion -> clip_gradients + the optimizer read ``p.grad``) the averaged bucket values must be what they read: replicas
    fed DIFFERENT shards end with identical parameters (round 2 left the local gradients in place: silent divergence)."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    mp.start_processes(_worker_custom_criterion, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r0 = torch.load(tmp_path / "r0.pt", weights_only=False)
    r1 = torch.load(tmp_path / "r1.pt", weights_only=False)
    for k in r0["sd"]:
        if "running" in k or "num_batches" in k:
            continue
        assert torch.equal(r0["sd"][k], r1["sd"][k]), k


@pytest.mark.timeout(300)
@pytest.mark.parametrize("arch", ["crnn", "mobilenetv3"])
def test_generic_models_train_data_parallel(tmp_path, arch):
    """crnn / mobilenetv3 through the Trainer's generic step with 2 ranks: parameters broadcast from rank 0, gradients averaged
    every step -> the replicas' weights stay identical (BatchNorm running statistics are per rank, as under DDP)."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    port = _free_port()
    mp.start_processes(_worker_generic, args=(2, port, str(tmp_path), arch), nprocs=2, join=True, start_method="spawn")
    r0 = torch.load(tmp_path / "r0.pt", weights_only=False)
    r1 = torch.load(tmp_path / "r1.pt", weights_only=False)
    assert len(r0["loss"]) == 2 and all(np.isfinite(r0["loss"] + r1["loss"]))
    for k in r0["sd"]:
        if "running" in k or "num_batches" in k:
            continue
        assert torch.equal(r0["sd"][k], r1["sd"][k]), k


def _worker_bad_target(rank, world, port, out_dir, arch):
    sys.path.insert(0, str(REPO))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from wakeword_trainer_home_amd.models import create_model
    from wakeword_trainer_home_amd.training import Trainer
    cfg = _cfg()
    torch.manual_seed(50)
    model = create_model(arch, dropout=cfg.model.dropout, dropout_seed=5)
    wave, y = _data()
    y = y.clone()
    batches = [[wave[16 * s + 8 * rank:16 * s + 8 * rank + 8], y[16 * s + 8 * rank:16 * s + 8 * rank + 8].clone()] for s in range(2)]
    if rank == 1:
        batches[0][1][2] = 7                           # an invalid target in rank 1's FIRST batch only
    t = Trainer(model, [tuple(b) for b in batches], [tuple(batches[1])], cfg, checkpoint_dir=Path(out_dir) / f"ck{rank}",
                device="cuda:0")
    done = []
    t.add_callback(type("R", (), {"on_batch_end": lambda self, i, l, a: done.append(i)})())
    t.train_epoch(0)
    torch.save({"sd": {k: v.cpu() for k, v in model.state_dict().items()}, "done": done, "launched": t.launched_steps,
                "applied": t.optimizer.step_count() if hasattr(t.optimizer, "step_count") else None},
               Path(out_dir) / f"r{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("arch", ["cnn_small", "mobilenetv3", "crnn"])
def test_bad_batch_on_one_rank_is_skipped_by_all(tmp_path, arch):
    """The skip decision of the sync-free steps is per-step device state (found_inf).  It rides in the spare last element of
    the gradient bucket through the same all-reduce, so a bad target on ONE rank (finite gradients there, because bad
    targets are clamped) makes EVERY rank's fused optimizer skip: replicas stay identical and nothing hangs."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    port = _free_port()
    mp.start_processes(_worker_bad_target, args=(2, port, str(tmp_path), arch), nprocs=2, join=True, start_method="spawn")
    r0 = torch.load(tmp_path / "r0.pt", weights_only=False)
    r1 = torch.load(tmp_path / "r1.pt", weights_only=False)
    assert r0["done"] == [1] and r1["done"] == [1], (r0["done"], r1["done"])
    assert r0["launched"] == r1["launched"] == 2
    if r0["applied"] is not None:
        assert r0["applied"] == r1["applied"] == 1     # one update applied, one skipped -- on both ranks
    for k in r0["sd"]:
        if "running" in k or "num_batches" in k:
            continue
        assert torch.equal(r0["sd"][k], r1["sd"][k]), k
        assert torch.isfinite(r0["sd"][k]).all(), k


@pytest.mark.timeout(600)
def test_bench_one_rank_rccl_plumbing():
    """bench.py --force-dist: the step with the RCCL process group initialised (backend nccl, ReduceOp.AVG on the two
    gradient buckets, the found_inf slot) on the one GPU of this box -- the code path the driver's 2/4/8-GPU runs take."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import json
    import subprocess
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--force-dist", "--steps", "4", "--warmup", "2", "--batch", "64",
                        "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=540)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                                   # stdout carries exactly ONE JSON line
    out = json.loads(lines[0])
    assert out["config"]["ranks_seen"] == 1 and out["config"]["collective"].startswith("nccl")
    assert out["value"] > 0 and np.isfinite(out["config"]["last_loss"])


@pytest.mark.timeout(900)
def test_bench_two_ranks_through_its_own_spawner():
    """``python bench.py --gpus 2`` with no launcher around it: bench.py starts its own ranks (torch.distributed.run, before
    anything in the parent touches HIP) and relays rank 0's line -- the command path of the SCALE record.  On this one-GPU
    box both ranks use cuda:0 and the process group is gloo (test-only switches; the driver's runs are nccl, one GPU each)."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--batch", "64",
                        "--passes", "2", "--backend", "gloo", "--share-device0"], capture_output=True, text=True, env=env,
                       timeout=840)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                                   # exactly ONE stdout line: rank 0's record
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["ranks_seen"] == 2 and len(out["config"]["devices"]) == 2
    assert out["config"]["global_batch"] == 128 and out["config"]["parallelism"] == "dp2"
    assert out["steps"] == 4 and out["warmup"] == 2 and len(out["passes_ms_per_step"]) == 2
    assert out["config"]["collective"].startswith("gloo")
    assert out["value"] > 0 and np.isfinite(out["config"]["last_loss"])
    assert "cpu_baseline" not in out                                   # reported at N=1 only
