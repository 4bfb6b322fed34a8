"""CPU, world_size 2 over gloo: the Trainer's data-parallel wiring (parameter broadcast, gradient averaging
before the clip, per-rank BatchNorm statistics, epoch-level metric all-reduce, rank-0 checkpoints) equals a
single-process emulation of PyTorch-DDP semantics without SyncBN (SURVEY.md §7 hard parts, §8e)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _data():
    g = torch.Generator().manual_seed(77)
    x = torch.randn(32, 1, 16, 24, generator=g) * 2 - 4
    y = (torch.rand(32, generator=g) < 0.4).long()
    return x, y


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, str(REPO))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    import wakeword_trainer_home_amd.training.trainer as T
    from wakeword_trainer_home_amd.config import WakewordConfig
    from oracle.cnn_small import CNNSmallOracle
    from oracle.train_step import TorchLoss
    T.enforce_cuda = lambda: None
    cfg = WakewordConfig()
    cfg.training.epochs, cfg.optimizer.warmup_epochs, cfg.training.batch_size = 1, 0, 8
    cfg.training.checkpoint_frequency = "every_epoch"
    torch.manual_seed(100 + rank)                      # different init per rank: the broadcast must fix it
    model = CNNSmallOracle(dropout=0.0)
    x, y = _data()
    shard = slice(rank * 16, (rank + 1) * 16)          # rank r owns a contiguous half; 2 batches of 8 each
    xs, ys = x[shard], y[shard]
    batches = [(xs[i:i + 8], ys[i:i + 8]) for i in (0, 8)]
    t = T.Trainer(model, batches, batches, cfg, checkpoint_dir=Path(out_dir) / "ckpt", device="cpu",
                  criterion=TorchLoss("cross_entropy", eps=0.05))
    assert t.world_size == world and t.rank == rank
    res = t.train()
    torch.save({"sd": model.state_dict(), "hist": res["history"]}, Path(out_dir) / f"rank{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_matches_ddp_emulation(tmp_path):
    port = _free_port()
    mp.start_processes(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=False)
    r1 = torch.load(tmp_path / "rank1.pt", weights_only=False)
    for k in r0["sd"]:
        if "running" in k or "num_batches" in k:
            continue                                    # BatchNorm statistics stay per rank (DDP semantics)
        assert torch.equal(r0["sd"][k], r1["sd"][k]), k
    assert r0["hist"]["val_loss"] == r1["hist"]["val_loss"]          # epoch metrics are all-reduced
    assert sorted(p.name for p in (tmp_path / "ckpt").iterdir()) == ["best_model.pt", "checkpoint_epoch_001.pt"]

    # single-process emulation: two replicas from rank 0's init, per-shard BN, averaged gradients, clip, step
    from oracle.cnn_small import CNNSmallOracle
    from oracle.train_step import TorchLoss
    torch.manual_seed(100)
    reps = [CNNSmallOracle(dropout=0.0) for _ in range(2)]
    reps[1].load_state_dict(reps[0].state_dict())
    opts = [torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=1e-4) for m in reps]
    crit = TorchLoss("cross_entropy", eps=0.05)
    x, y = _data()
    for step in range(2):
        for r, m in enumerate(reps):
            m.train()
            opts[r].zero_grad(set_to_none=True)
            sl = slice(r * 16 + step * 8, r * 16 + step * 8 + 8)
            crit(m(x[sl].to(memory_format=torch.channels_last)), y[sl]).backward()
        for p0, p1 in zip(reps[0].parameters(), reps[1].parameters()):
            avg = (p0.grad + p1.grad) / 2
            p0.grad, p1.grad = avg.clone(), avg.clone()
        for r, m in enumerate(reps):
            torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
            opts[r].step()
    for (k, v), w in zip(reps[0].state_dict().items(), r0["sd"].values()):
        if "num_batches" in k:
            continue
        # Adam turns round-off-level gradient differences (2-thread workers vs this process's conv reduction
        # order) into lr-sized (1e-3) parameter differences on near-zero-gradient weights; 2 steps => 3e-4
        np.testing.assert_allclose(w.numpy(), v.numpy(), atol=3e-4, err_msg=k)


def _worker_bad_batch(rank, world, port, out_dir):
    sys.path.insert(0, str(REPO))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    import wakeword_trainer_home_amd.training.trainer as T
    from wakeword_trainer_home_amd.config import WakewordConfig
    from oracle.cnn_small import CNNSmallOracle
    from oracle.train_step import TorchLoss
    T.enforce_cuda = lambda: None
    cfg = WakewordConfig()
    cfg.training.epochs, cfg.optimizer.warmup_epochs, cfg.training.batch_size = 1, 0, 8
    torch.manual_seed(100)
    model = CNNSmallOracle(dropout=0.0)
    x, y = _data()
    xs, ys = x[rank * 16:(rank + 1) * 16].clone(), y[rank * 16:(rank + 1) * 16]
    if rank == 1:
        xs[3, 0, 2, 5] = float("nan")                   # rank 1's FIRST batch is poisoned; rank 0's is fine
    batches = [(xs[i:i + 8], ys[i:i + 8]) for i in (0, 8)]
    t = T.Trainer(model, batches, batches[1:], cfg, checkpoint_dir=Path(out_dir) / f"ckpt{rank}", device="cpu",
                  criterion=TorchLoss("cross_entropy", eps=0.05))
    done = []
    t.add_callback(type("R", (), {"on_batch_end": lambda self, i, l, a: done.append(i)})())
    t.train_epoch(0)
    torch.save({"sd": model.state_dict(), "done": done, "launched": t.launched_steps, "global": t.state.global_step},
               Path(out_dir) / f"rank{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_skip_is_a_global_decision(tmp_path):
    """One rank's non-finite loss: the reference skips that batch (trainer.py:177-179).  Data parallel, EVERY rank must skip
    it -- before the gradient all-reduce, or the healthy rank would wait in it alone (hang) or apply an update its peer did
    not (silent divergence).  Both ranks end with identical parameters, one applied step, two launched."""
    port = _free_port()
    mp.start_processes(_worker_bad_batch, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=False)
    r1 = torch.load(tmp_path / "rank1.pt", weights_only=False)
    assert r0["done"] == [1] and r1["done"] == [1]                  # batch 0 skipped on BOTH ranks, batch 1 trained
    assert r0["launched"] == r1["launched"] == 2 and r0["global"] == r1["global"] == 1
    for k in r0["sd"]:
        if "running" in k or "num_batches" in k:
            continue
        assert torch.equal(r0["sd"][k], r1["sd"][k]), k
        assert torch.isfinite(r0["sd"][k]).all(), k


def _bucketed_model():
    from wakeword_trainer_home_amd.models.flat_buckets import FlatBuckets

    class Tiny(FlatBuckets, torch.nn.Module):
        """A bucketed model whose layers do NOT write their gradients into the bucket (like the recurrent layers of crnn / gru):
        autograd allocates every .grad outside ``flat_grad``."""

        def __init__(self):
            super().__init__()
            self.a, self.b = torch.nn.Linear(16 * 24, 8), torch.nn.Linear(8, 2)

        def forward(self, x):
            return self.b(torch.tanh(self.a(x.flatten(1))))
    return Tiny()


def _worker_bucketed(rank, world, port, out_dir):
    sys.path.insert(0, str(REPO))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    import wakeword_trainer_home_amd.training.trainer as T
    from wakeword_trainer_home_amd.config import WakewordConfig
    from oracle.train_step import TorchLoss
    T.enforce_cuda = lambda: None
    cfg = WakewordConfig()
    cfg.training.epochs, cfg.optimizer.warmup_epochs, cfg.training.batch_size = 1, 0, 8
    cfg.optimizer.optimizer = "sgd"                      # lr-proportional updates: un-averaged gradients show at once
    torch.manual_seed(100)
    model = _bucketed_model()
    x, y = _data()
    xs, ys = x[rank * 16:(rank + 1) * 16], y[rank * 16:(rank + 1) * 16]
    batches = [(xs[i:i + 8], ys[i:i + 8]) for i in (0, 8)]
    t = T.Trainer(model, batches, batches[:1], cfg, checkpoint_dir=Path(out_dir) / f"ckpt{rank}", device="cpu",
                  criterion=TorchLoss("cross_entropy", eps=0.05))
    assert hasattr(model, "flat_grad_ext") and not t.native and not t._fused_optimizer
    t.train_epoch(0)
    torch.save({"sd": model.state_dict()}, Path(out_dir) / f"rank{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_bucketed_model_with_custom_criterion_averages_gradients(tmp_path):
    """A model with flat buckets whose gradients are born OUTSIDE the bucket, on the reference-style step (custom criterion, torch
    optimizer): the all-reduce runs on the bucket, and the averaged values must be what clip_gradients and torch.optim read
    through ``p.grad`` -- otherwise every rank applies its local gradient and the replicas drift apart without an error."""
    port = _free_port()
    mp.start_processes(_worker_bucketed, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=False)
    r1 = torch.load(tmp_path / "rank1.pt", weights_only=False)
    for k in r0["sd"]:
        assert torch.equal(r0["sd"][k], r1["sd"][k]), k              # lock-step replicas
    from oracle.train_step import TorchLoss
    torch.manual_seed(100)
    reps = [_bucketed_model(), _bucketed_model()]
    reps[1].load_state_dict(reps[0].state_dict())
    opts = [torch.optim.SGD(m.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4, nesterov=True) for m in reps]
    crit = TorchLoss("cross_entropy", eps=0.05)
    x, y = _data()
    for step in range(2):
        for r, m in enumerate(reps):
            opts[r].zero_grad(set_to_none=True)
            sl = slice(r * 16 + step * 8, r * 16 + step * 8 + 8)
            crit(m(x[sl]), y[sl]).backward()
        for p0, p1 in zip(reps[0].parameters(), reps[1].parameters()):
            avg = (p0.grad + p1.grad) / 2
            p0.grad, p1.grad = avg.clone(), avg.clone()
        for r, m in enumerate(reps):
            torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
            opts[r].step()
    for k, v in reps[0].state_dict().items():
        np.testing.assert_allclose(r0["sd"][k].numpy(), v.numpy(), atol=1e-6, err_msg=k)
