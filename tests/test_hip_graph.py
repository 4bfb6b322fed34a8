"""GPU: the HIP-graph-captured training step (BASELINE config 5 "hipGraph-captured step"; Trainer._graph_capture).

What a replay must reproduce is everything an eager step does, with the per-step quantities (Philox step of SpecAugment
and dropout, learning rate, optimizer step_state slot, pinned stats buffer) read from the device-resident control block
instead of baked launch arguments.  The kernels are deterministic (fixed-order reductions), so the bar is BIT equality:
loss trace, every parameter, every BatchNorm buffer and the optimizer's moments after N replayed steps equal N eager steps."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cfg(graph, epochs=2, sched="cosine"):
    from wakeword_trainer_home_amd.config import get_preset
    cfg = get_preset("cnn_small_logmel40")
    cfg.training.epochs, cfg.training.batch_size = epochs, 16
    cfg.optimizer.scheduler, cfg.optimizer.warmup_epochs = sched, 0
    cfg.optimizer.mixed_precision = True              # bf16 storage: the benched mode
    cfg.model.dropout = 0.3
    cfg.training.hip_graph, cfg.training.hip_graph_auto = graph, False      # the eager leg must stay eager for every model
    cfg.training.checkpoint_frequency = "best_only"
    return cfg


def _run(tmp_path, graph, batches, val, epochs=2):
    from wakeword_trainer_home_amd.models import create_model
    from wakeword_trainer_home_amd.training import Trainer
    cfg = _cfg(graph, epochs)
    torch.manual_seed(7)
    model = create_model("cnn_small", dropout=cfg.model.dropout, dropout_seed=3)
    t = Trainer(model, batches, val, cfg, checkpoint_dir=tmp_path / ("g" if graph else "e"), device=DEV)
    rec = []
    t.add_callback(type("R", (), {"on_batch_end": lambda self, i, l, a: rec.append((i, l, a))})())
    res = t.train()
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    opt = copy.deepcopy(t.optimizer.state_dict())
    return t, rec, sd, opt, res


def _assert_same(a, b):
    ta, ra, sa, oa, resa = a
    tb, rb, sb, ob, resb = b
    assert [r[0] for r in ra] == [r[0] for r in rb]
    assert [r[1] for r in ra] == [r[1] for r in rb], "loss traces differ"          # float equality: bit-identical steps
    assert [r[2] for r in ra] == [r[2] for r in rb]
    bits = lambda t: t.contiguous().view(torch.int32) if t.dtype == torch.float32 else t     # NaN == NaN bitwise
    for k in sa:
        assert torch.equal(bits(sa[k]), bits(sb[k])), k
    for (ka, va), (kb, vb) in zip(sorted(oa["state"].items()), sorted(ob["state"].items())):
        for f in va:
            assert torch.equal(torch.as_tensor(va[f]).cpu(), torch.as_tensor(vb[f]).cpu()), (ka, f)
    assert resa["history"]["train_loss"] == resb["history"]["train_loss"]
    assert resa["history"]["learning_rates"] == resb["history"]["learning_rates"]
    assert ta.launched_steps == tb.launched_steps and ta.state.global_step == tb.state.global_step


def test_replayed_steps_equal_eager_steps_bit_for_bit(tmp_path):
    """Waveform batches (the input stage stays on the side stream, one batch ahead; the model step is the replayed graph),
    2 epochs x 6 steps with a cosine schedule (the learning rate reaches the replay through the control block)."""
    from wakeword_trainer_home_amd.data import make_synthetic_batch
    wave, y = make_synthetic_batch(16 * 6, 24000, seed=5)
    y[::4] = 1
    batches = [(wave[16 * i:16 * i + 16], y[16 * i:16 * i + 16]) for i in range(6)]
    eager = _run(tmp_path, False, batches, batches[:1])
    graph = _run(tmp_path, True, batches, batches[:1])
    assert graph[0]._graph is not None and eager[0]._graph is None
    assert len(graph[1]) == 12
    _assert_same(eager, graph)


def test_ragged_batch_and_feature_inputs_fall_back_and_resume(tmp_path):
    """Feature-map batches (B,1,40,151), with a smaller batch in the middle: that batch takes the eager
    step, the replays before and after it stay in step with the Philox / optimizer-slot bookkeeping."""
    from tests.golden_util import make_inputs
    x, y = make_inputs(9, 16 * 5 + 5)
    cuts = [0, 16, 32, 37, 53, 69, 85]                       # batch 2 has 5 samples
    batches = [(x[a:b], y[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
    eager = _run(tmp_path, False, batches, batches[:1], epochs=1)
    graph = _run(tmp_path, True, batches, batches[:1], epochs=1)
    assert graph[0]._graph is not None and graph[0]._graph["in_shape"] == (16, 1, 40, 151)
    _assert_same(eager, graph)


def test_nonfinite_batch_is_skipped_inside_a_replay(tmp_path):
    """found_inf is device state: a poisoned batch replayed through the graph leaves parameters and the optimizer's step
    count untouched, exactly as the eager step does."""
    from tests.golden_util import make_inputs
    x, y = make_inputs(3, 16 * 5)
    x = x.clone()
    x[16 * 3 + 2, 0, 5, 7] = float("nan")                    # batch 3 is poisoned
    batches = [(x[16 * i:16 * i + 16], y[16 * i:16 * i + 16]) for i in range(5)]
    eager = _run(tmp_path, False, batches, batches[:1], epochs=1)
    graph = _run(tmp_path, True, batches, batches[:1], epochs=1)
    assert [r[0] for r in graph[1]] == [0, 1, 2, 4]
    assert graph[0].optimizer.step_count() == 4 and graph[0].launched_steps == 5
    _assert_same(eager, graph)


@pytest.mark.parametrize("arch", ["crnn", "mobilenetv3", "gru"])
def test_autograd_models_replay_bit_for_bit(tmp_path, arch):
    """crnn (BASELINE config 5's model), gru and mobilenetv3 run forward / loss / backward through autograd nodes whose
    bodies are C-ABI launches; with their parameters in one flat bucket the whole step (incl. the gather of the autograd
    gradients and the fused clip + optimizer) is captured and replayed.  Same bar: bit equality with the eager steps."""
    from wakeword_trainer_home_amd.data import make_synthetic_batch
    from wakeword_trainer_home_amd.models import create_model
    from wakeword_trainer_home_amd.training import Trainer
    from wakeword_trainer_home_amd.training.optimizer_factory import FlatFusedOptimizer
    wave, y = make_synthetic_batch(8 * 5, 24000, seed=6)
    y[::3] = 1
    batches = [(wave[8 * i:8 * i + 8], y[8 * i:8 * i + 8]) for i in range(5)]
    runs = []
    for graph in (False, True):
        cfg = _cfg(graph, epochs=2)
        cfg.optimizer.mixed_precision = False
        cfg.training.batch_size = 8
        torch.manual_seed(11)
        model = create_model(arch, dropout=0.3, dropout_seed=2)
        t = Trainer(model, batches, batches[:1], cfg, checkpoint_dir=tmp_path / f"{arch}{int(graph)}", device=DEV)
        assert isinstance(t.optimizer, FlatFusedOptimizer) and t._async_autograd
        rec = []
        t.add_callback(type("R", (), {"on_batch_end": lambda self, i, l, a: rec.append((i, l, a))})())
        res = t.train()
        runs.append((t, rec, {k: v.detach().cpu().clone() for k, v in model.state_dict().items()},
                     copy.deepcopy(t.optimizer.state_dict()), res))
    assert runs[1][0]._graph is not None and runs[0][0]._graph is None
    assert len(runs[1][1]) == 10
    _assert_same(runs[0], runs[1])


def test_graph_replay_is_the_default_for_models_that_prefer_it(tmp_path):
    """``training.hip_graph_auto`` (default on): a model that declares ``prefers_hip_graph`` (mobilenetv3: ~340 short launches per
    step, host-bound when issued eagerly) trains through the replayed graph with a default config -- and bit-identically to the
    eager steps it replaces; cnn_small (no preference) stays eager; the switch turns it off."""
    from tests.golden_util import make_inputs
    from wakeword_trainer_home_amd.config import get_preset
    from wakeword_trainer_home_amd.models import create_model
    from wakeword_trainer_home_amd.training import Trainer
    x, y = make_inputs(4, 8 * 5)
    batches = [(x[8 * i:8 * i + 8], y[8 * i:8 * i + 8]) for i in range(5)]
    out = {}
    for name, auto in (("auto", True), ("off", False)):
        cfg = get_preset("cnn_small_logmel40")
        cfg.training.epochs, cfg.training.batch_size, cfg.optimizer.warmup_epochs = 1, 8, 0
        cfg.training.checkpoint_frequency = "best_only"
        cfg.training.hip_graph_auto = auto
        assert cfg.training.hip_graph is False
        torch.manual_seed(5)
        model = create_model("mobilenetv3", dropout=0.2, dropout_seed=2)
        t = Trainer(model, batches, batches[:1], cfg, checkpoint_dir=tmp_path / name, device=DEV)
        assert t.use_hip_graph is auto
        rec = []
        t.add_callback(type("R", (), {"on_batch_end": lambda self, i, l, a: rec.append(l)})())
        t.train()
        assert (t._graph is not None) is auto
        out[name] = (rec, {k: v.detach().cpu().clone() for k, v in model.state_dict().items()})
    assert out["auto"][0] == out["off"][0]                       # loss trace, float equality
    for k, v in out["auto"][1].items():
        assert torch.equal(v, out["off"][1][k]), k
    cfg = get_preset("cnn_small_logmel40")
    t = Trainer(create_model("cnn_small"), batches, batches[:1], cfg, checkpoint_dir=tmp_path / "c", device=DEV)
    assert t.use_hip_graph is False


def test_unasked_capture_that_fails_leaves_an_eager_run(tmp_path, caplog):
    """``hip_graph_auto`` captures a graph nobody asked for; whatever goes wrong in that capture must not end a run that the
    eager step completes (the reference re-raises only errors of the step itself, trainer.py:214-226).  A model whose forward
    raises only while the stream is capturing trains to the SAME losses and parameters as with the switch off, the failure is
    logged once and graph mode is off afterwards.  The same failure under an explicit ``hip_graph=True`` is the caller's to see."""
    import logging
    from tests.golden_util import make_inputs
    from wakeword_trainer_home_amd.config import get_preset
    from wakeword_trainer_home_amd.models import create_model
    from wakeword_trainer_home_amd.training import Trainer
    x, y = make_inputs(4, 8 * 5)
    batches = [(x[8 * i:8 * i + 8], y[8 * i:8 * i + 8]) for i in range(5)]

    def build(name, auto, asked=False, poisoned=True):
        cfg = get_preset("cnn_small_logmel40")
        cfg.training.epochs, cfg.training.batch_size, cfg.optimizer.warmup_epochs = 1, 8, 0
        cfg.training.checkpoint_frequency = "best_only"
        cfg.training.hip_graph_auto, cfg.training.hip_graph = auto, asked
        torch.manual_seed(5)
        model = create_model("mobilenetv3", dropout=0.2, dropout_seed=2)
        if poisoned:
            inner = model.forward

            def forward(inp):
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("this model cannot be captured")
                return inner(inp)
            model.forward = forward
        t = Trainer(model, batches, batches[:1], cfg, checkpoint_dir=tmp_path / name, device=DEV)
        rec = []
        t.add_callback(type("R", (), {"on_batch_end": lambda self, i, l, a: rec.append(l)})())
        return t, model, rec

    t, model, rec = build("auto", True)
    assert t.use_hip_graph
    with caplog.at_level(logging.WARNING):
        t.train()
    assert len(rec) == 5 and t.use_hip_graph is False and t._graph is None and not t._graphs
    assert sum("HIP graph capture failed" in r.message for r in caplog.records) == 1
    t0, model0, rec0 = build("off", False, poisoned=False)
    t0.train()
    assert rec == rec0                                            # float equality: the failed capture changed nothing
    for (k, v), w in zip(model.state_dict().items(), model0.state_dict().values()):
        assert torch.equal(v, w), k
    t1, _, _ = build("asked", False, asked=True)
    with pytest.raises(RuntimeError, match="cannot be captured"):
        t1.train_epoch(0)


def test_full_and_ragged_batch_keep_one_graph_each(tmp_path):
    """An epoch of full batches plus a ragged last one, several epochs: each shape is captured ONCE (the second time it is seen)
    and both graphs stay -- the ragged batch does not evict the full batch's graph every epoch.  Still bit-identical to eager."""
    from tests.golden_util import make_inputs
    from wakeword_trainer_home_amd.training import Trainer
    x, y = make_inputs(9, 16 * 3 + 5)
    cuts = [0, 16, 32, 48, 53]                               # the last batch has 5 samples
    batches = [(x[a:b], y[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
    captures = []
    orig = Trainer._graph_capture

    def counting(self, feats):
        captures.append(tuple(feats.shape))
        return orig(self, feats)
    eager = _run(tmp_path, False, batches, batches[:1], epochs=4)
    Trainer._graph_capture = counting
    try:
        graph = _run(tmp_path, True, batches, batches[:1], epochs=4)
    finally:
        Trainer._graph_capture = orig
    assert captures == [(16, 1, 40, 151), (5, 1, 40, 151)], captures
    assert len(graph[0]._graphs) == 2
    _assert_same(eager, graph)


def test_changed_clip_norm_and_weight_decay_reach_the_replays(tmp_path):
    """``max_norm`` and the optimizer's hyper-parameters are baked into a captured launch by value.  Changing
    ``trainer.gradient_clip`` / ``param_groups[0]['weight_decay']`` between epochs must act on the following steps as it does on
    eager ones: the stale graph is dropped and the step re-captured (bit equality with the eager run that honours the change)."""
    from tests.golden_util import make_inputs
    from wakeword_trainer_home_amd.models import create_model
    from wakeword_trainer_home_amd.training import Trainer
    x, y = make_inputs(2, 16 * 4)
    batches = [(x[16 * i:16 * i + 16], y[16 * i:16 * i + 16]) for i in range(4)]
    runs = []
    for graph in (False, True):
        cfg = _cfg(graph, epochs=3)
        torch.manual_seed(7)
        model = create_model("cnn_small", dropout=cfg.model.dropout, dropout_seed=3)
        t = Trainer(model, batches, batches[:1], cfg, checkpoint_dir=tmp_path / ("g" if graph else "e"), device=DEV)
        rec = []

        class Cb:
            def on_batch_end(self, i, l, a):
                rec.append((i, l, a))

            def on_epoch_start(self, epoch):
                if epoch == 1:
                    t.gradient_clip = 0.05
                    t.optimizer.param_groups[0]["weight_decay"] = 0.1
        t.add_callback(Cb())
        res = t.train()
        runs.append((t, rec, {k: v.detach().cpu().clone() for k, v in model.state_dict().items()},
                     copy.deepcopy(t.optimizer.state_dict()), res))
    assert runs[1][0]._graph is not None and runs[1][0]._graph["key"][2] == 0.05
    _assert_same(runs[0], runs[1])
