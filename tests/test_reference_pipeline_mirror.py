"""GPU: the reference's own end-to-end test script (``tests/test_training_pipeline.py``: ten numbered checks, each a
shape / range / finiteness assertion on the public API) run step for step against THIS build's drop-in modules -- same
calls, same arguments, same assertions; where the reference uses ``resnet18`` (outside the hot path) the native
``cnn_small`` / ``mobilenetv3`` / ``crnn`` / ``gru`` models stand in.  The numerical parity of each piece is pinned
elsewhere (golden fixtures, oracles); this file checks that a user of the reference's API finds it working."""
import pytest
import torch
import torch.nn as nn
from torch.utils.data import DataLoader, TensorDataset

pytestmark = pytest.mark.gpu
DEV = "cuda"          # the reference passes device="cuda" everywhere (test_training_pipeline.py:30)


def test_1_model_architectures():
    """test_training_pipeline.py:49-82: every architecture maps its test input to (4, 2) logits."""
    from wakeword_trainer_home_amd.models import create_model
    for arch, shape in (("cnn_small", (4, 1, 64, 50)), ("mobilenetv3", (4, 1, 64, 50)), ("crnn", (4, 1, 64, 50)),
                        ("gru", (4, 50, 40))):
        model = create_model(arch, num_classes=2, pretrained=False).to(DEV)
        model.eval()
        with torch.no_grad():
            out = model(torch.randn(*shape).to(DEV))
        assert out.shape == (4, 2), f"Expected output shape (4, 2), got {out.shape}"
        assert sum(p.numel() for p in model.parameters()) > 0
    for arch in ("resnet18", "lstm", "tcn"):          # named by the reference factory, outside this build's hot path
        with pytest.raises(ValueError, match="outside this build"):
            create_model(arch, num_classes=2, pretrained=False)
    with pytest.raises(ValueError, match="Unknown architecture"):
        create_model("nope")


def test_2_loss_functions():
    """:84-119: both losses give a finite scalar on (32, 2) logits."""
    from wakeword_trainer_home_amd.models import create_loss_function
    predictions = torch.randn(32, 2).to(DEV)
    targets = torch.randint(0, 2, (32,)).to(DEV)
    for loss_name in ("cross_entropy", "focal_loss"):
        criterion = create_loss_function(loss_name=loss_name, num_classes=2, label_smoothing=0.1, device=DEV)
        loss = criterion(predictions, targets)
        assert loss.dim() == 0, "Loss should be scalar"
        assert torch.isfinite(loss), "Loss should be finite"


def test_3_metrics_calculation():
    """:121-151."""
    from wakeword_trainer_home_amd.training.metrics import MetricsCalculator
    predictions = torch.randn(100, 2).to(DEV)
    targets = torch.randint(0, 2, (100,)).to(DEV)
    m = MetricsCalculator(device=DEV).calculate(predictions, targets)
    for v in (m.accuracy, m.precision, m.recall, m.f1_score, m.fpr, m.fnr):
        assert 0 <= v <= 1


def test_4_metrics_tracker():
    """:153-189: three epochs of five batches; history and best epoch."""
    from wakeword_trainer_home_amd.training.metrics import MetricsTracker
    tracker = MetricsTracker(device=DEV)
    for _ in range(3):
        tracker.reset()
        for _ in range(5):
            tracker.update(torch.randn(20, 2).to(DEV), torch.randint(0, 2, (20,)).to(DEV))
        tracker.save_epoch_metrics(tracker.compute())
    assert len(tracker.get_epoch_history()) == 3, "Should have 3 epochs"
    best_epoch, best_metrics = tracker.get_best_epoch("f1_score")
    assert 0 <= best_epoch < 3 and 0 <= best_metrics.f1_score <= 1


def test_5_class_weights():
    """:191-221."""
    from wakeword_trainer_home_amd.training.metrics import calculate_class_weights
    for method in ("balanced", "inverse", "sqrt_inverse"):
        weights = calculate_class_weights({"positive": 200, "negative": 1800}, method=method, device=DEV)
        assert weights.shape == (2,) and (weights > 0).all()


def test_6_augmentation():
    """:223-271: AudioAugmentation on a (1, 24000) clip keeps shape and finiteness; SpecAugment on (1, 64, 50) likewise."""
    from wakeword_trainer_home_amd.data import AudioAugmentation, SpecAugment
    augmentor = AudioAugmentation(sample_rate=16000, device=DEV, time_stretch_range=(0.9, 1.1), pitch_shift_range=(-1, 1),
                                  background_noise_prob=0.0)
    waveform = torch.randn(1, 16000 * 3 // 2).to(DEV)
    augmented = augmentor(waveform)
    assert augmented.shape == waveform.shape, "Augmentation changed shape"
    assert torch.isfinite(augmented).all(), "Augmentation produced non-finite values"
    spec_aug = SpecAugment(freq_mask_param=15, time_mask_param=35, n_freq_masks=2, n_time_masks=2)
    spectrogram = torch.randn(1, 64, 50).to(DEV)
    augmented_spec = spec_aug(spectrogram)
    assert augmented_spec.shape == spectrogram.shape, "SpecAugment changed shape"


def test_7_optimizer_and_scheduler():
    """:273-336: the three optimizers step a dummy model; the four schedulers construct and step."""
    from wakeword_trainer_home_amd.training.optimizer_factory import create_optimizer, create_scheduler
    model = nn.Sequential(nn.Linear(10, 50), nn.ReLU(), nn.Linear(50, 2)).to(DEV)
    for opt_name in ("adam", "adamw", "sgd"):
        optimizer = create_optimizer(model, optimizer_name=opt_name, learning_rate=0.001, weight_decay=1e-4)
        loss = model(torch.randn(4, 10, device=DEV)).sum()
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
    for sched_name in ("cosine", "step", "plateau", "none"):
        optimizer = create_optimizer(model, optimizer_name="adam", learning_rate=0.001)
        scheduler = create_scheduler(optimizer, scheduler_name=sched_name, epochs=50, warmup_epochs=0)
        if scheduler is not None:
            scheduler.step(0.5) if sched_name == "plateau" else scheduler.step()
        assert (scheduler is None) == (sched_name == "none")


@pytest.mark.parametrize("arch", ["cnn_small", "mobilenetv3"])
def test_8_to_10_training_loop_checkpointing_and_loading(tmp_path, arch):
    """:338-480: three epochs on 100 random (1, 64, 50) feature maps with the default config -> result dict, checkpoints on
    disk, and a fresh Trainer loads the best one and reports the saved epoch."""
    from wakeword_trainer_home_amd.config import WakewordConfig
    from wakeword_trainer_home_amd.models import create_model
    from wakeword_trainer_home_amd.training import Trainer
    config = WakewordConfig()
    config.training.epochs = 3
    config.training.batch_size = 8
    config.training.early_stopping_patience = 10
    # the defaults' warmup_epochs (3) is not < epochs (3): create_scheduler raises on that -- in the reference too
    # (optimizer_factory.py:253; its test script logs the step as failed).  Pin the behaviour, then train with a valid one.
    with pytest.raises(ValueError, match=r"Warmup epochs \(3\) must be less than total epochs \(3\)"):
        Trainer(model=create_model(arch, num_classes=2, pretrained=False), train_loader=[], val_loader=[], config=config,
                checkpoint_dir=tmp_path / "never", device=DEV)
    config.optimizer.warmup_epochs = 1
    model = create_model(arch, num_classes=2, pretrained=False)
    dummy_features = torch.randn(100, 1, 64, 50)
    dummy_labels = torch.randint(0, 2, (100,))
    dataset = TensorDataset(dummy_features, dummy_labels)
    train_loader = DataLoader(dataset, batch_size=8, shuffle=True)
    val_loader = DataLoader(dataset, batch_size=8, shuffle=False)
    checkpoint_dir = tmp_path / "checkpoints"
    trainer = Trainer(model=model, train_loader=train_loader, val_loader=val_loader, config=config,
                      checkpoint_dir=checkpoint_dir, device=DEV)
    results = trainer.train()
    assert "history" in results, "Missing history in results"
    assert len(results["history"]["train_loss"]) == 3, "Should have 3 epochs"
    assert results["final_epoch"] == 2, "Final epoch should be 2"
    assert all(torch.isfinite(torch.tensor(results["history"][k])).all() for k in ("train_loss", "val_loss"))
    ckpts = sorted(p.name for p in checkpoint_dir.iterdir())
    assert "best_model.pt" in ckpts, "No checkpoints found"
    # :434-480 -- a new model + Trainer, load_checkpoint, the state carries the saved epoch
    model2 = create_model(arch, num_classes=2, pretrained=False)
    trainer2 = Trainer(model=model2, train_loader=train_loader, val_loader=val_loader, config=config,
                       checkpoint_dir=checkpoint_dir, device=DEV)
    trainer2.load_checkpoint(checkpoint_dir / "best_model.pt")
    assert 0 <= trainer2.state.epoch <= 2
    ck = torch.load(checkpoint_dir / "best_model.pt", map_location="cpu", weights_only=False)
    loaded = model2.state_dict()
    for k, v in ck["model_state_dict"].items():
        assert torch.equal(loaded[k].cpu(), v.cpu()), k
