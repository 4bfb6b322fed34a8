"""ctypes binding of libwwhip.so (include/wwhip.h) -- the only door to the HIP hot path.

There is deliberately NO fallback: if the shared library is missing, or a tensor is not
on an MI355X, these wrappers raise.  PyTorch is used for device memory and streams only.
"""
import contextlib
import ctypes as C
import math
import os
from pathlib import Path

import torch

_LIB_PATH = Path(__file__).resolve().parent / "csrc" / "libwwhip.so"
_lib = None
_ctx = {}

ABI_VERSION = 14
BWD_ALL, BWD_LATE, BWD_EARLY = 0, 1, 2
ACT_F32, ACT_BF16, ACT_F16 = 0, 1, 2
LOSS_CE, LOSS_FOCAL = 0, 1
WAVE_F32, WAVE_I16 = 0, 1
CNN_SMALL_NPTR = 47


class NativeError(RuntimeError):
    pass


class FeatCfg(C.Structure):
    _fields_ = [("sample_rate", C.c_int32), ("n_fft", C.c_int32), ("hop", C.c_int32), ("n_mels", C.c_int32),
                ("n_mfcc", C.c_int32), ("f_min", C.c_float), ("f_max", C.c_float), ("log_eps", C.c_float)]


class SpecAugCfg(C.Structure):
    _fields_ = [("freq_mask_param", C.c_int32), ("time_mask_param", C.c_int32), ("n_freq_masks", C.c_int32),
                ("n_time_masks", C.c_int32), ("freq_mask_prob", C.c_float), ("time_mask_prob", C.c_float)]


class BN(C.Structure):
    _fields_ = [("gamma", C.c_void_p), ("beta", C.c_void_p), ("running_mean", C.c_void_p),
                ("running_var", C.c_void_p), ("momentum", C.c_float), ("eps", C.c_float), ("training", C.c_int32)]


class AudioAugCfg(C.Structure):
    _fields_ = [("rir_prob", C.c_float), ("noise_prob", C.c_float), ("snr_min_db", C.c_float), ("snr_max_db", C.c_float)]


class LinearEpi(C.Structure):
    _fields_ = [("act", C.c_int32), ("dropout_p", C.c_float), ("seed", C.c_uint64), ("step", C.c_uint64),
                ("sample_offset", C.c_uint64)]


LIN_NONE, LIN_HARDSWISH, LIN_RELU, LIN_HARDSIGMOID = 0, 1, 2, 3


class OptimCfg(C.Structure):
    _fields_ = [("kind", C.c_int32), ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("weight_decay", C.c_float), ("momentum", C.c_float), ("max_norm", C.c_float)]


OPT_ADAM, OPT_ADAMW, OPT_SGD = 0, 1, 2


class GruDir(C.Structure):
    """ww_gru_dir: one direction of a bidirectional GRU layer (include/wwhip.h)."""
    _fields_ = [(n, C.c_void_p) for n in ("w_ih", "w_hh", "b_ih", "b_hh", "h0", "h_n", "ws", "dh_n", "dw_ih", "dw_hh", "db_ih",
                                          "db_hh", "dh0")]


class StepCtl(C.Structure):
    _fields_ = [("step", C.c_uint64), ("lr", C.c_float), ("parity", C.c_int32), ("loss_scale", C.c_float),
                ("growth_tracker", C.c_int32), ("reserved", C.c_int32 * 2)]


STEP_CTL_BYTES = C.sizeof(StepCtl)


class LossScale(C.Structure):
    _fields_ = [("scale", C.c_float * 2), ("growth_tracker", C.c_int32 * 2), ("growth_factor", C.c_float),
                ("backoff_factor", C.c_float), ("growth_interval", C.c_int32), ("reserved", C.c_int32)]


LOSS_SCALE_BYTES = C.sizeof(LossScale)


def loss_scale_new(dev, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000, growth_tracker=0):
    """Device ww_loss_scale (uint8[32] tensor) with torch.amp.GradScaler's defaults; both slots start equal."""
    host = LossScale((C.c_float * 2)(init_scale, init_scale), (C.c_int32 * 2)(growth_tracker, growth_tracker),
                     growth_factor, backoff_factor, growth_interval, 0)
    return torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).to(dev)


def loss_scale_read(t, slot=0) -> dict:
    s = LossScale.from_buffer_copy(bytes(t.cpu().numpy().tobytes()))
    return {"scale": s.scale[slot], "growth_tracker": s.growth_tracker[slot], "growth_factor": s.growth_factor,
            "backoff_factor": s.backoff_factor, "growth_interval": s.growth_interval}


class StepStats(C.Structure):
    _fields_ = [("loss", C.c_float), ("grad_norm", C.c_float), ("correct", C.c_int32), ("tp", C.c_int32),
                ("tn", C.c_int32), ("fp", C.c_int32), ("fn", C.c_int32), ("nonfinite", C.c_int32),
                ("bad_target", C.c_int32), ("count", C.c_int32), ("found_inf", C.c_float), ("reserved", C.c_int32)]


STEP_STATS_BYTES = C.sizeof(StepStats)

_vp, _i, _f, _u64, _sz = C.c_void_p, C.c_int, C.c_float, C.c_uint64, C.c_size_t
_SIGS = {
    "ww_abi_version": (C.c_int, []),
    "ww_last_error": (C.c_char_p, []),
    "ww_ctx_create": (C.c_int, [_i, C.POINTER(_vp)]),
    "ww_ctx_destroy": (C.c_int, [_vp]),
    "ww_ctx_bind_step_ctl": (C.c_int, [_vp, _vp]),
    "ww_ctx_set_logmel_workgroups": (C.c_int, [_vp, C.c_int]),
    "ww_step_ctl_advance": (C.c_int, [_vp, _vp]),
    "ww_feat_num_frames": (C.c_int, [_i, _i]),
    "ww_feat_mel_tables": (C.c_int, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp]),
    "ww_logmel_fwd": (C.c_int, [_vp, _vp, _i, _i, _i, C.POINTER(FeatCfg), _vp, C.POINTER(SpecAugCfg), _u64, _u64, _u64,
                                _vp, _vp]),
    "ww_specaug_apply": (C.c_int, [_vp, _vp, _i, _i, _i, C.POINTER(SpecAugCfg), _u64, _u64, _u64, _vp, _vp]),
    "ww_audio_augment_scratch_bytes": (_sz, [_i, _i]),
    "ww_audio_rir_spectra_bytes": (_sz, [_i]),
    "ww_audio_rir_spectra": (C.c_int, [_vp, _vp, _i, _i, _vp, _sz, _vp]),
    "ww_audio_augment": (C.c_int, [_vp, _vp, _vp, _i, _i, _vp, _i, _i, _vp, _vp, _i, _i, C.POINTER(AudioAugCfg), _u64, _u64,
                                   _u64, _vp, _vp, _sz, _vp]),
    "ww_gemm16_nt": (C.c_int, [_vp, _i, _vp, _vp, _vp, _i, C.c_long, C.c_long, C.c_long, _vp]),
    "ww_linear_mfma_fwd": (C.c_int, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, C.POINTER(LinearEpi), _vp, _vp, _vp]),
    "ww_linear_mfma_bwd_scratch_bytes": (_sz, [_i, _i, _i]),
    "ww_linear_mfma_bwd": (C.c_int, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, C.POINTER(LinearEpi), _vp, _vp, _vp, _vp, _sz,
                                     _vp]),
    "ww_dropout_bt": (C.c_int, [_vp, _vp, C.c_long, _i, _i, _i, _f, _u64, _u64, _u64, _i, _vp, C.c_long, _vp]),
    "ww_nhwc_scratch_bytes": (_sz, [_i]),
    "ww_bn_act_fwd": (C.c_int, [_vp, _vp, C.c_long, _i, C.POINTER(BN), _i, _vp, _vp, _vp, _vp, _vp]),
    "ww_bn_act_bwd": (C.c_int, [_vp, _vp, _vp, C.c_long, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "ww_conv1x1_bn_act_fwd": (C.c_int, [_vp, _i, _vp, _vp, _i, _i, _i, C.POINTER(BN), _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "ww_dwconv_bn_act_fwd": (C.c_int, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, C.POINTER(BN), _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ww_dwconv_nhwc_fwd": (C.c_int, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "ww_dwconv_nhwc_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "ww_pool_hw_fwd": (C.c_int, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "ww_scale_bc_fwd": (C.c_int, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "ww_scale_bc_bwd_gate": (C.c_int, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "ww_scale_pool_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "ww_se_bwd_scratch_bytes": (_sz, [_i, _i, _i]),
    "ww_se_fwd": (C.c_int, [_vp, _vp, _i, _i, _i, _i] + [_vp] * 9),
    "ww_se_bwd": (C.c_int, [_vp] * 8 + [_i, _i, _i, _i] + [_vp] * 6 + [_sz, _vp]),
    "ww_stem3x3s2_bn_act_fwd": (C.c_int, [_vp, _vp, _vp, _i, _i, _i, _i, C.POINTER(BN), _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ww_stem3x3s2_bwd_dw": (C.c_int, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "ww_im2col3x3s2": (C.c_int, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "ww_add_f32": (C.c_int, [_vp, _vp, _vp, _sz, _vp, _vp]),
    "ww_gru_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "ww_gru_fwd": (C.c_int, [_vp, _i, _vp, C.c_long, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, C.c_long, _vp, _vp, _sz, _vp]),
    "ww_gru_bidir_fwd": (C.c_int, [_vp, _i, _vp, C.c_long, _vp, _i, _i, _i, _i, _vp, C.c_long, _sz, _vp]),
    "ww_gru_bidir_bwd": (C.c_int, [_vp, _i, _vp, C.c_long, _vp, _vp, C.c_long, _i, _i, _i, _i, _sz, _vp, C.c_long, _vp]),
    "ww_gru_bwd": (C.c_int, [_vp, _i, _vp, C.c_long, _vp, _vp, _vp, C.c_long, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp, C.c_long, _i,
                             _vp, _vp, _vp, _vp, _vp, _vp]),
    "ww_clip_optim_step": (C.c_int, [_vp, C.POINTER(OptimCfg), _vp, _vp, _vp, _vp, _sz, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ww_layer_scratch_bytes": (_sz, []),
    "ww_conv_stem_fwd": (C.c_int, [_vp, _i, _vp, _vp, _i, _i, _i, _vp, C.POINTER(BN), _vp, _vp, _vp, _vp]),
    "ww_dwconv3x3_fwd": (C.c_int, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _vp, C.POINTER(BN), _vp, _vp, _vp, _vp]),
    "ww_pwconv1x1_fwd": (C.c_int, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _vp, C.POINTER(BN), _vp, _vp, _vp, _vp]),
    "ww_gap_fwd": (C.c_int, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "ww_head_fwd": (C.c_int, [_vp, _vp, _i, _i, _vp, _vp, _f, _i, _u64, _u64, _u64, _vp, _vp, _vp]),
    "ww_head_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _i, _i, _vp, _f, _i, _u64, _u64, _u64, _vp, _vp, _vp, _vp, _vp, _vp,
                              _vp, _vp, _vp]),
    "ww_pwconv1x1_bwd": (C.c_int, [_vp, _i] + [_vp] * 10 + [_i, _i, _i] + [_vp] * 7),
    "ww_dwconv3x3_bwd": (C.c_int, [_vp, _i] + [_vp] * 8 + [_i, _i, _i] + [_vp] * 7),
    "ww_conv_stem_bwd": (C.c_int, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "ww_cnn_small_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "ww_cnn_small_fwd": (C.c_int, [_vp, _i, C.POINTER(_vp), _vp, _i, _i, _i, _i, _f, _f, _f, _u64, _u64, _u64, _vp, _sz,
                                   _vp, _vp]),
    "ww_cnn_front_fwd": (C.c_int, [_vp, _i, C.POINTER(_vp), _vp, _i, _i, _i, _i, _f, _f, _vp, _sz, _vp, _vp]),
    "ww_cnn_front_bwd": (C.c_int, [_vp, _i, C.POINTER(_vp), C.POINTER(_vp), _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "ww_cnn_small_bwd": (C.c_int, [_vp, _i, C.POINTER(_vp), C.POINTER(_vp), _vp, _vp, _i, _i, _i, _f, _u64, _u64, _u64,
                                   _vp, _sz, _i, _vp]),
    "ww_ce2_loss_fwd_bwd": (C.c_int, [_vp, _vp, _vp, _i, _i, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "ww_grad_norm_clip": (C.c_int, [_vp, _vp, _sz, _f, _vp, _vp, _vp]),
    "ww_prof_num_classes": (C.c_int, []),
    "ww_prof_class_name": (C.c_char_p, [_i]),
    "ww_ctx_set_deferred_reduce": (C.c_int, [_vp, _i]),
    "ww_deferred_reduce_pending": (C.c_int, [_vp]),
    "ww_deferred_reduce_flush": (C.c_int, [_vp, _vp]),
    "ww_deferred_reduce_discard": (C.c_int, [_vp]),
    "ww_prof_enable": (C.c_int, [_vp, C.c_uint32]),
    "ww_prof_collect": (C.c_int, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_int32)]),
    "ww_prob_threshold": (_u64, [C.c_double]),
    "ww_philox4x32_10": (None, [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
}
EXPORTS = tuple(_SIGS)


def lib_path() -> Path:
    return _LIB_PATH


def load():
    """dlopen libwwhip.so (once).  Raises NativeError if it has not been built."""
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists():
            raise NativeError(
                f"{_LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"(or `make -C {_LIB_PATH.parent}`); there is no CPU fallback for the HIP hot path")
        lib = C.CDLL(os.fspath(_LIB_PATH))
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        if lib.ww_abi_version() != ABI_VERSION:
            raise NativeError(f"libwwhip ABI {lib.ww_abi_version()} != binding {ABI_VERSION}: rebuild")
        _lib = lib
    return _lib


def _check(rc, what):
    if rc != 0:
        msg = _lib.ww_last_error().decode(errors="replace")
        if rc == -1:
            raise ValueError(f"{what}: {msg}")
        raise NativeError(f"{what} failed ({rc}): {msg}")


def ctx(device) -> int:
    """Per-device ww_ctx handle (created on first use)."""
    idx = device.index if isinstance(device, torch.device) else torch.device(device).index
    if idx is None:
        idx = torch.cuda.current_device()
    h = _ctx.get(idx)
    if h is not None:
        return h
    lib = load()
    if idx not in _ctx:
        h = _vp()
        with torch.cuda.device(idx):
            _check(lib.ww_ctx_create(idx, C.byref(h)), "ww_ctx_create")
        _ctx[idx] = h
    return _ctx[idx]


def _dev(*tensors):
    d = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise NativeError("the HIP hot path needs tensors on an MI355X ('cuda') device; got a CPU tensor "
                              "(there is no CPU fallback)")
        if not t.is_contiguous():
            raise ValueError("non-contiguous tensor passed to the native path")
        d = t.device if d is None else d
        if t.device != d:
            raise ValueError("tensors on different devices")
    return d


def _dev_rows(*tensors):
    """Device check for (B,T,F) row-strided views (validated separately by _bt_rows); no contiguity requirement."""
    d = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise NativeError("the HIP hot path needs tensors on an MI355X ('cuda') device; got a CPU tensor "
                              "(there is no CPU fallback)")
        d = t.device if d is None else d
        if t.device != d:
            raise ValueError("tensors on different devices passed to the native path")
    return d


def _p(t):
    return None if t is None else t.data_ptr()


_ACT = {torch.float32: ACT_F32, torch.bfloat16: ACT_BF16, torch.float16: ACT_F16}


def act_code(dtype) -> int:
    """storage type of the conv-stack activation tensors: 'fp32'/torch.float32 or 'bf16'/torch.bfloat16."""
    if isinstance(dtype, str):
        dtype = {"fp32": torch.float32, "f32": torch.float32, "float32": torch.float32, "bf16": torch.bfloat16,
                 "bfloat16": torch.bfloat16, "fp16": torch.float16, "f16": torch.float16, "float16": torch.float16,
                 "half": torch.float16}.get(dtype.lower())
    if dtype not in _ACT:
        raise ValueError("activation storage must be float32, bfloat16 or float16")
    return _ACT[dtype]


def act_torch_dtype(code):
    return {ACT_BF16: torch.bfloat16, ACT_F16: torch.float16}.get(code, torch.float32)


def _stream(dev):
    """Raw handle of torch's current stream on `dev` (the C-ABI launches on it).  torch.cuda.current_stream(dev).cuda_stream
    builds a Stream object per call (4 us); the raw getter is the same value."""
    idx = dev.index if isinstance(dev, torch.device) else torch.device(dev).index
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice() if idx is None else idx)


class _NullGuard:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NULL_GUARD = _NullGuard()


def _guard(dev):
    """Device guard around a C-ABI call: a no-op when `dev` is already the current device (one process per GPU: always),
    torch.cuda.device(dev) otherwise."""
    idx = dev.index if isinstance(dev, torch.device) else torch.device(dev).index
    if idx is None or idx == torch._C._cuda_getDevice():
        return _NULL_GUARD
    return torch.cuda.device(dev)


def num_frames(n, hop):
    return load().ww_feat_num_frames(n, hop)


MELQ_TAB = 172


def mel_tables(cfg):
    """Host only: the mel filterbank of ``cfg`` as the kernels read it -> dict(start, len (n_mels,) int32; w compact band weights;
    melq_tab (MELQ_TAB,) int32 and melq_w: the matrix-pipe form of the n_fft-1024 kernel, see include/wwhip.h)."""
    import numpy as np
    M = int(cfg.n_mels)
    start, length = np.zeros(M, np.int32), np.zeros(M, np.int32)
    n_w, n_q = C.c_int32(0), C.c_int32(0)
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)
    _check(load().ww_feat_mel_tables(C.byref(cfg), ptr(start), ptr(length), None, 0, C.byref(n_w), None, None, 0, C.byref(n_q)),
           "ww_feat_mel_tables")
    w, qtab, qw = np.zeros(max(n_w.value, 1), np.float32), np.zeros(MELQ_TAB, np.int32), np.zeros(max(n_q.value, 1), np.float32)
    _check(load().ww_feat_mel_tables(C.byref(cfg), ptr(start), ptr(length), ptr(w), w.size, C.byref(n_w), ptr(qtab), ptr(qw),
                                     qw.size, C.byref(n_q)), "ww_feat_mel_tables")
    return dict(start=start, len=length, w=w[:n_w.value], melq_tab=qtab, melq_w=qw[:n_q.value])


def prob_threshold(p):
    return load().ww_prob_threshold(float(p))


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    load().ww_philox4x32_10(c, k, o)
    return list(o)


def make_feat_cfg(sample_rate=16000, n_fft=1024, hop=160, n_mels=40, n_mfcc=0, f_min=0.0, f_max=0.0, log_eps=1e-6):
    return FeatCfg(sample_rate, n_fft, hop, n_mels, n_mfcc, f_min, f_max, log_eps)


def make_specaug_cfg(freq_mask_param=15, time_mask_param=35, n_freq_masks=2, n_time_masks=2, freq_mask_prob=1.0,
                     time_mask_prob=1.0):
    return SpecAugCfg(freq_mask_param, time_mask_param, n_freq_masks, n_time_masks, freq_mask_prob, time_mask_prob)


def logmel_fwd(wave, cfg: FeatCfg, specaug: SpecAugCfg = None, seed=0, step=0, sample_offset=0, want_idx=False):
    """wave (B,N) f32|i16 cuda -> (B,1,F,T) f32 [, mask_idx (B,K,2) i32]."""
    dev = _dev(wave)
    if wave.dim() != 2:
        raise ValueError(f"waveform batch must be (B,N), got {tuple(wave.shape)}")
    if wave.dtype == torch.float32:
        dt = WAVE_F32
    elif wave.dtype == torch.int16:
        dt = WAVE_I16
    else:
        raise ValueError(f"waveform dtype must be float32 or int16, got {wave.dtype}")
    B, N = wave.shape
    if cfg.hop <= 0:
        raise ValueError("hop must be positive")
    T = 1 + N // cfg.hop
    F = cfg.n_mfcc if cfg.n_mfcc > 0 else cfg.n_mels
    out = torch.empty((B, 1, F, T), dtype=torch.float32, device=dev)
    idx = None
    if want_idx and specaug is not None:
        idx = torch.zeros((B, specaug.n_freq_masks + specaug.n_time_masks, 2), dtype=torch.int32, device=dev)
    with _guard(dev):
        _check(load().ww_logmel_fwd(ctx(dev), _p(wave), dt, B, N, C.byref(cfg), _p(out),
                                    C.byref(specaug) if specaug is not None else None, seed, step, sample_offset,
                                    _p(idx), _stream(dev)), "ww_logmel_fwd")
    return (out, idx) if want_idx else out


def specaug_apply_(x, specaug: SpecAugCfg, seed=0, step=0, sample_offset=0, want_idx=False):
    """in-place on x (B,1,F,T) or (B,F,T) f32 cuda."""
    dev = _dev(x)
    if x.dtype != torch.float32 or x.dim() not in (3, 4) or (x.dim() == 4 and x.shape[1] != 1):
        raise ValueError(f"features must be float32 (B,1,F,T) or (B,F,T), got {x.dtype} {tuple(x.shape)}")
    B, F, T = x.shape[0], x.shape[-2], x.shape[-1]
    idx = None
    if want_idx:
        idx = torch.zeros((B, specaug.n_freq_masks + specaug.n_time_masks, 2), dtype=torch.int32, device=dev)
    with _guard(dev):
        _check(load().ww_specaug_apply(ctx(dev), _p(x), B, F, T, C.byref(specaug), seed, step, sample_offset, _p(idx),
                                       _stream(dev)), "ww_specaug_apply")
    return idx


def audio_rir_spectra(rirs):
    """RIR bank (R,L) f32 cuda -> spectra for the FFT form of ww_audio_augment (compute once per bank)."""
    dev = _dev(rirs)
    if rirs.dim() != 2 or rirs.dtype != torch.float32:
        raise ValueError("rirs must be a float32 (count, length) tensor")
    R, L = rirs.shape
    nbytes = load().ww_audio_rir_spectra_bytes(R)
    spectra = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    with _guard(dev):
        _check(load().ww_audio_rir_spectra(ctx(dev), _p(rirs), R, L, _p(spectra), nbytes, _stream(dev)),
               "ww_audio_rir_spectra")
    return spectra


def audio_augment(wave, rirs, noises, rir_prob, noise_prob, snr_min_db, snr_max_db, seed=0, step=0, sample_offset=0,
                  want_choices=False, rir_spectra=None):
    """wave (B,N) f32 cuda -> augmented copy [, choices int32 (B,4)]; rirs (R,L) / noises (K,Nn) f32 cuda or None.
    ``rir_spectra`` (from ``audio_rir_spectra``) selects the FFT convolution; None the direct time-domain form."""
    dev = _dev(wave, rirs, noises)
    if wave.dim() != 2 or wave.dtype != torch.float32:
        raise ValueError(f"waveform batch must be float32 (B,N), got {wave.dtype} {tuple(wave.shape)}")
    for name, t in (("rirs", rirs), ("noises", noises)):
        if t is not None and (t.dim() != 2 or t.dtype != torch.float32):
            raise ValueError(f"{name} must be a float32 (count, length) tensor")
    B, N = wave.shape
    out = torch.empty_like(wave)
    R, L = (0, 0) if rirs is None else rirs.shape
    K, Nn = (0, 0) if noises is None else noises.shape
    nbytes = load().ww_audio_augment_scratch_bytes(B, N)
    scratch = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    choices = torch.empty((B, 4), dtype=torch.int32, device=dev) if want_choices else None
    cfg = AudioAugCfg(rir_prob, noise_prob, snr_min_db, snr_max_db)
    with _guard(dev):
        _check(load().ww_audio_augment(ctx(dev), _p(wave), _p(out), B, N, _p(rirs), R, L, _p(rir_spectra), _p(noises), K, Nn, C.byref(cfg),
                                       seed, step, sample_offset, _p(choices), _p(scratch), nbytes, _stream(dev)),
               "ww_audio_augment")
    return (out, choices) if want_choices else out


def _lin_check(x, w, bias):
    if x.dim() != 2 or w.dim() != 2 or x.shape[1] != w.shape[1] or x.dtype != torch.float32 or w.dtype != torch.float32:
        raise ValueError(f"linear: need float32 x (M,K) and weight (N,K), got {tuple(x.shape)} and {tuple(w.shape)}")
    if bias is not None and (bias.dim() != 1 or bias.shape[0] != w.shape[0] or bias.dtype != torch.float32):
        raise ValueError("linear: bias must be float32 (N,)")


def linear_mfma_fwd(x, w, bias=None, act=LIN_NONE, dropout_p=0.0, seed=0, step=0, sample_offset=0, mode=torch.float32,
                    want_pre=False):
    """y = dropout(act(x @ w.T + bias)) on the matrix cores -> y [, pre]."""
    dev = _dev(x, w, bias)
    _lin_check(x, w, bias)
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty((M, N), dtype=torch.float32, device=dev)
    pre = torch.empty((M, N), dtype=torch.float32, device=dev) if want_pre else None
    epi = LinearEpi(act, dropout_p, seed, step, sample_offset)
    with _guard(dev):
        _check(load().ww_linear_mfma_fwd(ctx(dev), act_code(mode), _p(x.contiguous()), _p(w.contiguous()), _p(bias), M, K, N,
                                         C.byref(epi), _p(pre), _p(y), _stream(dev)), "ww_linear_mfma_fwd")
    return (y, pre) if want_pre else y


def gemm16_nt(a, b, out_dtype=torch.float32):
    """a (M,K) @ b (N,K).T for bf16 / fp16 operands resident in HBM (fp32 accumulation) -> (M,N) float32 or the operand type."""
    dev = _dev(a, b)
    if a.dim() != 2 or b.dim() != 2 or a.shape[1] != b.shape[1] or a.dtype != b.dtype or a.dtype not in (torch.bfloat16, torch.float16):
        raise ValueError(f"gemm16_nt: need two bf16 or two fp16 matrices (M,K), (N,K), got {tuple(a.shape)} {a.dtype}, {tuple(b.shape)} {b.dtype}")
    if out_dtype not in (torch.float32, a.dtype):
        raise ValueError("gemm16_nt: out_dtype must be float32 or the operand type")
    a, b = a.contiguous(), b.contiguous()
    out = torch.empty((a.shape[0], b.shape[0]), dtype=out_dtype, device=dev)
    with _guard(dev):
        _check(load().ww_gemm16_nt(ctx(dev), act_code(a.dtype), _p(a), _p(b), _p(out), int(out_dtype == torch.float32),
                                   a.shape[0], b.shape[0], a.shape[1], _stream(dev)), "ww_gemm16_nt")
    return out


def _out(buf, shape, dev):
    """`buf` (a caller's contiguous fp32 tensor of `shape`'s size -- e.g. a gradient-bucket slot) or a fresh tensor."""
    if buf is None:
        return torch.empty(shape, dtype=torch.float32, device=dev)
    if buf.dtype != torch.float32 or not buf.is_contiguous() or buf.numel() != math.prod(shape) or buf.device != dev:
        raise ValueError("output buffer must be a contiguous float32 tensor of the result's size on the same device")
    return buf


# ---- deferred partial sums (include/wwhip.h: ww_ctx_set_deferred_reduce).  The backward nodes of the autograd models queue the
# "sum the partials" step of their weight-gradient kernels; ONE launch at the end of the backward pass (an autograd-engine
# callback) runs them all.  A queued gradient is not valid before the flush, so a node may only defer a gradient that autograd
# ADOPTS without reading it: a fresh bucket slot (models/flat_buckets.grad_slot) of a parameter that has no gradient yet.
_defer = {}          # device -> {"armed": the end-of-backward callback is queued, "keep": partial buffers alive until the flush}


def defer_begin(dev):
    """Called from a backward node: True when partial sums may be deferred in this backward pass (arms the flush callback)."""
    dev = torch.device(dev)
    st = _defer.setdefault(dev, {"armed": False, "keep": []})
    if not st["armed"]:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(lambda d=dev: _defer_end(d))
        except Exception:                          # not inside an autograd backward pass: nothing would flush
            return False
        st["armed"] = True
    return True


def defer_reset(dev):
    """Start of a forward pass: whatever a previous backward pass left queued or armed is stale (that pass raised before its
    end-of-backward callback ran) -- forget it, so the next backward arms a fresh flush."""
    st = _defer.get(torch.device(dev))
    if st is not None and (st["armed"] or st["keep"]):
        _check(load().ww_deferred_reduce_discard(ctx(torch.device(dev))), "ww_deferred_reduce_discard")
        st["armed"] = False
        st["keep"].clear()


def _defer_end(dev):
    deferred_flush(dev)
    _defer[dev]["armed"] = False


def deferred_flush(dev):
    """Run everything queued so far as one launch on the current stream (also used before a mid-backward all-reduce)."""
    dev = torch.device(dev)
    st = _defer.get(dev)
    if st is None or not st["keep"]:
        return
    with _guard(dev):
        _check(load().ww_deferred_reduce_flush(ctx(dev), _stream(dev)), "ww_deferred_reduce_flush")
    st["keep"].clear()


class _deferring:
    """with _deferring(dev, scratch): the C call inside queues its partial-sum step; `scratch` lives until the flush."""

    def __init__(self, dev, *keep):
        self.dev, self.keep = torch.device(dev), keep

    def __enter__(self):
        _check(load().ww_ctx_set_deferred_reduce(ctx(self.dev), 1), "ww_ctx_set_deferred_reduce")

    def __exit__(self, *exc):
        load().ww_ctx_set_deferred_reduce(ctx(self.dev), 0)
        _defer.setdefault(self.dev, {"armed": False, "keep": []})["keep"].extend(self.keep)


def linear_mfma_bwd(x, w, pre, dy, act=LIN_NONE, dropout_p=0.0, seed=0, step=0, sample_offset=0, mode=torch.float32,
                    need_dx=True, need_db=True, dw_out=None, db_out=None, defer=False):
    """-> (dx | None, dw, db | None).  dw_out / db_out: write the parameter gradients there (returned as given).
    defer: queue the split-K sum of dw (valid after deferred_flush)."""
    dev = _dev(x, w, pre, dy)
    _lin_check(x, w, None)
    M, K = x.shape
    N = w.shape[0]
    if tuple(dy.shape) != (M, N) or dy.dtype != torch.float32:
        raise ValueError(f"linear backward: dy must be float32 {(M, N)}, got {tuple(dy.shape)}")
    dx = torch.empty((M, K), dtype=torch.float32, device=dev) if need_dx else None
    dw = _out(dw_out, (N, K), dev)
    db = _out(db_out, (N,), dev) if need_db else None
    nbytes = load().ww_linear_mfma_bwd_scratch_bytes(M, K, N)
    scratch = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    epi = LinearEpi(act, dropout_p, seed, step, sample_offset)
    xc, dyc = x.contiguous(), dy.contiguous()
    with _guard(dev), (_deferring(dev, scratch) if defer else contextlib.nullcontext()):
        _check(load().ww_linear_mfma_bwd(ctx(dev), act_code(mode), _p(xc), _p(w.contiguous()), _p(pre),
                                         _p(dyc), M, K, N, C.byref(epi), _p(dx), _p(dw), _p(db), _p(scratch), nbytes,
                                         _stream(dev)), "ww_linear_mfma_bwd")
    return dx, dw, db


def _bt_rows(t, name):
    """(B,T,F) tensor (possibly a column slice of a wider buffer) -> row stride, checking the (b,t) rows are equidistant."""
    if t.dim() != 3 or t.dtype != torch.float32 or t.stride(2) != 1 or t.stride(0) != t.shape[1] * t.stride(1):
        raise ValueError(f"{name} must be float32 (B,T,F) with contiguous features and equidistant (b,t) rows")
    return t.stride(1)


def dropout_bt(x, p, seed=0, step=0, sample_offset=0, stream_id=0, out=None):
    """Philox dropout of a (B,T,C) tensor (rows may be strided); also its own backward when applied to the gradient."""
    dev = _dev_rows(x, out)
    ldx = _bt_rows(x, "x")
    B, T, Cc = x.shape
    if out is None:
        out = torch.empty((B, T, Cc), dtype=torch.float32, device=dev)
    with _guard(dev):
        _check(load().ww_dropout_bt(ctx(dev), _p(x), ldx, B, T, Cc, p, seed, step, sample_offset, stream_id, _p(out),
                                    _bt_rows(out, "out"), _stream(dev)), "ww_dropout_bt")
    return out


# ---- generic channels-last layers (MobileNetV3 body); activations are contiguous fp32 (M, C) / (B, H, W, C) tensors
_nhwc_scratch = {}


def nhwc_scratch(Cn, dev):
    """Scratch of one layer call, cached per (C, device): calls on one stream are ordered, so the buffer can be shared."""
    key = (Cn, dev)
    buf = _nhwc_scratch.get(key)
    if buf is None:
        buf = _nhwc_scratch[key] = torch.empty(load().ww_nhwc_scratch_bytes(Cn) // 4, dtype=torch.float32, device=dev)
    return buf


def bn_act_fwd(x, bn: BN, act, Cn):
    """x (..., C) -> (y, ss (2C), mr (2C))."""
    dev = _dev(x)
    M = x.numel() // Cn
    y = torch.empty_like(x)
    ss = torch.empty(2 * Cn, dtype=torch.float32, device=dev)
    mr = torch.empty(2 * Cn, dtype=torch.float32, device=dev)
    with _guard(dev):
        _check(load().ww_bn_act_fwd(ctx(dev), _p(x), M, Cn, C.byref(bn), act, _p(y), _p(ss), _p(mr), _p(nhwc_scratch(Cn, dev)),
                                    _stream(dev)), "ww_bn_act_fwd")
    return y, ss, mr


def bn_act_bwd(x, da, ss, mr, act, training, Cn, dgamma_out=None, dbeta_out=None):
    dev = _dev(x, da, ss, mr)
    M = x.numel() // Cn
    dx = torch.empty_like(x)
    dgamma = _out(dgamma_out, (Cn,), dev)
    dbeta = _out(dbeta_out, (Cn,), dev)
    with _guard(dev):
        _check(load().ww_bn_act_bwd(ctx(dev), _p(x), _p(da), M, Cn, _p(ss), _p(mr), act, int(training), _p(dx), _p(dgamma),
                                    _p(dbeta), _p(nhwc_scratch(Cn, dev)), _stream(dev)), "ww_bn_act_bwd")
    return dx, dgamma, dbeta


def conv1x1_bn_act_fwd(x, w, bn: BN, act, mode=torch.float32, residual=None):
    """Training-mode conv (x (M,K) @ w (N,K)^T) + BatchNorm + activation with the statistics taken in the GEMM's epilogue.
    residual (M,N), optional: added to the activated output in the same pass.  -> (y pre-BN (M,N), a (M,N), ss (2N), mr (2N))."""
    dev = _dev(x, w, residual)
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty((M, N), dtype=torch.float32, device=dev)
    a = torch.empty_like(y)
    ss = torch.empty(2 * N, dtype=torch.float32, device=dev)
    mr = torch.empty(2 * N, dtype=torch.float32, device=dev)
    scratch = nhwc_scratch(N, dev)
    if scratch.numel() < ((M + 63) // 64) * 2 * N:          # very tall and narrow: a scratch of its own size
        scratch = torch.empty(((M + 63) // 64) * 2 * N, dtype=torch.float32, device=dev)
    with _guard(dev):
        _check(load().ww_conv1x1_bn_act_fwd(ctx(dev), act_code(mode), _p(x), _p(w), M, K, N, C.byref(bn), act, _p(residual), _p(y), _p(a),
                                            _p(ss), _p(mr), _p(scratch), scratch.numel() * 4, _stream(dev)), "ww_conv1x1_bn_act_fwd")
    return y, a, ss, mr


def dwconv_bn_act_fwd(x, w, k, stride, bn: BN, act):
    """Depthwise conv + BatchNorm + activation -> (y pre-BN, a, ss, mr); the LDS kernel takes the statistics where it applies."""
    dev = _dev(x, w)
    B, H, W, Cn = x.shape
    Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    y = torch.empty((B, Ho, Wo, Cn), dtype=torch.float32, device=dev)
    a = torch.empty_like(y)
    ss = torch.empty(2 * Cn, dtype=torch.float32, device=dev)
    mr = torch.empty(2 * Cn, dtype=torch.float32, device=dev)
    with _guard(dev):
        _check(load().ww_dwconv_bn_act_fwd(ctx(dev), _p(x), _p(w), B, H, W, Cn, k, stride, C.byref(bn), act, _p(y), _p(a), _p(ss),
                                           _p(mr), _p(nhwc_scratch(Cn, dev)), _stream(dev)), "ww_dwconv_bn_act_fwd")
    return y, a, ss, mr


def dwconv_nhwc_fwd(x, w, k, stride):
    dev = _dev(x, w)
    B, H, W, Cn = x.shape
    Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    y = torch.empty((B, Ho, Wo, Cn), dtype=torch.float32, device=dev)
    with _guard(dev):
        _check(load().ww_dwconv_nhwc_fwd(ctx(dev), _p(x), _p(w), B, H, W, Cn, k, stride, _p(y), _stream(dev)), "ww_dwconv_nhwc_fwd")
    return y


def dwconv_nhwc_bwd(x, w, dy, k, stride, need_dx=True, dw_out=None, defer=False):
    """defer: queue the sum of the weight-gradient partials (dw valid after deferred_flush); the partials then get a buffer of
    their own instead of the per-(C, device) scratch the next layer call would overwrite."""
    dev = _dev(x, w, dy)
    B, H, W, Cn = x.shape
    dx = torch.empty_like(x) if need_dx else None
    dw = _out(dw_out, tuple(w.shape), dev)
    scratch = torch.empty(load().ww_nhwc_scratch_bytes(Cn) // 4, dtype=torch.float32, device=dev) if defer else nhwc_scratch(Cn, dev)
    with _guard(dev), (_deferring(dev, scratch) if defer else contextlib.nullcontext()):
        _check(load().ww_dwconv_nhwc_bwd(ctx(dev), _p(x), _p(w), _p(dy), B, H, W, Cn, k, stride, _p(dx), _p(dw),
                                         _p(scratch), _stream(dev)), "ww_dwconv_nhwc_bwd")
    return dx, dw


def pool_hw_fwd(x):
    dev = _dev(x)
    B, HW, Cn = x.shape
    s = torch.empty((B, Cn), dtype=torch.float32, device=dev)
    with _guard(dev):
        _check(load().ww_pool_hw_fwd(ctx(dev), _p(x), B, HW, Cn, _p(s), _stream(dev)), "ww_pool_hw_fwd")
    return s


def scale_bc_fwd(x, gate):
    dev = _dev(x, gate)
    B, HW, Cn = x.shape
    y = torch.empty_like(x)
    with _guard(dev):
        _check(load().ww_scale_bc_fwd(ctx(dev), _p(x), _p(gate), B, HW, Cn, _p(y), _stream(dev)), "ww_scale_bc_fwd")
    return y


def scale_bc_bwd_gate(x, dy):
    dev = _dev(x, dy)
    B, HW, Cn = x.shape
    dg = torch.empty((B, Cn), dtype=torch.float32, device=dev)
    with _guard(dev):
        _check(load().ww_scale_bc_bwd_gate(ctx(dev), _p(x), _p(dy), B, HW, Cn, _p(dg), _stream(dev)), "ww_scale_bc_bwd_gate")
    return dg


def scale_pool_bwd(dy, gate, dpool, shape):
    """dx (B,HW,C) = dy*gate + dpool/HW (either term may be absent)."""
    dev = _dev(dy, gate, dpool)
    B, HW, Cn = shape
    dx = torch.empty(shape, dtype=torch.float32, device=dev)
    with _guard(dev):
        _check(load().ww_scale_pool_bwd(ctx(dev), _p(dy), _p(gate), _p(dpool), B, HW, Cn, _p(dx), _stream(dev)),
               "ww_scale_pool_bwd")
    return dx


def se_supported(Cn, Cs, *tensors):
    """Shapes (and 16-byte alignment of the float4-read tensors) the one-launch squeeze-excitation kernels take -- ww_se_fwd's
    documented limits."""
    return (Cn % 4 == 0 and Cs % 4 == 0 and 4 <= Cn <= 1024 and 4 <= Cs <= 256
            and all(t.data_ptr() % 16 == 0 for t in tensors))


def se_fwd(x, w1, b1, w2, b2):
    """x (B,HW,C), w1 (Cs,C), w2 (C,Cs) -> (y, s (B,C), pre1 (B,Cs), pre2 (B,C)); one launch."""
    dev = _dev(x, w1, b1, w2, b2)
    B, HW, Cn = x.shape
    Cs = w1.shape[0]
    y = torch.empty_like(x)
    s = torch.empty((B, Cn), dtype=torch.float32, device=dev)
    pre1 = torch.empty((B, Cs), dtype=torch.float32, device=dev)
    pre2 = torch.empty((B, Cn), dtype=torch.float32, device=dev)
    with _guard(dev):
        _check(load().ww_se_fwd(ctx(dev), _p(x), B, HW, Cn, Cs, _p(w1), _p(b1), _p(w2), _p(b2), _p(y), _p(s), _p(pre1), _p(pre2),
                                _stream(dev)), "ww_se_fwd")
    return y, s, pre1, pre2


def se_bwd(x, dy, s, pre1, pre2, w1, w2, outs=(None, None, None, None)):
    """-> (dx, dw1, db1, dw2, db2); two launches.  outs: optional buffers for (dw1, db1, dw2, db2)."""
    dev = _dev(x, dy, s, pre1, pre2, w1, w2)
    B, HW, Cn = x.shape
    Cs = w1.shape[0]
    dx = torch.empty_like(x)
    dw1, dw2 = _out(outs[0], (Cs, Cn), dev), _out(outs[2], (Cn, Cs), dev)
    db1 = _out(outs[1], (Cs,), dev)
    db2 = _out(outs[3], (Cn,), dev)
    nbytes = load().ww_se_bwd_scratch_bytes(B, Cn, Cs)
    scratch = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    with _guard(dev):
        _check(load().ww_se_bwd(ctx(dev), _p(x), _p(dy), _p(s), _p(pre1), _p(pre2), _p(w1), _p(w2), B, HW, Cn, Cs, _p(dx), _p(dw1),
                                _p(db1), _p(dw2), _p(db2), _p(scratch), nbytes, _stream(dev)), "ww_se_bwd")
    return dx, dw1, db1, dw2, db2


def stem_supported(Cn):
    return Cn % 4 == 0 and 4 <= Cn <= 64


def stem3x3s2_bn_act_fwd(x, w, bn: BN, act):
    """x (B,H,W) one-channel images, w (C,1,3,3): direct 3x3 stride-2 convolution + training-mode BatchNorm + activation.
    -> (y pre-BN (B,Ho,Wo,C), a, ss, mr)."""
    dev = _dev(x, w)
    B, H, W = x.shape
    Cn = w.shape[0]
    y = torch.empty((B, (H + 1) // 2, (W + 1) // 2, Cn), dtype=torch.float32, device=dev)
    a = torch.empty_like(y)
    ss = torch.empty(2 * Cn, dtype=torch.float32, device=dev)
    mr = torch.empty(2 * Cn, dtype=torch.float32, device=dev)
    with _guard(dev):
        _check(load().ww_stem3x3s2_bn_act_fwd(ctx(dev), _p(x), _p(w), B, H, W, Cn, C.byref(bn), act, _p(y), _p(a), _p(ss), _p(mr),
                                              _p(nhwc_scratch(Cn, dev)), _stream(dev)), "ww_stem3x3s2_bn_act_fwd")
    return y, a, ss, mr


def stem3x3s2_bwd_dw(x, dy, w_shape, dw_out=None, defer=False):
    dev = _dev(x, dy)
    B, H, W = x.shape
    Cn = dy.shape[-1]
    dw = _out(dw_out, tuple(w_shape), dev)
    scratch = torch.empty(load().ww_nhwc_scratch_bytes(Cn) // 4, dtype=torch.float32, device=dev) if defer else nhwc_scratch(Cn, dev)
    with _guard(dev), (_deferring(dev, scratch) if defer else contextlib.nullcontext()):
        _check(load().ww_stem3x3s2_bwd_dw(ctx(dev), _p(x), _p(dy), B, H, W, Cn, _p(dw), _p(scratch), _stream(dev)),
               "ww_stem3x3s2_bwd_dw")
    return dw


def im2col3x3s2(x):
    """x (B,H,W) one-channel images -> (B*ceil(H/2)*ceil(W/2), 9) patches."""
    dev = _dev(x)
    B, H, W = x.shape
    cols = torch.empty((B * ((H + 1) // 2) * ((W + 1) // 2), 9), dtype=torch.float32, device=dev)
    with _guard(dev):
        _check(load().ww_im2col3x3s2(ctx(dev), _p(x), B, H, W, _p(cols), _stream(dev)), "ww_im2col3x3s2")
    return cols


def add_f32(a, b):
    dev = _dev(a, b)
    y = torch.empty_like(a)
    with _guard(dev):
        _check(load().ww_add_f32(ctx(dev), _p(a), _p(b), a.numel(), _p(y), _stream(dev)), "ww_add_f32")
    return y


def gru_workspace(B, T, I, H, dev):
    n = load().ww_gru_workspace_bytes(B, T, I, H)
    if n == 0:
        raise NativeError(f"GRU shape B={B} T={T} I={I} H={H} is not implemented (hidden size 128 only)")
    return torch.empty(n // 4, dtype=torch.float32, device=dev)


def gru_fwd(x, w_ih, w_hh, b_ih, b_hh, y, ws, h0=None, reverse=False, mode=torch.float32):
    """One GRU direction: x (B,T,I) -> writes y (B,T,H) (may be a column slice of a (B,T,2H) buffer); returns h_n (B,H)."""
    dev = _dev(w_ih, w_hh, b_ih, b_hh, ws, h0)
    _dev_rows(x, y)
    B, T, I = x.shape
    H = w_hh.shape[1]
    ldx, ldy = _bt_rows(x, "x"), _bt_rows(y, "y")
    if tuple(w_ih.shape) != (3 * H, I) or tuple(w_hh.shape) != (3 * H, H) or tuple(y.shape) != (B, T, H):
        raise ValueError("GRU parameter / output shapes do not match (w_ih (3H,I), w_hh (3H,H), y (B,T,H))")
    h_n = torch.empty((B, H), dtype=torch.float32, device=dev)
    with _guard(dev):
        _check(load().ww_gru_fwd(ctx(dev), act_code(mode), _p(x), ldx, _p(w_ih.contiguous()), _p(w_hh.contiguous()), _p(b_ih), _p(b_hh), _p(h0),
                                 B, T, I, H, int(reverse), _p(y), ldy, _p(h_n), _p(ws), ws.numel() * 4, _stream(dev)),
               "ww_gru_fwd")
    return h_n


def gru_bwd(x, w_ih, w_hh, dy, dh_n, ws, reverse=False, dx=None, accumulate_dx=False, want_dh0=False, mode=torch.float32):
    """Backward of the gru_fwd that filled ``ws``: -> (dw_ih, dw_hh, db_ih, db_hh, dh0 | None); dx written/accumulated in place."""
    dev = _dev(w_ih, w_hh, dh_n, ws)
    _dev_rows(x, dy, dx)
    B, T, I = x.shape
    H = w_hh.shape[1]
    ldx = _bt_rows(x, "x")
    ldy = _bt_rows(dy, "dy") if dy is not None else H
    lddx = _bt_rows(dx, "dx") if dx is not None else I
    dw_ih, dw_hh = torch.empty_like(w_ih), torch.empty_like(w_hh)
    db_ih = torch.empty(3 * H, dtype=torch.float32, device=dev)
    db_hh = torch.empty(3 * H, dtype=torch.float32, device=dev)
    dh0 = torch.empty((B, H), dtype=torch.float32, device=dev) if want_dh0 else None
    with _guard(dev):
        _check(load().ww_gru_bwd(ctx(dev), act_code(mode), _p(x), ldx, _p(w_ih.contiguous()), _p(w_hh.contiguous()), _p(dy), ldy, _p(dh_n),
                                 B, T, I, H, int(reverse), _p(ws), ws.numel() * 4, _p(dx), lddx, int(accumulate_dx),
                                 _p(dw_ih), _p(dw_hh), _p(db_ih), _p(db_hh), _p(dh0), _stream(dev)), "ww_gru_bwd")
    return dw_ih, dw_hh, db_ih, db_hh, dh0


def _ptr(t):
    return None if t is None else t.data_ptr()


def gru_bidir_fwd(x, params, y, ws, mode=torch.float32):
    """Both directions of a bidirectional layer, ONE recurrent launch: x (B,T,I); params = [(w_ih, w_hh, b_ih, b_hh)] x 2
    (forward, reverse); y (B,T,2H) written; ws = two gru_workspace tensors.  -> [h_n forward, h_n reverse], each (B,H)."""
    dev = _dev(*params[0], *params[1], ws[0], ws[1])
    _dev_rows(x, y)
    B, T, I = x.shape
    H = params[0][1].shape[1]
    ldx, ldy = _bt_rows(x, "x"), _bt_rows(y, "y")
    if tuple(y.shape) != (B, T, 2 * H) or any(tuple(p[0].shape) != (3 * H, I) or tuple(p[1].shape) != (3 * H, H) for p in params):
        raise ValueError("GRU parameter / output shapes do not match (w_ih (3H,I), w_hh (3H,H), y (B,T,2H))")
    h_n = [torch.empty((B, H), dtype=torch.float32, device=dev) for _ in range(2)]
    keep = [[t.contiguous() for t in p] for p in params]
    dirs = (GruDir * 2)()
    for k in range(2):
        dirs[k].w_ih, dirs[k].w_hh, dirs[k].b_ih, dirs[k].b_hh = (t.data_ptr() for t in keep[k])
        dirs[k].h_n, dirs[k].ws = h_n[k].data_ptr(), ws[k].data_ptr()
    with _guard(dev):
        _check(load().ww_gru_bidir_fwd(ctx(dev), act_code(mode), _p(x), ldx, C.byref(dirs), B, T, I, H, _p(y), ldy,
                                       min(ws[0].numel(), ws[1].numel()) * 4, _stream(dev)), "ww_gru_bidir_fwd")
    return h_n


def gru_bidir_bwd(x, params, dy, dh_n, ws, dx=None, mode=torch.float32, outs=None, defer=False):
    """Backward of gru_bidir_fwd: dy (B,T,2H) or None, dh_n = [(B,H) | None] x 2; dx (B,T,I) written when given.
    -> [(dw_ih, dw_hh, db_ih, db_hh)] x 2.  outs: the same structure of tensors to write the gradients into (bucket slots);
    defer: queue the sums of the weight-gradient / bias partials (valid after deferred_flush; ``ws`` is kept until then)."""
    dev = _dev(params[0][0], params[1][0], ws[0], ws[1])
    _dev_rows(x, dy, dx)
    B, T, I = x.shape
    H = params[0][1].shape[1]
    ldx = _bt_rows(x, "x")
    ldy = _bt_rows(dy, "dy") if dy is not None else 2 * H
    lddx = _bt_rows(dx, "dx") if dx is not None else I
    keep = [[p[0].contiguous(), p[1].contiguous()] for p in params]
    if outs is None:
        outs = [(None, None, None, None)] * 2
    grads = [(_out(o[0], tuple(p[0].shape), dev), _out(o[1], tuple(p[1].shape), dev), _out(o[2], (3 * H,), dev),
              _out(o[3], (3 * H,), dev)) for p, o in zip(params, outs)]
    dirs = (GruDir * 2)()
    for k in range(2):
        dirs[k].w_ih, dirs[k].w_hh = keep[k][0].data_ptr(), keep[k][1].data_ptr()
        dirs[k].ws, dirs[k].dh_n = ws[k].data_ptr(), _ptr(dh_n[k])
        dirs[k].dw_ih, dirs[k].dw_hh, dirs[k].db_ih, dirs[k].db_hh = (g.data_ptr() for g in grads[k])
    with _guard(dev), (_deferring(dev, ws[0], ws[1]) if defer else contextlib.nullcontext()):
        _check(load().ww_gru_bidir_bwd(ctx(dev), act_code(mode), _p(x), ldx, C.byref(dirs), _p(dy), ldy, B, T, I, H,
                                       min(ws[0].numel(), ws[1].numel()) * 4, _p(dx), lddx, _stream(dev)), "ww_gru_bidir_bwd")
    return grads


def step_ctl_new(dev, step=0, lr=0.0, parity=0):
    """A device control block (uint8[32] tensor) initialised to (step, lr, parity); see include/wwhip.h ww_step_ctl."""
    host = StepCtl(step, lr, parity, 1.0, 0)
    return torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).to(dev)


def step_ctl_write(ctl, step=None, lr=None, parity=None):
    """Update fields of a control block from the host (small async copies on the current stream, between replays)."""
    if step is not None:
        ctl[0:8].copy_(torch.frombuffer(bytearray(int(step).to_bytes(8, "little")), dtype=torch.uint8), non_blocking=False)
    if lr is not None:
        ctl[8:12].view(torch.float32).fill_(float(lr))
    if parity is not None:
        ctl[12:16].view(torch.int32).fill_(int(parity))


def step_ctl_read(ctl) -> dict:
    s = StepCtl.from_buffer_copy(bytes(ctl.cpu().numpy().tobytes()))
    return {"step": s.step, "lr": s.lr, "parity": s.parity, "loss_scale": s.loss_scale, "growth_tracker": s.growth_tracker}


def set_logmel_workgroups(dev, n: int):
    """Persistent workgroups of the log-mel kernel for the launches that follow on ``dev``: 0 = fill the device (the front
    end alone), n > 0 = at most n (a front end running beside a training step; results do not depend on it)."""
    _check(load().ww_ctx_set_logmel_workgroups(ctx(dev), int(n)), "ww_ctx_set_logmel_workgroups")


def bind_step_ctl(dev, ctl):
    """Bind (ctl = uint8[32] device tensor) or unbind (None) the device-resident step control of ``dev``'s context."""
    if ctl is not None and (not ctl.is_cuda or ctl.numel() < STEP_CTL_BYTES or ctl.dtype != torch.uint8):
        raise ValueError("the step control block must be a uint8 device tensor of at least 32 bytes")
    _check(load().ww_ctx_bind_step_ctl(ctx(dev), _p(ctl)), "ww_ctx_bind_step_ctl")


def step_ctl_advance(dev):
    with _guard(dev):
        _check(load().ww_step_ctl_advance(ctx(dev), _stream(dev)), "ww_step_ctl_advance")


def clip_optim_step_(cfg: OptimCfg, flat_params, flat_grads, exp_avg, exp_avg_sq, step_state, parity, norm_out=None,
                     stats=None, stats_host=None, found_inf_extra=None, stats_host_alt=None, loss_scale=None):
    """In place: clip flat_grads to cfg.max_norm, then one Adam/AdamW/SGD step on flat_params (skipped on found_inf).
    ``stats_host``: pinned uint8[48] host tensor the kernel copies the step's ww_step_stats into.
    ``found_inf_extra``: float32[1] device tensor; non-zero also skips the step (data parallel: some rank's bad batch)."""
    dev = _dev(flat_params, flat_grads, exp_avg, exp_avg_sq, step_state, norm_out, stats, found_inf_extra, loss_scale)
    if stats_host is not None and not (stats_host.is_pinned() and stats_host.numel() * stats_host.element_size() >= STEP_STATS_BYTES):
        raise ValueError("stats_host must be a pinned host tensor of at least 48 bytes")
    if flat_params.dtype != torch.float32 or flat_grads.dtype != torch.float32 or flat_params.numel() != flat_grads.numel():
        raise ValueError("parameter and gradient buckets must be float32 and of equal length")
    if step_state.dtype != torch.int64 or step_state.numel() != 2:
        raise ValueError("step_state must be an int64 tensor of two elements")
    with _guard(dev):
        _check(load().ww_clip_optim_step(ctx(dev), C.byref(cfg), _p(flat_params), _p(flat_grads), _p(exp_avg), _p(exp_avg_sq),
                                         flat_params.numel(), _p(step_state), parity, _p(norm_out), _p(stats),
                                         None if stats_host is None else C.c_void_p(stats_host.data_ptr()),
                                         None if stats_host_alt is None else C.c_void_p(stats_host_alt.data_ptr()),
                                         _p(found_inf_extra), _p(loss_scale), _stream(dev)), "ww_clip_optim_step")


def layer_scratch(dev):
    return torch.empty(load().ww_layer_scratch_bytes() // 4, dtype=torch.float32, device=dev)


def make_bn(gamma, beta, running_mean, running_var, momentum=0.1, eps=1e-5, training=True):
    return BN(_p(gamma), _p(beta), _p(running_mean), _p(running_var), momentum, eps, 1 if training else 0)


def conv_stem_fwd(x, w, bn: BN, scratch, act=torch.float32):
    dev = _dev(x, w, scratch)
    B, Hin, Win = x.shape[0], x.shape[-2], x.shape[-1]
    Ho, Wo = (Hin + 1) // 2, (Win + 1) // 2
    y = torch.empty((B, Ho, Wo, 64), dtype=act, device=dev)
    ss = torch.empty(128, dtype=torch.float32, device=dev)
    mr = torch.empty(128, dtype=torch.float32, device=dev)
    with _guard(dev):
        _check(load().ww_conv_stem_fwd(ctx(dev), act_code(act), _p(x), _p(w), B, Hin, Win, _p(y), C.byref(bn), _p(ss), _p(mr),
                                       _p(scratch), _stream(dev)), "ww_conv_stem_fwd")
    return y, ss, mr


def _conv_fwd(name, y_in, ss_in, w, bn, scratch):
    dev = _dev(y_in, ss_in, w, scratch)
    B, H, W, Cc = y_in.shape
    if Cc != 64:
        raise ValueError("conv stack width is 64")
    y = torch.empty_like(y_in)
    ss = torch.empty(128, dtype=torch.float32, device=dev)
    mr = torch.empty(128, dtype=torch.float32, device=dev)
    with _guard(dev):
        _check(getattr(load(), name)(ctx(dev), act_code(y_in.dtype), _p(y_in), _p(ss_in), _p(w), B, H, W, _p(y), C.byref(bn), _p(ss), _p(mr),
                                     _p(scratch), _stream(dev)), name)
    return y, ss, mr


def dwconv3x3_fwd(y_in, ss_in, w, bn, scratch):
    return _conv_fwd("ww_dwconv3x3_fwd", y_in, ss_in, w, bn, scratch)


def pwconv1x1_fwd(y_in, ss_in, w, bn, scratch):
    return _conv_fwd("ww_pwconv1x1_fwd", y_in, ss_in, w, bn, scratch)


def gap_fwd(y, ss, mr):
    dev = _dev(y, ss, mr)
    B, H, W, _ = y.shape
    pool = torch.empty((B, 3, 64), dtype=torch.float32, device=dev)
    with _guard(dev):
        _check(load().ww_gap_fwd(ctx(dev), act_code(y.dtype), _p(y), _p(ss), _p(mr), B, H, W, _p(pool), _stream(dev)),
               "ww_gap_fwd")
    return pool


def head_fwd(pool, HW, fc_w, fc_b, dropout_p=0.0, training=True, seed=0, step=0, sample_offset=0):
    dev = _dev(pool, fc_w, fc_b)
    B = pool.shape[0]
    pd = torch.empty((B, 64), dtype=torch.float32, device=dev)
    logits = torch.empty((B, 2), dtype=torch.float32, device=dev)
    with _guard(dev):
        _check(load().ww_head_fwd(ctx(dev), _p(pool), B, HW, _p(fc_w), _p(fc_b), dropout_p, int(training), seed, step,
                                  sample_offset, _p(pd), _p(logits), _stream(dev)), "ww_head_fwd")
    return pd, logits


def head_bwd(dlogits, pd, pool, HW, fc_w, gamma_last, mr_last, dropout_p=0.0, training=True, seed=0, step=0,
             sample_offset=0):
    dev = _dev(dlogits, pd, pool, fc_w, gamma_last, mr_last)
    B = pool.shape[0]
    f32 = dict(dtype=torch.float32, device=dev)
    dfc_w, dfc_b = torch.empty((2, 64), **f32), torch.empty(2, **f32)
    dpool, coef = torch.empty((B, 64), **f32), torch.empty(192, **f32)
    dgamma, dbeta = torch.empty(64, **f32), torch.empty(64, **f32)
    with _guard(dev):
        _check(load().ww_head_bwd(ctx(dev), _p(dlogits), _p(pd), _p(pool), B, HW, _p(fc_w), dropout_p, int(training),
                                  seed, step, sample_offset, _p(gamma_last), _p(mr_last), _p(dfc_w), _p(dfc_b),
                                  _p(dpool), _p(coef), _p(dgamma), _p(dbeta), _stream(dev)), "ww_head_bwd")
    return dfc_w, dfc_b, dpool, coef, dgamma, dbeta


def pwconv1x1_bwd(g, dpool, y_out, ss_out, coef, y_in, ss_in, mr_in, gamma_in, w, scratch):
    dev = _dev(g, dpool, y_out, ss_out, coef, y_in, ss_in, mr_in, gamma_in, w, scratch)
    B, H, W, _ = y_out.shape
    f32 = dict(dtype=torch.float32, device=dev)
    g_in, dw = torch.empty_like(y_in), torch.empty((64, 64), **f32)
    coef_in, dgamma, dbeta = torch.empty(192, **f32), torch.empty(64, **f32), torch.empty(64, **f32)
    with _guard(dev):
        _check(load().ww_pwconv1x1_bwd(ctx(dev), act_code(y_out.dtype), _p(g), _p(dpool), _p(y_out), _p(ss_out), _p(coef), _p(y_in), _p(ss_in),
                                       _p(mr_in), _p(gamma_in), _p(w), B, H, W, _p(g_in), _p(dw), _p(coef_in),
                                       _p(dgamma), _p(dbeta), _p(scratch), _stream(dev)), "ww_pwconv1x1_bwd")
    return g_in, dw, coef_in, dgamma, dbeta


def dwconv3x3_bwd(g, y_out, coef, y_in, ss_in, mr_in, gamma_in, w, scratch):
    dev = _dev(g, y_out, coef, y_in, ss_in, mr_in, gamma_in, w, scratch)
    B, H, W, _ = y_out.shape
    f32 = dict(dtype=torch.float32, device=dev)
    g_in, dw = torch.empty_like(y_in), torch.empty((64, 1, 3, 3), **f32)
    coef_in, dgamma, dbeta = torch.empty(192, **f32), torch.empty(64, **f32), torch.empty(64, **f32)
    with _guard(dev):
        _check(load().ww_dwconv3x3_bwd(ctx(dev), act_code(y_out.dtype), _p(g), _p(y_out), _p(coef), _p(y_in), _p(ss_in), _p(mr_in),
                                       _p(gamma_in), _p(w), B, H, W, _p(g_in), _p(dw), _p(coef_in), _p(dgamma),
                                       _p(dbeta), _p(scratch), _stream(dev)), "ww_dwconv3x3_bwd")
    return g_in, dw, coef_in, dgamma, dbeta


def conv_stem_bwd(g, y_out, coef, x, scratch):
    dev = _dev(g, y_out, coef, x, scratch)
    B, Hin, Win = x.shape[0], x.shape[-2], x.shape[-1]
    dw = torch.empty((64, 1, 3, 3), dtype=torch.float32, device=dev)
    with _guard(dev):
        _check(load().ww_conv_stem_bwd(ctx(dev), act_code(y_out.dtype), _p(g), _p(y_out), _p(coef), _p(x), B, Hin, Win, _p(dw), _p(scratch),
                                       _stream(dev)), "ww_conv_stem_bwd")
    return dw


def cnn_small_workspace_bytes(B, F, T, act=ACT_F32):
    return load().ww_cnn_small_workspace_bytes(B, F, T, act)


def ptr_array(tensors):
    arr = (_vp * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr


def cnn_small_fwd(params, x, ws, logits, training, bn_momentum=0.1, bn_eps=1e-5, dropout_p=0.0, seed=0, step=0,
                  sample_offset=0, act=ACT_F32):
    dev = _dev(x, ws, logits)
    B, F, T = x.shape[0], x.shape[-2], x.shape[-1]
    with _guard(dev):
        _check(load().ww_cnn_small_fwd(ctx(dev), act, params, _p(x), B, F, T, int(training), bn_momentum, bn_eps, dropout_p,
                                       seed, step, sample_offset, _p(ws), ws.numel() * ws.element_size(), _p(logits),
                                       _stream(dev)), "ww_cnn_small_fwd")


def cnn_small_bwd(params, grads, x, dlogits, ws, dropout_p=0.0, seed=0, step=0, sample_offset=0, act=ACT_F32,
                  part=BWD_ALL):
    dev = _dev(x, ws, dlogits)
    B, F, T = x.shape[0], x.shape[-2], x.shape[-1]
    with _guard(dev):
        _check(load().ww_cnn_small_bwd(ctx(dev), act, params, grads, _p(x), _p(dlogits), B, F, T, dropout_p, seed, step,
                                       sample_offset, _p(ws), ws.numel() * ws.element_size(), part, _stream(dev)),
               "ww_cnn_small_bwd")


def cnn_front_fwd(params, x, ws, seq, training, bn_momentum=0.1, bn_eps=1e-5, act=ACT_F32):
    dev = _dev(x, ws, seq)
    B, F, T = x.shape[0], x.shape[-2], x.shape[-1]
    with _guard(dev):
        _check(load().ww_cnn_front_fwd(ctx(dev), act, params, _p(x), B, F, T, int(training), bn_momentum, bn_eps, _p(ws),
                                       ws.numel() * ws.element_size(), _p(seq), _stream(dev)), "ww_cnn_front_fwd")


def cnn_front_bwd(params, grads, x, dseq, ws, act=ACT_F32):
    dev = _dev(x, ws, dseq)
    B, F, T = x.shape[0], x.shape[-2], x.shape[-1]
    with _guard(dev):
        _check(load().ww_cnn_front_bwd(ctx(dev), act, params, grads, _p(x), _p(dseq), B, F, T, _p(ws),
                                       ws.numel() * ws.element_size(), _stream(dev)), "ww_cnn_front_bwd")


def ce2_loss_fwd_bwd(logits, targets, kind=LOSS_CE, label_smoothing=0.0, focal_alpha=0.25, focal_gamma=2.0,
                     stats=None, found_inf_out=None, loss_scale=None, loss_scale_slot=0):
    """-> (loss (1,) f32, dlogits (B,2) f32, stats uint8[STEP_STATS_BYTES]).  ``found_inf_out``: float32[1] device tensor
    that also receives the step's found_inf flag (the spare slot of a data-parallel gradient bucket)."""
    dev = _dev(logits, targets, stats, found_inf_out, loss_scale)
    if logits.dim() != 2 or logits.shape[1] != 2 or logits.dtype != torch.float32:
        raise ValueError(f"native loss needs float32 logits of shape (B,2), got {logits.dtype} {tuple(logits.shape)}")
    if targets.dim() != 1 or targets.shape[0] != logits.shape[0] or targets.dtype != torch.int64:
        raise ValueError("targets must be int64 of shape (B,)")
    B = logits.shape[0]
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    dl = torch.empty_like(logits)
    if stats is None:
        stats = torch.zeros(STEP_STATS_BYTES, dtype=torch.uint8, device=dev)
    with _guard(dev):
        _check(load().ww_ce2_loss_fwd_bwd(ctx(dev), _p(logits), _p(targets), B, kind, label_smoothing, focal_alpha,
                                          focal_gamma, _p(loss), _p(dl), _p(stats), _p(found_inf_out), _p(loss_scale),
                                          int(loss_scale_slot), _stream(dev)),
               "ww_ce2_loss_fwd_bwd")
    return loss, dl, stats


def grad_norm_clip_(flat, max_norm, norm_out=None, stats=None):
    """clip_grad_norm_ on a flat bucket.  `stats` (uint8[STEP_STATS_BYTES], optional): its grad_norm / found_inf
    fields are updated on the device."""
    dev = _dev(flat, norm_out, stats)
    if norm_out is None and stats is None:
        norm_out = torch.empty(1, dtype=torch.float32, device=dev)
    with _guard(dev):
        _check(load().ww_grad_norm_clip(ctx(dev), _p(flat), flat.numel(), float(max_norm), _p(norm_out), _p(stats),
                                        _stream(dev)), "ww_grad_norm_clip")
    return norm_out


FOUND_INF_FLOAT_INDEX = StepStats.found_inf.offset // 4


def prof_classes():
    lib = load()
    return [lib.ww_prof_class_name(i).decode() for i in range(lib.ww_prof_num_classes())]


def prof_enable(dev, names=None):
    """Time the given kernel classes (None = all, [] = off) with HIP events on the launch stream."""
    classes = prof_classes()
    mask = 0
    for i, n in enumerate(classes):
        if names is None or n in names:
            mask |= 1 << i
    _check(load().ww_prof_enable(ctx(dev), mask), "ww_prof_enable")


def prof_collect(dev):
    """-> {class: (total_ms, launches)} since the last collect (waits for the recorded events)."""
    classes = prof_classes()
    ms = (C.c_float * len(classes))()
    cnt = (C.c_int32 * len(classes))()
    _check(load().ww_prof_collect(ctx(dev), ms, cnt), "ww_prof_collect")
    return {n: (float(ms[i]), int(cnt[i])) for i, n in enumerate(classes) if cnt[i]}


def decode_stats(stats_cpu: torch.Tensor) -> dict:
    s = StepStats.from_buffer_copy(bytes(stats_cpu.numpy().tobytes()))
    return {k: getattr(s, k) for k, _ in StepStats._fields_}
