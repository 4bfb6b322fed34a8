"""Flat fp32 parameter / gradient buckets for HIP-backed models whose backward runs through autograd.

``CNNSmallWakeword`` builds its own buckets (its backward writes gradients straight into them).  Models composed of
autograd nodes (``MobileNetV3Wakeword``) get the same contract from this mixin -- ``flat_param``, ``flat_grad``,
``grads_in_bucket()`` -- so ``create_optimizer`` hands them the fused clip+optimizer kernel (``FlatFusedOptimizer``) and
data-parallel training all-reduces ONE tensor instead of one per parameter: at ~140 parameter tensors
``clip_grad_norm_`` + the torch optimizer cost 1.5 ms of host time per step."""
import weakref

import torch


def grad_slot(param):
    """A fresh view of ``param``'s slot in its model's flat gradient bucket, for a backward kernel to write into and RETURN as the
    gradient -- or None (no bucket, or the parameter already holds a gradient that this backward must be added to).  autograd
    adopts a returned gradient tensor as ``param.grad`` without copying when nothing else references it and its layout is the
    parameter's, so the gradient is born in the bucket and ``gather_grads`` has nothing to move (it was three multi-tensor copy
    launches per step, 142 tensors)."""
    slot = getattr(param, "_ww_grad_slot", None)
    if slot is None or param.grad is not None:
        return None
    ref, idx, off = slot
    mod = ref()
    # the slot must still be THIS parameter's (a copied attribute, a rebuilt or dropped bucket: no slot)
    if mod is None or mod._fb_plist is None or idx >= len(mod._fb_plist) or mod._fb_plist[idx] is not param:
        return None
    return mod._fb_grad[off:off + param.numel()].view_as(param)


class FlatBuckets:
    """Mixin for an nn.Module (list it BEFORE nn.Module).  Parameters become views of one flat fp32 tensor in
    ``parameters()`` order, built lazily on first use and rebuilt after ``.to()`` / ``.cuda()`` (``_apply``)."""

    _fb_param = None
    _fb_grad = None
    _fb_grad_ext = None
    _fb_plist = None
    _fb_views = None

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._fb_param = self._fb_grad = self._fb_grad_ext = self._fb_plist = self._fb_views = None
        return r

    def _fb_build(self):
        plist = list(self.parameters())
        if not plist:
            raise ValueError("FlatBuckets: the module has no parameters")
        dev = plist[0].device
        for t in plist:
            if t.dtype != torch.float32 or t.device != dev:
                raise ValueError("FlatBuckets: parameters must be float32 on one device")
        flat = torch.empty(sum(t.numel() for t in plist), dtype=torch.float32, device=dev)
        off = 0
        with torch.no_grad():
            for t in plist:
                view = flat[off:off + t.numel()].view_as(t)
                view.copy_(t.data)
                t.data = view                       # state_dict() / load_state_dict() go through the views
                off += t.numel()
        self._fb_param, self._fb_plist = flat, plist
        # one spare float behind the gradients: the data-parallel found_inf slot (see CNNSmallWakeword._prepare)
        self._fb_grad_ext = torch.zeros(flat.numel() + 1, dtype=torch.float32, device=dev)
        self._fb_grad = self._fb_grad_ext[:-1]
        views, off = [], 0
        for t in plist:
            views.append(self._fb_grad[off:off + t.numel()].view_as(t))
            t._ww_grad_slot = (weakref.ref(self), len(views) - 1, off)      # see grad_slot()
            off += t.numel()
        self._fb_views = views

    def _fb_ready(self):
        # rebuilt when missing or when the parameters no longer sit in the bucket (copy.deepcopy copies tensors one by one)
        if self._fb_param is None or self._fb_plist[0].data_ptr() != self._fb_param.data_ptr():
            self._fb_build()

    @property
    def flat_param(self):
        self._fb_ready()
        return self._fb_param

    @property
    def flat_grad(self):
        self._fb_ready()
        return self._fb_grad

    @property
    def flat_grad_ext(self):
        """``flat_grad`` plus the trailing data-parallel found_inf slot."""
        self._fb_ready()
        return self._fb_grad_ext

    def grads_in_bucket(self) -> bool:
        """autograd hands every parameter a fresh gradient tensor: they are gathered (``gather_grads``), never in place."""
        return False

    @torch.no_grad()
    def gather_grads(self):
        """Copy every ``.grad`` into its slot of ``flat_grad`` (one multi-tensor launch).  Every parameter must have a
        gradient: the fused optimizer updates the whole bucket (decay and moments included), where torch.optim would skip a
        parameter without one -- silently treating it as a zero gradient would make the two differ."""
        self._fb_ready()
        missing = [i for i, p in enumerate(self._fb_plist) if p.grad is None]
        if missing:
            raise RuntimeError(f"gather_grads: {len(missing)} parameter(s) have no gradient (first index {missing[0]}); frozen or "
                               "unused parameters are not supported by the bucketed optimizer step")
        todo = [(v, p.grad) for v, p in zip(self._fb_views, self._fb_plist) if p.grad.data_ptr() != v.data_ptr()]
        if todo:                                    # gradients the backward kernels wrote straight into their slots need no copy
            torch._foreach_copy_([v for v, _ in todo], [g for _, g in todo])
        return self._fb_grad
