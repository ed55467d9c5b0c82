"""The optimizer half of the reference's `--fp16` training step (README.md:241,274; BASELINE C5), for bench.py and the C5 tests.

fairseq trains DASpeech with a HALF-PRECISION MODEL (model.half(), trainer.py:_setup / fp16) under FP16Optimizer
(fairseq/fairseq/optim/fp16_optimizer.py:252-330 with `build_fp32_params(flatten=True)` :38-72): one flat fp32 master copy of all
parameters, fp16 gradients copied into a flat fp32 gradient (:111-145), multiply_grads / clip_grad_norm on that ONE tensor (:187-211), Adam on
that one tensor, master copied back into the fp16 parameters (:146-170), loss scale from DynamicLossScaler
(fairseq/fairseq/optim/dynamic_loss_scaler.py:7-70: x2 after `scale_window` overflow-free updates, /2 on an overflow, which skips the update).
torch.autocast — what r01-r03 timed — is not what the reference does and costs ~1 500 cast launches per step on this model.

Here: the same scheme with multi-tensor copies (torch._foreach_copy_) instead of fairseq's per-parameter Python loops."""
from typing import Iterable

import torch


class DynamicLossScaler:
    """dynamic_loss_scaler.py:7-70 (tolerance 0, no threshold)."""

    def __init__(self, init_scale=2.0 ** 7, scale_factor=2.0, scale_window=2000, min_loss_scale=1e-4):
        self.loss_scale, self.scale_factor, self.scale_window, self.min_loss_scale = float(init_scale), scale_factor, scale_window, min_loss_scale
        self._iter, self._last_overflow_iter = 0, -1

    def update(self):
        if (self._iter - self._last_overflow_iter) % self.scale_window == 0:
            self.loss_scale *= self.scale_factor
        self._iter += 1

    def overflow(self):
        self._last_overflow_iter = self._iter
        self.loss_scale /= self.scale_factor
        self._iter += 1
        if self.loss_scale <= self.min_loss_scale:
            raise FloatingPointError(f"Minimum loss scale reached ({self.min_loss_scale})")


class FP16FlatOptimizer:
    """fp16 parameters / gradients, flat fp32 master + Adam (fairseq's `adam`: decoupled weight decay, fairseq/optim/adam.py:195-198)."""

    def __init__(self, params: Iterable[torch.nn.Parameter], lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, clip_norm=1.0,
                 init_scale=2.0 ** 7, scale_window=2000):
        self.params = [p for p in params if p.requires_grad]
        assert self.params and all(p.dtype == torch.float16 for p in self.params), "FP16FlatOptimizer wants a model.half() model"
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.master = torch.nn.Parameter(torch.empty(n, dtype=torch.float32, device=dev))
        self.master.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        sizes = [p.numel() for p in self.params]
        self._mviews = [v.view_as(p) for v, p in zip(self.master.data.split(sizes), self.params)]
        self._gviews = [v.view_as(p) for v, p in zip(self.master.grad.split(sizes), self.params)]
        torch._foreach_copy_(self._mviews, [p.data for p in self.params])
        self.opt = torch.optim.AdamW([self.master], lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, fused=True)
        self.scaler = DynamicLossScaler(init_scale=init_scale, scale_window=scale_window)
        self.clip_norm = clip_norm
        self.last_grad_norm = None

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def backward(self, loss: torch.Tensor):
        (loss.float() * self.scaler.loss_scale).backward()                                   # fp16_optimizer.py:101-109

    def step(self, grad_mult: float = 1.0) -> bool:
        """Unscale (x grad_mult: the trainer's world_size / sample_size, trainer.py:932-946), clip, update.  Returns False when the step
        was skipped on an overflow (fairseq raises OverflowError and the trainer moves on, trainer.py:1020-1028)."""
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in self.params]
        torch._foreach_copy_(self._gviews, grads)                                            # :111-145, one multi-tensor launch
        g = self.master.grad
        g.mul_(grad_mult / self.scaler.loss_scale)                                           # :172-186
        norm = torch.linalg.vector_norm(g)
        nv = float(norm)                                                                     # (fairseq syncs here too: the norm is logged and checked, :191-211)
        self.last_grad_norm = nv
        if nv != nv or nv == float("inf"):
            self.scaler.overflow()
            return False
        if self.clip_norm and nv > self.clip_norm:
            g.mul_(self.clip_norm / (nv + 1e-6))
        self.opt.step()
        with torch.no_grad():                                                                # :146-170 — into the parameters THEMSELVES, so that
            torch._foreach_copy_(self.params, self._mviews)                                  # p._version advances (weight caches key on it)
        self.scaler.update()
        return True


def half_sample(sample):
    """trainer.py:_prepare_sample under --fp16: every float32 tensor of the batch becomes float16 (utils.apply_to_sample(apply_half))."""
    if isinstance(sample, dict):
        return {k: half_sample(v) for k, v in sample.items()}
    if torch.is_tensor(sample) and sample.dtype == torch.float32:
        return sample.half()
    return sample
