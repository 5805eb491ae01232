"""AdamW over the C-ABI fused multi-tensor kernel (ococc_adamw_f32).

Same constructor arguments and update rule as torch.optim.AdamW, the optimizer the reference
configures (configs/_base_/schedules/cosine_2x.py:2-8, lr at configs/ococc/ococcnet.py:468-470;
amsgrad / maximize are not used there and not offered here).  One launch per parameter group per
48 tensors; the step counter is a device float, so ``step()`` has no host dependency and can be
recorded into a HIP graph (graph.GraphedStep)."""
import ctypes

import torch

from . import _lib as L

_MAX = 48


class AdamW(torch.optim.Optimizer):

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError('invalid AdamW hyper-parameter')
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    def _init_group(self, group):
        for p in group['params']:
            st = self.state[p]
            if not st:
                st['exp_avg'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        if 'step_dev' not in group:
            dev = group['params'][0].device
            group['step_dev'] = torch.zeros(2, dtype=torch.float32, device=dev)  # {count, ticket of the kernel}

    def init_state(self):
        """Allocate the moment buffers and the device step counter now (call before capturing
        ``step()`` in a HIP graph: a capture must not contain their zero fill)."""
        for group in self.param_groups:
            if group['params']:
                self._init_group(group)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            if not group['params']:
                continue
            self._init_group(group)
            ps = [p for p in group['params'] if p.grad is not None]
            for p in ps:
                L.require_device(p, p.grad)
                if p.dtype != torch.float32 or p.grad.dtype != torch.float32:
                    raise L.OcoccError('AdamW kernel takes float32 parameters and gradients')
                if not (p.is_contiguous() and p.grad.is_contiguous()):
                    raise L.OcoccError('AdamW kernel takes contiguous parameters and gradients')
            b1, b2 = group['betas']
            # every launch of one step must see the same step count: only the last one bumps it
            for lo in range(0, len(ps), _MAX):
                chunk = ps[lo:lo + _MAX]
                n = len(chunk)
                arr = ctypes.c_void_p * n
                last = lo + _MAX >= len(ps)
                L.check(L.lib.ococc_adamw_f32(
                    n, arr(*[p.data_ptr() for p in chunk]), arr(*[p.grad.data_ptr() for p in chunk]),
                    arr(*[self.state[p]['exp_avg'].data_ptr() for p in chunk]),
                    arr(*[self.state[p]['exp_avg_sq'].data_ptr() for p in chunk]),
                    (ctypes.c_int64 * n)(*[p.numel() for p in chunk]), float(group['lr']), float(b1),
                    float(b2), float(group['eps']), float(group['weight_decay']), group['step_dev'].data_ptr(),
                    2 if last else 0, L.stream()), 'adamw')
            for p in ps:  # the kernel wrote through raw pointers: tell autograd / version-keyed caches
                torch.autograd.graph.increment_version(p)
        return loss
