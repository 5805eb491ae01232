"""AdamW over the C-ABI fused multi-tensor kernel (ococc_adamw_f32).

Same constructor arguments and update rule as torch.optim.AdamW, the optimizer the reference
configures (configs/_base_/schedules/cosine_2x.py:2-8, lr at configs/ococc/ococcnet.py:468-470;
amsgrad / maximize are not used there and not offered here).  One launch per parameter group per
48 tensors; the step counter is a device float, so ``step()`` has no host dependency and can be
recorded into a HIP graph (graph.GraphedStep).  ``device_lr=True`` keeps each group's learning rate in a device
scalar as well: ``set_lr()`` (or a schedule) updates it between steps, graph replays included.

``param_groups_from_cfg`` and ``cyclic_lr`` restate the two pieces of mmcv the reference's schedule relies on
(configs/_base_/schedules/cosine_2x.py:2-15): DefaultOptimizerConstructor's ``paramwise_cfg.custom_keys`` and
CyclicLrUpdaterHook (un-vendored mmcv-full 1.3.8-1.4.0, requirements/mminstall.txt)."""
import ctypes

import torch

from . import _lib as L

REFRESH_CONV_OPERANDS = __import__('os').environ.get('OCOCC_ADAMW_OPERANDS', '1') == '1'   # see AdamW.step
_MAX = 48
# who bumps the device-side step count: 2 = the last workgroup of the update launch (a ticket atomic per workgroup),
# 1 = a one-thread launch behind it (profiling: what the tickets cost)
_BUMP = int(__import__('os').environ.get('OCOCC_ADAMW_BUMP', '2'))


class AdamW(torch.optim.Optimizer):

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, device_lr=False):
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError('invalid AdamW hyper-parameter')
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.device_lr = bool(device_lr)
        self._plans = {}   # group index -> the per-step pointer tables of step() (see there)

    def set_lr(self, lr, group=None):
        """New learning rate for one group (or all): the host value and, with device_lr, the device scalar the kernel
        reads -- an asynchronous 4-byte fill on the current stream, legal between replays of a captured step."""
        for i, g in enumerate(self.param_groups):
            if group is None or group == i:
                g['lr'] = float(lr)
                if self.device_lr and 'lr_dev' in g:
                    g['lr_dev'].fill_(float(lr))

    def _init_group(self, group):
        for p in group['params']:
            st = self.state[p]
            if not st:
                st['exp_avg'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        if 'step_dev' not in group:
            dev = group['params'][0].device
            group['step_dev'] = torch.zeros(2, dtype=torch.float32, device=dev)  # {count, ticket of the kernel}
        if self.device_lr and 'lr_dev' not in group:
            group['lr_dev'] = torch.full((1,), float(group['lr']), dtype=torch.float32, device=group['params'][0].device)

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
        for gi, group in enumerate(self.param_groups):
            if not group['params']:
                continue
            ps = [p for p in group['params'] if p.grad is not None]
            # Host side of a step with hundreds of tensors (configs[2]: 269; at 4 tracklets the step is bound by the host): what
            # does not change from step to step -- the parameters' and moments' addresses, sizes, the checks on them, which of
            # them are convolution weights with cached operand layouts -- is kept per group and validated by identity + address;
            # only the gradients' addresses are collected every step (backward allocates them anew).
            plan = self._plans.get(gi)
            if plan is not None and ps and self.state[ps[0]].get('exp_avg') is not plan['m0']:
                plan = None   # (the moments were replaced: a state dict loaded behind our back)
            if plan is None or len(plan['ps']) != len(ps) or any(a is not b for a, b in zip(plan['ps'], ps)) \
                    or plan['ptrs'] != [p.data_ptr() for p in ps]:
                self._init_group(group)
                for p in ps:
                    L.require_device(p, p.grad)
                    if p.dtype != torch.float32:
                        raise L.OcoccError('AdamW kernel takes float32 parameters and gradients')
                    if not p.is_contiguous():
                        raise L.OcoccError('AdamW kernel takes contiguous parameters and gradients')
                chunks = []
                for lo in range(0, len(ps), _MAX):
                    chunk = ps[lo:lo + _MAX]
                    n = len(chunk)
                    arr = ctypes.c_void_p * n
                    chunks.append((chunk, n, arr, arr(*[p.data_ptr() for p in chunk]),
                                   arr(*[self.state[p]['exp_avg'].data_ptr() for p in chunk]),
                                   arr(*[self.state[p]['exp_avg_sq'].data_ptr() for p in chunk]),
                                   (ctypes.c_int64 * n)(*[p.numel() for p in chunk]),
                                   any(p.dim() >= 3 for p in chunk)))   # (only convolution weights have operand layouts)
                plan = self._plans[gi] = dict(ps=list(ps), ptrs=[p.data_ptr() for p in ps], chunks=chunks,
                                             m0=self.state[ps[0]]['exp_avg'] if ps else None)
            for p in ps:
                g = p.grad
                if g.dtype != torch.float32 or not g.is_cuda:
                    raise L.OcoccError('AdamW kernel takes float32 parameters and gradients')
                if not g.is_contiguous():
                    raise L.OcoccError('AdamW kernel takes contiguous parameters and gradients')
            b1, b2 = group['betas']
            refreshed = []
            # every launch of one step must see the same step count: only the last one bumps it
            for ci, (chunk, n, arr, p_arr, m_arr, v_arr, n_arr, has_conv) in enumerate(plan['chunks']):
                last = ci + 1 == len(plan['chunks'])
                g_arr = arr(*[p.grad.data_ptr() for p in chunk])
                # convolution weights whose bf16 kernel operands are cached for the current step: the update rewrites the
                # operands too (ococc_adamw_operands_f32) and the preparation launch of the next step disappears
                targets = []
                if REFRESH_CONV_OPERANDS and has_conv:
                    from .spconv import ops as sp_ops
                    for i, p in enumerate(chunk):
                        for mode, kvol, cin, cout, wn in sp_ops.refresh_targets(p):
                            targets.append((i, mode, kvol, cin, cout, wn, p))
                if targets and len(targets) <= 8:
                    no = len(targets)
                    i32 = ctypes.c_int32 * no
                    L.check(L.lib.ococc_adamw_operands_f32(
                        n, p_arr, g_arr, m_arr, v_arr, n_arr, float(group['lr']),
                        group['lr_dev'].data_ptr() if self.device_lr else None, float(b1), float(b2), float(group['eps']),
                        float(group['weight_decay']), group['step_dev'].data_ptr(), _BUMP if last else 0, no,
                        i32(*[t[0] for t in targets]), i32(*[t[1] for t in targets]), i32(*[t[2] for t in targets]),
                        i32(*[t[3] for t in targets]), i32(*[t[4] for t in targets]),
                        (ctypes.c_void_p * no)(*[t[5].data_ptr() for t in targets]), L.stream()), 'adamw_operands')
                    refreshed += [t[6] for t in targets]
                    continue
                fn, lr_arg = ((L.lib.ococc_adamw_lr_dev_f32, group['lr_dev'].data_ptr()) if self.device_lr
                              else (L.lib.ococc_adamw_f32, float(group['lr'])))
                L.check(fn(n, p_arr, g_arr, m_arr, v_arr, n_arr, lr_arg, float(b1), float(b2), float(group['eps']),
                           float(group['weight_decay']), group['step_dev'].data_ptr(), _BUMP if last else 0, L.stream()), 'adamw')
            for p in ps:  # the kernel wrote through raw pointers: tell autograd / version-keyed caches
                torch.autograd.graph.increment_version(p)
            if refreshed:
                from .spconv import ops as sp_ops
                for p in refreshed:
                    sp_ops.operands_refreshed(p)
        return loss


    def state_dict(self):
        sd = super().state_dict()
        for g_out, g in zip(sd['param_groups'], self.param_groups):  # device scalars -> plain numbers
            g_out.pop('lr_dev', None)
            g_out['step_dev'] = float(g['step_dev'][0]) if 'step_dev' in g else 0.0
        return sd

    def load_state_dict(self, state_dict):
        counts = [g.get('step_dev', 0.0) for g in state_dict['param_groups']]
        for g in state_dict['param_groups']:
            g.pop('step_dev', None)
        super().load_state_dict(state_dict)
        self._plans.clear()   # (new moment tensors: the cached pointer tables of step() are stale)
        self.init_state()
        for g, c in zip(self.param_groups, counts):
            g['step_dev'][0] = float(c)
            if self.device_lr:
                g['lr_dev'].fill_(float(g['lr']))


def param_groups_from_cfg(named_params, weight_decay, paramwise_cfg=None):
    """mmcv DefaultOptimizerConstructor with ``paramwise_cfg=dict(custom_keys={key: dict(lr_mult=, decay_mult=)})``:
    a parameter whose NAME contains a key (the longest matching key wins) gets that key's multipliers; with the
    reference's ``{'norm': dict(decay_mult=0.)}`` the LayerNorms called ``...norm...`` are exempt from weight decay
    (the ones build_mlp creates are called ``<mlp>.<i>.1`` and are not -- reproduced as is).
    -> list of groups ``dict(params=[...], weight_decay=..., lr_mult=...)``."""
    keys = sorted((paramwise_cfg or {}).get('custom_keys', {}).items(), key=lambda kv: -len(kv[0]))
    groups = {}
    for name, p in named_params:
        if not p.requires_grad:
            continue
        lr_mult, decay_mult = 1.0, 1.0
        for key, mult in keys:
            if key in name:
                lr_mult, decay_mult = float(mult.get('lr_mult', 1.0)), float(mult.get('decay_mult', 1.0))
                break
        groups.setdefault((lr_mult, decay_mult), []).append(p)
    return [dict(params=ps, weight_decay=weight_decay * dm, lr_mult=lm) for (lm, dm), ps in groups.items()]


def cyclic_lr(base_lr, it, max_iters, target_ratio=(100, 1e-3), cyclic_times=1, step_ratio_up=0.1):
    """mmcv CyclicLrUpdaterHook (by_epoch=False, anneal_strategy='cos', gamma=1), the ``lr_config`` of
    cosine_2x.py:10-15: within each of ``cyclic_times`` cycles the rate rises from base_lr to
    base_lr * target_ratio[0] over the first step_ratio_up of the cycle and falls to base_lr * target_ratio[1]
    over the rest, both along half a cosine."""
    import math
    cycle = max_iters // cyclic_times
    up = int(step_ratio_up * cycle)
    t = it % cycle
    if t < up:
        start, end, pct = 1.0, float(target_ratio[0]), t / max(up, 1)
    else:
        start, end, pct = float(target_ratio[0]), float(target_ratio[1]), (t - up) / max(cycle - up, 1)
    ratio = end + 0.5 * (start - end) * (math.cos(math.pi * pct) + 1.0)
    return base_lr * ratio
