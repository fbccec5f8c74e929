"""Yogi optimiser on the fused HIP step kernel (reference call site: experiments/vargp.py:23,37 uses
torch_optimizer.Yogi).

torch_optimizer is not available in this environment and is unpinned in the reference's
environment.yml, so the arithmetic follows the published algorithm (Zaheer et al., NeurIPS 2018) with
the defaults torch_optimizer documents: betas (0.9, 0.999), eps 1e-3, initial_accumulator 1e-6 for
both moment buffers.  PARITY UNPINNED for optimiser trajectories.
"""
import ctypes

import torch

from ._lib import check, lib, ptr, require_device, stream_ptr


class Yogi(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-2, betas=(0.9, 0.999), eps=1e-3, initial_accumulator=1e-6):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, initial_accumulator=initial_accumulator))
        # True: somebody else advances the device-side step count before step() (ElboTrainer lets the ELBO
        # program's first kernel do it), so step() launches nothing but the update kernel
        self.external_step = False
        self._chunk_cache = {}

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._chunk_cache = {}                 # the moment buffers are new tensors

    def __getstate__(self):
        # the cache holds ctypes pointer arrays (not picklable, not deep-copyable) of addresses that mean nothing elsewhere
        state = super().__getstate__() if hasattr(super(), '__getstate__') else dict(self.__dict__)
        state = dict(state)
        state.pop('_chunk_cache', None)
        return state

    def __setstate__(self, state):
        super().__setstate__(state)
        self._chunk_cache = {}
        self.__dict__.setdefault('external_step', False)

    def step_counter(self, device):
        """The device-side step count of the (single) parameter group."""
        group = self.param_groups[0]
        if 'step' not in group:
            group['step'] = torch.zeros(1, dtype=torch.float32, device=device)
        return group['step']

    @torch.no_grad()
    def step(self, hyper=None):
        """One fused launch for all parameter tensors of a group (<= 8).  The step count lives on the
        device (bias corrections are computed inside the kernel), so the whole step can sit inside a
        captured hipGraph.
        hyper = (HyperGradDesc, log_mean, log_logvar): the gradients of those two parameters are finished inside this
        launch (an ELBO program's backward ran with defer_hyper=True) and stored into their .grad as well."""
        for group in self.param_groups:
            b1, b2 = group['betas']
            ps = [p for p in group['params'] if p.grad is not None]
            if not ps:
                continue
            if 'step' not in group:
                group['step'] = torch.zeros(1, dtype=torch.float32, device=ps[0].device)
            if not self.external_step:
                group['step'].add_(1.0)
            for p in ps:
                st = self.state[p]
                if not st:
                    require_device(p, p.grad)
                    st['exp_avg'] = torch.full_like(p, group['initial_accumulator'])
                    st['exp_avg_sq'] = torch.full_like(p, group['initial_accumulator'])
            for i in range(0, len(ps), 8):
                chunk = ps[i:i + 8]
                k = len(chunk)
                grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in chunk]
                if not all(g.is_cuda for g in grads):
                    require_device(*grads)
                arr = lambda ts: (ctypes.c_void_p * k)(*[t.data_ptr() for t in ts])
                # parameters and moment buffers keep their addresses from step to step: their pointer arrays are built once
                # per chunk (the eager drop-in loop is host-bound; this call used to cost ~60 us of Python)
                key = tuple(id(p) for p in chunk)
                cached = self._chunk_cache.get(key)
                # valid while the parameters AND both moment buffers sit where they sat when the arrays were built (a moment
                # tensor replaced through optim.state, state.clear(), a move to another device: rebuilt)
                where = tuple(t.data_ptr() for p in chunk for t in (p, self.state[p]['exp_avg'], self.state[p]['exp_avg_sq']))
                if cached is None or cached[0] != where:
                    cached = (where, arr(chunk), arr([self.state[p]['exp_avg'] for p in chunk]),
                              arr([self.state[p]['exp_avg_sq'] for p in chunk]),
                              (ctypes.c_int64 * k)(*[p.numel() for p in chunk]))
                    self._chunk_cache[key] = cached
                _, a_p, a_m, a_v, sizes = cached
                ids = list(key)
                if hyper is not None and id(hyper[1]) in ids:
                    h, p_mean, p_logvar = hyper
                    i_lv = ids.index(id(p_logvar)) if (p_logvar is not None and id(p_logvar) in ids) else -1
                    check(lib().vargp_yogi_step_multi_hyper(
                        k, a_p, arr(grads), a_m, a_v, sizes, group['lr'], b1, b2, group['eps'],
                        ptr(group['step']), 0, ctypes.byref(h), ids.index(id(p_mean)), i_lv, stream_ptr()),
                        'vargp_yogi_step_multi_hyper')
                    continue
                check(lib().vargp_yogi_step_multi(k, a_p, arr(grads), a_m, a_v, sizes,
                                                  group['lr'], b1, b2, group['eps'], ptr(group['step']), 0,
                                                  stream_ptr()),
                      'vargp_yogi_step_multi')
