"""The ELBO through the native programs: `vargp_elbo_t0_fwd / _bwd` (csrc/elbo_t0.hip, first task) and
`vargp_elbo_tn_fwd / _bwd` (csrc/elbo_tn.hip, models with previous tasks: `TnProgram`, `elbo_tn`, at the end of this file).

First task:

`VARGP.loss` of a model without previous tasks (reference: var_gp/vargp.py:156-194) is, on this path, two C-ABI
calls: the forward sequences ~10 kernels (hyper-parameter sampling + KL, both kernel matrices in one GEMM launch,
one batched Cholesky/inverse, one GEMM for everything multiplied by Lz^-1, predictive moments, KL, softmax
likelihood), the backward ~20.  This module holds
  * `T0Program` — descriptor + workspace for one problem shape; `forward()` / `backward()` are the two calls;
  * `elbo_t0()` — the same program as ONE autograd node, which is what `VARGP.loss` returns into the reference's
    training loop (`loss.backward()`, experiments/vargp.py:34-35);
`train.ElboTrainer` drives a persistent `T0Program` directly (no autograd graph, gradients written straight
into the optimiser's buffers).
"""
import ctypes
import weakref

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import ops
from ._lib import ElboT0Desc, ElboTnDesc, HyperGradDesc, check, lib, ptr, require_device, stream_ptr, workspace
from .ops import JITTER


def _p(t):
    return t.data_ptr() if t is not None else None


class T0Program:
    """Descriptor + workspace of the native first-task ELBO for fixed (S, C, M, D, B, F).  The workspace carries
    every intermediate from `forward` to `backward`; one `backward` per `forward`."""

    def __init__(self, S, C, M, D, B, F, device, map_est=False):
        self.shape = (S, C, M, D, B, F)
        self.map_est = bool(map_est)
        nbytes = lib().vargp_elbo_t0_workspace_bytes(S, C, M, D, B, F)
        self.ws = workspace(nbytes, device)
        self.scalars = torch.empty(3, dtype=torch.float32, device=device)
        self.info = torch.empty(S * C + C, dtype=torch.int32, device=device)
        self.desc = ElboT0Desc(S=S, C=C, M=M, D=D, B=B, F=F, map_est=int(self.map_est), jitter=JITTER,
                               scalars=_p(self.scalars), info=_p(self.info), ws=_p(self.ws),
                               ws_bytes=self.ws.numel() * 4)
        self._keep = None
        self._rng = None
        # busy / _gen: as TnProgram -- a forward recorded by autograd owns the workspace until its backward has run or the
        # graph node has died (VARGP._t0_program then hands out a spare of the same shape)
        self.busy = False
        self._gen = 0

    def set_rng(self, seed, counter, sample_offset=0):
        """Native noise: `forward(eps_theta=None, eps_f=None)` then draws both noise tensors inside the program from a
        Philox4x32-10 generator keyed by `seed`; `counter` (device int32/uint32 tensor of one element) is the step
        number, advanced by every forward; `sample_offset` = index of this rank's first hyper-sample in the global
        draw (sample-parallel ranks: rank * S)."""
        assert counter.is_cuda and counter.numel() == 1 and counter.element_size() == 4
        self._rng = (int(seed), counter, int(sample_offset))
        self.desc.rng_seed, self.desc.rng_counter, self.desc.rng_sample_offset = int(seed), _p(counter), int(sample_offset)

    def _view(self, index, shape):
        # workspace layout (csrc/elbo_t0.hip, carve_t0): theta | eps_theta | eps_f | ..., each rounded up to 64 floats
        S, C, M, D, B, F_ = self.shape
        sizes = [S * (D + 1), S * (D + 1), S * F_ * C * B]
        off = sum((n + 63) // 64 * 64 for n in sizes[:index])
        n = sizes[index]
        return self.ws[off:off + n].view(shape)

    def eps_theta(self):
        """Hyper-parameter noise drawn by the last native-noise forward, (S, D+1)."""
        return self._view(1, (self.shape[0], self.shape[3] + 1))

    def eps_f(self):
        """Likelihood noise drawn by the last native-noise forward, (S, F, C, B)."""
        S, C, M, D, B, F_ = self.shape
        return self._view(2, (S, F_, C, B))

    @staticmethod
    def shape_of(n_v, z, x, n_f):
        return (n_v, z.shape[0], z.shape[1], z.shape[2], x.shape[0], n_f)

    def lik_buffers(self):
        """(mu, var, gmu, gvar), each (S, C, B): the predictive moments of the last forward and the likelihood-gradient buffers
        the backward reads (views into the workspace).  With `forward(ext_lik=True)` the caller fills gmu / gvar (seeded)."""
        S, C, M, D, B, F_ = self.shape[:6]
        ps = [ctypes.c_void_p() for _ in range(4)]
        fn = lib().vargp_elbo_tn_lik_buffers if isinstance(self, TnProgram) else lib().vargp_elbo_t0_lik_buffers
        check(fn(ctypes.byref(self.desc), *(ctypes.byref(q) for q in ps)), 'lik_buffers')
        base = self.ws.data_ptr()
        return tuple(self.ws[(q.value - base) // 4:(q.value - base) // 4 + S * C * B].view(S, C, B) for q in ps)

    def forward(self, log_mean, log_logvar, prior_log_mean, prior_log_logvar, z, u_mean, u_tril_vec, x, y, eps_theta,
                eps_f, bump=None, defer_softmax=False, ext_lik=False):
        """-> scalars (3,) = (kl_hypers, kl_u, nll).  All tensors contiguous fp32 on the ROCm device (y int64).
        ext_lik: the likelihood is the caller's (class-sharded ranks, include/vargp_hip.h): moments + KL only, nll stays 0;
        y / eps_f may be None; fill lik_buffers()[2:] before `backward`.
        defer_softmax: the caller runs `backward` right behind this forward and reads nll only afterwards (ElboTrainer): the
        likelihood is then evaluated inside the backward's tile kernel where the shapes allow (include/vargp_hip.h).
        eps_theta = eps_f = None: the program draws the noise itself (see set_rng).  `bump`: optional device float that the forward increments by one (an optimiser's step counter)."""
        tensors = (log_mean, log_logvar, prior_log_mean, prior_log_logvar, z, u_mean, u_tril_vec, x, y, eps_theta, eps_f)
        require_device(*tensors)
        for t in tensors:
            if t is not None and not t.is_contiguous():
                raise ValueError('T0Program.forward needs contiguous tensors')
        S, C, M, D, B, F_ = self.shape
        assert z.shape == (C, M, D) and x.shape == (B, D) and (ext_lik or y.dtype == torch.int64)
        assert u_mean.numel() == C * M and u_tril_vec.shape == (C, M * (M + 1) // 2) and log_mean.numel() == D + 1
        if ext_lik:
            assert self.map_est or eps_theta.shape == (S, D + 1)
        elif eps_f is None:
            assert self._rng is not None and eps_theta is None, 'native noise: call set_rng() and pass no eps tensors'
        else:
            assert eps_f.shape == (S, F_, C, B) and (self.map_est or eps_theta.shape == (S, D + 1))
        d = self.desc
        d.log_mean, d.log_logvar = _p(log_mean), _p(log_logvar)
        d.prior_log_mean, d.prior_log_logvar = _p(prior_log_mean), _p(prior_log_logvar)
        d.z, d.u_mean, d.u_tril_vec, d.x, d.y = _p(z), _p(u_mean), _p(u_tril_vec), _p(x), _p(y)
        d.eps_theta, d.eps_f = _p(eps_theta), _p(eps_f)
        d.bump = _p(bump)
        d.scalars = _p(self.scalars)                  # (VARGP.loss hands out a fresh slot of a small ring per forward: fused.elbo_lazy)
        d.defer_softmax = int(bool(defer_softmax))
        d.ext_lik = int(bool(ext_lik))
        self._keep = tensors + (bump,)   # the descriptor holds raw pointers: keep the tensors alive until backward
        self._ver = tuple(t._version for t in (log_mean, log_logvar, z, u_mean, u_tril_vec))
        # 'raise' mode: the status words are copied out and an event recorded right behind the factorisation launch (the second of
        # four): the host waits for that, and the rest of the forward runs while it carries on (combine, the backward's launch)
        # ('lazy': the same copy + event, looked at by a later call -- no torch launch for the status words at all)
        early = ops._chol_mode in ('raise', 'lazy') and not torch.cuda.is_current_stream_capturing()
        if early:
            host, ev = ops.raise_slot(self.info.numel()) if ops._chol_mode == 'raise' else ops.lazy_slot(self.info.numel())
            d.info_host, d.info_event = host.data_ptr(), ev.cuda_event
        else:
            d.info_host, d.info_event = None, None
        check(lib().vargp_elbo_t0_fwd(ctypes.byref(d), stream_ptr()), 'vargp_elbo_t0_fwd')
        self._bwd_ok = True
        if not early:
            ops._note_chol_errors(self.info)
        elif ops._chol_mode == 'raise':
            ops.raise_wait(host, ev)
        else:
            ops._pending.append((host, ev))
        return self.scalars

    def backward(self, seeds, g_log_mean, g_log_logvar, g_z, g_u_mean, g_u_tril_vec, defer_hyper=False):
        """seeds (3,) device = d total / d (kl_hypers, kl_u, nll); overwrites the five gradient buffers.
        defer_hyper: the last kernel (theta-gradient -> log_mean / log_logvar) is left to the optimiser's launch
        (`hyper_desc()` -> optim.Yogi.step(hyper=...)); g_log_mean / g_log_logvar are then written by that launch."""
        require_device(seeds, g_log_mean, g_log_logvar, g_z, g_u_mean, g_u_tril_vec)
        # (the forward clears the accumulators the backward adds into: include/vargp_hip.h, vargp_elbo_t0_bwd)
        if self._keep is None or not getattr(self, '_bwd_ok', False):
            raise RuntimeError('T0Program.backward: one backward per forward (the forward clears the accumulators the backward '
                               'adds into); call rerun_forward() first to evaluate the same forward again for another backward')
        self._bwd_ok = False
        for g in (g_log_mean, g_log_logvar, g_z, g_u_mean, g_u_tril_vec):
            assert g.is_contiguous() and g.dtype == torch.float32
        self.desc.defer_hyper = int(bool(defer_hyper))
        self._seeds = seeds
        check(lib().vargp_elbo_t0_bwd(ctypes.byref(self.desc), ptr(seeds), ptr(g_log_mean), ptr(g_log_logvar), ptr(g_z),
                                      ptr(g_u_mean), ptr(g_u_tril_vec), stream_ptr()), 'vargp_elbo_t0_bwd')

    def rerun_forward(self):
        """The last forward again -- same operands, same noise, no counter advanced, status words not re-reported -- so that a
        SECOND backward can run on it (`loss.backward(retain_graph=True)` followed by another backward is legal in the
        reference's loop, experiments/vargp.py:35: ordinary autograd).  The program's backward consumes the accumulators its
        forward cleared, so the retained "graph" is re-evaluated rather than kept: results equal the first evaluation's up to the
        order of the float atomics."""
        if self._keep is None:
            raise RuntimeError('rerun_forward without a forward')
        (log_mean, log_logvar, _, _, z, u_mean, u_tril_vec) = self._keep[:7]
        if tuple(t._version for t in (log_mean, log_logvar, z, u_mean, u_tril_vec)) != self._ver:
            raise RuntimeError('one of the variables needed for gradient computation has been modified by an inplace operation '
                               '(a parameter changed between VARGP.loss and this second backward of its retained graph)')
        d = self.desc
        if not d.eps_f and not d.ext_lik:                 # the program drew the noise itself: it is still in the workspace
            et = None if self.map_est else self.eps_theta().clone()
            ef = self.eps_f().clone()
            d.eps_theta, d.eps_f = _p(et), _p(ef)
            self._keep = self._keep + (et, ef)
        if getattr(self, '_rerun_scal', None) is None:
            self._rerun_scal = torch.empty(3, dtype=torch.float32, device=self.ws.device)
        d.bump, d.info_host, d.info_event = None, None, None
        d.scalars = _p(self._rerun_scal)                  # (the first evaluation's numbers stay where the caller reads them)
        fn = lib().vargp_elbo_tn_fwd if isinstance(self, TnProgram) else lib().vargp_elbo_t0_fwd
        check(fn(ctypes.byref(d), stream_ptr()), 'rerun_forward')
        self._bwd_ok = True

    def hyper_desc(self):
        """What the deferred last step of `backward(defer_hyper=True)` needs (pointers into this program's workspace)."""
        h = HyperGradDesc()
        check(lib().vargp_elbo_t0_hyper_desc(ctypes.byref(self.desc), ptr(self._seeds), ctypes.byref(h)), 'vargp_elbo_t0_hyper_desc')
        return h

    def theta(self):
        """The hyper-parameter samples of the last forward, (S, D+1) (view into the workspace)."""
        return self._view(0, (self.shape[0], self.shape[3] + 1))


def _release(prog, gen):
    if prog._gen == gen:
        prog.busy = False


_REUSED = ('VARGP.loss: the workspace of this ELBO node has been handed to a later loss() -- its forward cannot be re-evaluated '
           'for another backward.  Keep the graph with loss.backward(retain_graph=True) (the workspace then stays with this '
           'loss until it is dropped), or call loss() again')


class _ElboT0(Function):
    @staticmethod
    def forward(ctx, log_mean, log_logvar, z, u_mean, u_tril_vec, x, y, eps_theta, eps_f, prior_log_mean,
                prior_log_logvar, map_est, prog):
        args = [t.contiguous() if t is not None else None
                for t in (log_mean, log_logvar, prior_log_mean, prior_log_logvar, z, u_mean, u_tril_vec, x, y, eps_theta,
                          eps_f)]
        if prog is None:         # no cached program handed in: a workspace of this node's own
            S = 1 if map_est else eps_theta.shape[0]
            prog = T0Program(*T0Program.shape_of(S, z, x, eps_f.shape[1]), z.device, map_est)
        scal = prog.forward(*args).clone()          # the program's scalars are overwritten by its next forward
        # the workspace belongs to this node until its backward has run or the node has died (see _ElboTn.forward)
        prog._gen += 1
        prog.busy = True
        weakref.finalize(ctx, _release, prog, prog._gen)
        ctx.gen = prog._gen
        ctx.prog = prog
        ctx.map_est = map_est
        ctx.shapes = (log_mean.shape, z.shape, u_mean.shape, u_tril_vec.shape)
        return scal[0], scal[1], scal[2]

    @staticmethod
    @once_differentiable
    def backward(ctx, g_klh, g_klu, g_nll):
        prog = ctx.prog
        if prog._gen != ctx.gen:
            raise RuntimeError(_REUSED)
        if getattr(ctx, 'ran', False):
            prog.rerun_forward()             # second backward of a retained graph: the forward is evaluated again
        ctx.ran = True
        seeds = torch.stack([g_klh.reshape(()), g_klu.reshape(()), g_nll.reshape(())]).float()
        sh_mean, sh_z, sh_um, sh_uv = ctx.shapes
        dev = seeds.device
        # the five gradients as views of ONE allocation (each starting on a 256-byte boundary): one allocator call instead of five
        ns = [sh_mean.numel(), sh_mean.numel(), sh_z.numel(), sh_um.numel(), sh_uv.numel()]
        offs, tot = [], 0
        for n in ns:
            offs.append(tot)
            tot += (n + 63) // 64 * 64
        flat = torch.empty(tot, dtype=torch.float32, device=dev)
        g_mean, g_logvar, g_z, g_um, g_uv = (flat[o:o + n].view(sh) for o, n, sh in zip(offs, ns, (sh_mean, sh_mean, sh_z, sh_um, sh_uv)))
        prog.backward(seeds, g_mean, g_logvar, g_z, g_um, g_uv)
        _release(prog, ctx.gen)              # (a later loss() may take the workspace: a further backward of THIS node then raises)
        return (g_mean, None if ctx.map_est else g_logvar, g_z, g_um, g_uv, None, None, None, None, None, None, None, None)


def elbo_t0(kernel, z, u_mean, u_tril_vec, x, y, eps_theta, eps_f, prog=None):
    """-> (kl_hypers, kl_u, nll) of VARGP.loss for a model with no previous tasks; `kernel` is the RBFKernel
    module (variational hyper-parameters and their prior).  `prog`: the (cached, not busy) T0Program of this shape."""
    return _ElboT0.apply(kernel.log_mean, kernel.log_logvar, z, u_mean, u_tril_vec, x, y, eps_theta, eps_f,
                         kernel.prior_log_mean, kernel.prior_log_logvar, bool(kernel.map_est), prog)


# ----------------------------------------------------------------------------------------------------------------
# models with previous tasks (and any forward-only evaluation): the block-structured program, csrc/elbo_tn.hip
# ----------------------------------------------------------------------------------------------------------------
def tn_row_width(M):
    """Row width NR of the packed operand rk_all: [u_mean | 0 0 0 | Lu (M columns)] rounded up to a multiple of 4."""
    return (4 + M + 3) // 4 * 4


def pack_tn_operands(prev, C, M, D, device):
    """Caller-maintained operands of the program for a model whose earlier tasks are `prev` (list of dicts with z
    (C,M,D), u_mean (C,M,1), u_tril (C,M,M), all M equal): z_all (C, Mt, D) and rk_all (C, nblk, M, NR) with the earlier
    tasks filled in and the last block left for the program (it writes the current task there on every forward)."""
    nblk = len(prev) + 1
    NR = tn_row_width(M)
    z_all = torch.zeros(C, nblk * M, D, dtype=torch.float32, device=device)
    rk_all = torch.zeros(C, nblk, M, NR, dtype=torch.float32, device=device)
    for i, p in enumerate(prev):
        z_all[:, i * M:(i + 1) * M] = p['z']
        rk_all[:, i, :, 0] = p['u_mean'].reshape(C, M)
        rk_all[:, i, :, 4:4 + M] = p['u_tril']
    return z_all, rk_all


class TnProgram:
    """Descriptor + workspace of `vargp_elbo_tn_*` for fixed (S, C, M, D, B, F, nblk).  One `backward` per `forward`;
    `forward(y=None)` evaluates the predictive moments only."""

    def __init__(self, S, C, M, D, B, F, nblk, device, map_est=False, forward_only=False):
        self.shape = (S, C, M, D, B, F, nblk)
        self.map_est = bool(map_est)
        # forward_only: predictive moments only (VARGP.forward / predict): none of the gradient buffers is carved
        self.forward_only = bool(forward_only)
        nbytes = (lib().vargp_elbo_tn_workspace_bytes_fwd if self.forward_only else lib().vargp_elbo_tn_workspace_bytes)(
            S, C, M, D, B, F, nblk)
        self.ws = workspace(nbytes, device)
        self.scalars = torch.empty(3, dtype=torch.float32, device=device)
        self.info = torch.empty(S * C, dtype=torch.int32, device=device)
        self.desc = ElboTnDesc(S=S, C=C, M=M, D=D, B=B, F=F, nblk=nblk, map_est=int(self.map_est), jitter=JITTER,
                               scalars=_p(self.scalars), info=_p(self.info), ws=_p(self.ws),
                               ws_bytes=self.ws.numel() * 4, forward_only=int(self.forward_only))
        self._keep = None
        self._rng = None
        # busy: a forward whose backward can still come owns the workspace (it carries the intermediates).  Set by the
        # autograd node (_ElboTn) only when a graph is being recorded, cleared by its backward or when the node dies.
        self.busy = False
        self._gen = 0

    def set_rng(self, seed, counter, sample_offset=0):
        """Native noise, as T0Program.set_rng."""
        assert counter.is_cuda and counter.numel() == 1 and counter.element_size() == 4
        self._rng = (int(seed), counter, int(sample_offset))
        self.desc.rng_seed, self.desc.rng_counter, self.desc.rng_sample_offset = int(seed), _p(counter), int(sample_offset)

    def _view(self, index, shape):
        S, C, M, D, B, F_, nblk = self.shape
        sizes = [S * (D + 1), S * (D + 1), S * F_ * C * B]
        off = sum((n + 63) // 64 * 64 for n in sizes[:index])
        return self.ws[off:off + sizes[index]].view(shape)

    def theta(self):
        return self._view(0, (self.shape[0], self.shape[3] + 1))

    def eps_theta(self):
        return self._view(1, (self.shape[0], self.shape[3] + 1))

    def eps_f(self):
        S, C, M, D, B, F_, nblk = self.shape
        return self._view(2, (S, F_, C, B))

    def moments(self, Bt=None):
        """(mu, var) (S, C, B) of the last forward (or (S, C, Bt) of the last moments-only tile): views into the workspace."""
        S, C, M, D, B, F_, nblk = self.shape
        B = B if Bt is None else int(Bt)
        pm, pv = ctypes.c_void_p(), ctypes.c_void_p()
        check(lib().vargp_elbo_tn_moments(ctypes.byref(self.desc), ctypes.byref(pm), ctypes.byref(pv)), 'vargp_elbo_tn_moments')
        base = self.ws.data_ptr()
        om, ov = (pm.value - base) // 4, (pv.value - base) // 4
        return self.ws[om:om + S * C * B].view(S, C, B), self.ws[ov:ov + S * C * B].view(S, C, B)

    # -- predictive sweep: the x-independent part once (sweep_begin), then moments per tile of <= B points --------------------
    def sweep_begin(self, log_mean, log_logvar, prior_log_mean, prior_log_logvar, z, u_mean, u_tril_vec, z_all, rk_all, eps_theta):
        """theta (from eps_theta (S, D+1); None under map_est), K(z_<=t), L, T and the small products: everything of the
        predictive moments that does not depend on x (vargp_elbo_tn_begin)."""
        tensors = (log_mean, log_logvar, prior_log_mean, prior_log_logvar, z, u_mean, u_tril_vec, z_all, rk_all, eps_theta)
        require_device(*tensors)
        for t in tensors:
            if t is not None and not t.is_contiguous():
                raise ValueError('TnProgram.sweep_begin needs contiguous tensors')
        S, C, M, D, B, F_, nblk = self.shape
        assert self.map_est or (eps_theta is not None and eps_theta.shape == (S, D + 1))
        d = self.desc
        d.log_mean, d.log_logvar = _p(log_mean), _p(log_logvar)
        d.prior_log_mean, d.prior_log_logvar = _p(prior_log_mean), _p(prior_log_logvar)
        d.z, d.u_mean, d.u_tril_vec = _p(z), _p(u_mean), _p(u_tril_vec)
        d.x, d.y = _p(z), None              # x is not read by begin (any non-null device pointer)
        d.z_all, d.rk_all = _p(z_all), _p(rk_all)
        d.eps_theta, d.eps_f = _p(eps_theta), None
        d.bump = None
        self._keep = tensors
        check(lib().vargp_elbo_tn_begin(ctypes.byref(d), stream_ptr()), 'vargp_elbo_tn_begin')
        ops._note_chol_errors(self.info)

    def sweep_moments(self, x):
        """x (Bt <= B, D) -> (mu, var) (S, C, Bt): views into the workspace, valid until the next tile."""
        S, C, M, D, B, F_, nblk = self.shape
        require_device(x)
        assert x.dim() == 2 and x.shape[1] == D and x.shape[0] <= B and x.is_contiguous()
        check(lib().vargp_elbo_tn_tile(ctypes.byref(self.desc), None, ptr(x), None, None, x.shape[0], stream_ptr()),
              'vargp_elbo_tn_tile')
        return self.moments(x.shape[0])

    lik_buffers = T0Program.lik_buffers
    rerun_forward = T0Program.rerun_forward

    def forward(self, log_mean, log_logvar, prior_log_mean, prior_log_logvar, z, u_mean, u_tril_vec, z_all, rk_all, x, y,
                eps_theta, eps_f, bump=None, ext_lik=False, eps_u=None):
        """-> scalars (3,) = (kl_hypers, kl_u, nll) (y given) or None (y None: moments only).  ext_lik: as T0Program.forward
        (y must still be given: it switches the KL on).  eps_u (n_v, S, C, (nblk - 1) M): ep_var_mean = False -- the KL keeps the
        conditional prior's mean at these n_v samples of u_<t (include/vargp_hip.h: no_var_mean)."""
        tensors = (log_mean, log_logvar, prior_log_mean, prior_log_logvar, z, u_mean, u_tril_vec, z_all, rk_all, x, y,
                   eps_theta, eps_f)
        require_device(*tensors)
        for t in tensors:
            if t is not None and not t.is_contiguous():
                raise ValueError('TnProgram.forward needs contiguous tensors')
        S, C, M, D, B, F_, nblk = self.shape
        assert z.shape == (C, M, D) and x.shape == (B, D) and (y is None or y.dtype == torch.int64)
        assert z_all.shape == (C, nblk * M, D) and rk_all.shape == (C, nblk, M, tn_row_width(M))
        assert u_mean.numel() == C * M and u_tril_vec.shape == (C, M * (M + 1) // 2) and log_mean.numel() == D + 1
        if ext_lik:
            assert y is not None and (self.map_est or eps_theta.shape == (S, D + 1))
        elif y is not None and eps_f is None:
            assert self._rng is not None and eps_theta is None, 'native noise: call set_rng() and pass no eps tensors'
        elif y is not None:
            assert eps_f.shape == (S, F_, C, B) and (self.map_est or eps_theta.shape == (S, D + 1))
        d = self.desc
        d.log_mean, d.log_logvar = _p(log_mean), _p(log_logvar)
        d.prior_log_mean, d.prior_log_logvar = _p(prior_log_mean), _p(prior_log_logvar)
        d.z, d.u_mean, d.u_tril_vec, d.x, d.y = _p(z), _p(u_mean), _p(u_tril_vec), _p(x), _p(y)
        d.z_all, d.rk_all = _p(z_all), _p(rk_all)
        d.eps_theta, d.eps_f = _p(eps_theta), _p(eps_f)
        d.bump = _p(bump)
        d.scalars = _p(self.scalars)
        d.ext_lik = int(bool(ext_lik))
        if eps_u is not None:
            require_device(eps_u)
            assert nblk > 1 and eps_u.is_contiguous() and eps_u.dim() == 4 and eps_u.shape[1:] == (S, C, (nblk - 1) * M), eps_u.shape
            d.eps_u, d.n_v, d.no_var_mean = _p(eps_u), int(eps_u.shape[0]), 1
        else:
            d.eps_u, d.n_v, d.no_var_mean = None, 0, 0
        self._keep = tensors + (bump, eps_u)
        self._ver = tuple(t._version for t in (log_mean, log_logvar, z, u_mean, u_tril_vec))
        early = ops._chol_mode in ('raise', 'lazy') and not torch.cuda.is_current_stream_capturing()      # (as T0Program.forward)
        if early:
            host, ev = ops.raise_slot(self.info.numel()) if ops._chol_mode == 'raise' else ops.lazy_slot(self.info.numel())
            d.info_host, d.info_event = host.data_ptr(), ev.cuda_event
        else:
            d.info_host, d.info_event = None, None
        check(lib().vargp_elbo_tn_fwd(ctypes.byref(d), stream_ptr()), 'vargp_elbo_tn_fwd')
        if not early:
            ops._note_chol_errors(self.info)
        elif ops._chol_mode == 'raise':
            ops.raise_wait(host, ev)
        else:
            ops._pending.append((host, ev))
        return self.scalars if y is not None else None

    def backward(self, seeds, g_log_mean, g_log_logvar, g_z, g_u_mean, g_u_tril_vec, defer_hyper=False):
        """As T0Program.backward."""
        require_device(seeds, g_log_mean, g_log_logvar, g_z, g_u_mean, g_u_tril_vec)
        assert self._keep is not None, 'TnProgram.backward without a forward'
        for g in (g_log_mean, g_log_logvar, g_z, g_u_mean, g_u_tril_vec):
            assert g.is_contiguous() and g.dtype == torch.float32
        self.desc.defer_hyper = int(bool(defer_hyper))
        self._seeds = seeds
        check(lib().vargp_elbo_tn_bwd(ctypes.byref(self.desc), ptr(seeds), ptr(g_log_mean), ptr(g_log_logvar), ptr(g_z),
                                      ptr(g_u_mean), ptr(g_u_tril_vec), stream_ptr()), 'vargp_elbo_tn_bwd')

    def hyper_desc(self):
        h = HyperGradDesc()
        check(lib().vargp_elbo_tn_hyper_desc(ctypes.byref(self.desc), ptr(self._seeds), ctypes.byref(h)), 'vargp_elbo_tn_hyper_desc')
        return h


    # -- N-tiled ELBO: loss and gradient over a data set swept in minibatch tiles (vargp_elbo_tn_begin / _tile / _end) ----
    def tiled_step(self, log_mean, log_logvar, prior_log_mean, prior_log_logvar, z, u_mean, u_tril_vec, z_all, rk_all, x, y,
                   seeds, grads, eps_theta=None, eps_f=None):
        """x (N, D), y (N): swept in tiles of this program's B columns (the last one may be narrower).  seeds (3,) device =
        d total / d (kl_hypers, kl_u, nll); grads = the five gradient buffers (log_mean, log_logvar, z, u_mean, u_tril_vec),
        overwritten.  eps_f (S, F, C, N) / eps_theta (S, D+1): injected noise (tests); None: native noise (set_rng).
        -> scalars (kl_hypers, kl_u, sum over the tiles of nll)."""
        S, C, M, D, B, F_, nblk = self.shape
        tensors = (log_mean, log_logvar, prior_log_mean, prior_log_logvar, z, u_mean, u_tril_vec, z_all, rk_all, x, y, seeds,
                   eps_theta, eps_f) + tuple(grads)
        require_device(*tensors)
        assert x.dim() == 2 and x.shape[1] == D and x.is_contiguous() and y.dtype == torch.int64 and y.is_contiguous()
        if eps_f is None:
            assert self._rng is not None and (eps_theta is None), 'native noise: call set_rng() and pass no eps tensors'
        d = self.desc
        d.log_mean, d.log_logvar = _p(log_mean), _p(log_logvar)
        d.prior_log_mean, d.prior_log_logvar = _p(prior_log_mean), _p(prior_log_logvar)
        d.z, d.u_mean, d.u_tril_vec, d.x, d.y = _p(z), _p(u_mean), _p(u_tril_vec), _p(x), _p(y)
        d.z_all, d.rk_all = _p(z_all), _p(rk_all)
        d.eps_theta, d.eps_f = _p(eps_theta), None
        d.bump = None
        self._keep = tensors
        st = stream_ptr()
        check(lib().vargp_elbo_tn_begin(ctypes.byref(d), st), 'vargp_elbo_tn_begin')
        N = x.shape[0]
        for i in range(0, N, B):
            bt = min(B, N - i)
            ef = None if eps_f is None else eps_f[..., i:i + bt].contiguous()
            check(lib().vargp_elbo_tn_tile(ctypes.byref(d), ptr(seeds), ptr(x[i:i + bt]), ptr(y[i:i + bt]), ptr(ef), bt, st),
                  'vargp_elbo_tn_tile')
        check(lib().vargp_elbo_tn_end(ctypes.byref(d), ptr(seeds), *(ptr(g) for g in grads), st), 'vargp_elbo_tn_end')
        ops._note_chol_errors(self.info)
        return self.scalars


class _ElboTn(Function):
    @staticmethod
    def forward(ctx, log_mean, log_logvar, z, u_mean, u_tril_vec, x, y, eps_theta, eps_f, prior_log_mean,
                prior_log_logvar, map_est, prog, z_all, rk_all, eps_u=None):
        args = [t.contiguous() if t is not None else None
                for t in (log_mean, log_logvar, prior_log_mean, prior_log_logvar, z, u_mean, u_tril_vec)]
        scal = prog.forward(*args, z_all, rk_all, x.contiguous(), y.contiguous(),
                            None if eps_theta is None else eps_theta.contiguous(), eps_f.contiguous(),
                            eps_u=None if eps_u is None else eps_u.contiguous())
        # The workspace is owned by this node until its backward has run -- or until the node dies without one (validation
        # ELBO under no_grad: no graph is recorded and ctx is released as soon as apply() returns; a dropped graph; a
        # skipped step).  grad mode is always off inside Function.forward, so the node's lifetime is the signal.
        prog._gen += 1
        prog.busy = True
        weakref.finalize(ctx, _release, prog, prog._gen)
        ctx.gen = prog._gen
        ctx.prog = prog
        ctx.map_est = map_est
        ctx.shapes = (log_mean.shape, z.shape, u_mean.shape, u_tril_vec.shape)
        return scal[0].clone(), scal[1].clone(), scal[2].clone()

    @staticmethod
    @once_differentiable
    def backward(ctx, g_klh, g_klu, g_nll):
        prog = ctx.prog
        if prog._gen != ctx.gen:
            raise RuntimeError(_REUSED)
        if getattr(ctx, 'ran', False):
            prog.rerun_forward()
        ctx.ran = True
        seeds = torch.stack([g_klh.reshape(()), g_klu.reshape(()), g_nll.reshape(())]).float()
        sh_mean, sh_z, sh_um, sh_uv = ctx.shapes
        dev = seeds.device
        g_mean, g_logvar = torch.empty(sh_mean, device=dev), torch.empty(sh_mean, device=dev)
        g_z, g_um, g_uv = torch.empty(sh_z, device=dev), torch.empty(sh_um, device=dev), torch.empty(sh_uv, device=dev)
        prog.backward(seeds, g_mean, g_logvar, g_z, g_um, g_uv)
        _release(prog, ctx.gen)
        return (g_mean, None if ctx.map_est else g_logvar, g_z, g_um, g_uv) + (None,) * 11


def elbo_tn(kernel, z, u_mean, u_tril_vec, x, y, eps_theta, eps_f, prog, z_all, rk_all, eps_u=None):
    """-> (kl_hypers, kl_u, nll) of VARGP.loss for a model with previous tasks as ONE autograd node (eps_u: ep_var_mean = False)."""
    return _ElboTn.apply(kernel.log_mean, kernel.log_logvar, z, u_mean, u_tril_vec, x, y, eps_theta, eps_f,
                         kernel.prior_log_mean, kernel.prior_log_logvar, bool(kernel.map_est), prog, z_all, rk_all, eps_u)


# ----------------------------------------------------------------------------------------------------------------
# VARGP.loss on a native program WITHOUT an autograd graph: lazy terms (vargp_amd/lazy.py)
# ----------------------------------------------------------------------------------------------------------------
_RING = 8


def elbo_lazy(model, x, y, block):
    """(kl_hypers, kl_u, nll) of `model.loss(x, y)` as lazy terms over ONE forward of the model's cached program; the caller's
    linear combination and its `.backward()` become one program backward with those coefficients as seeds (lazy.py).
    Noise: injected / sharded draws if set (noise.py), otherwise the program's own counter-based generator under a key drawn
    from torch's default generator at the model's first step -- no torch.randn launches on the step."""
    from . import noise
    from .lazy import PendingForward, terms_of
    kern = model.kernel
    B = x.size(0)
    prog = model._tn_program(B) if block else model._t0_program(B)
    if getattr(prog, '_ring', None) is None:
        prog._ring, prog._ring_i = torch.zeros(_RING, 3, dtype=torch.float32, device=x.device), 0
        prog._ring_owner = [None] * _RING
    prog._ring_i = (prog._ring_i + 1) % _RING
    # the slot's previous forward, _RING steps back: if a term of it is still alive (`running += lik.detach()` read at the end
    # of the epoch, losses kept in a list), its three numbers move into a tensor of their own before the slot is reused -- a
    # term never silently reads another step's values (the copy is queued in front of this forward, same stream)
    old = prog._ring_owner[prog._ring_i]
    old = old() if old is not None else None
    if old is not None and old.values.data_ptr() == prog._ring[prog._ring_i].data_ptr():
        old.values = old.values.clone()
    prog.scalars = prog._ring[prog._ring_i]
    eps_u = model.draw_u_noise(x) if block else None          # ep_var_mean = False only (None otherwise)
    if noise._injected or noise._shard is not None or eps_u is not None:
        eps_theta, eps_f = model.draw_t0_noise(x)
        eps_theta = None if eps_theta is None else eps_theta.contiguous()
        eps_f = eps_f.contiguous()
    else:
        if prog._rng is None:
            if getattr(model, '_noise_counter', None) is None or model._noise_counter.device != x.device:
                model._noise_counter = torch.zeros(1, dtype=torch.int32, device=x.device)
                # the stream's key is DRAWN from torch's default generator when the model takes its first step: it follows
                # torch.manual_seed like any other draw, and every model of a process (one per task in the continual-learning
                # driver, the members of an ensemble) gets a stream of its own -- keyed by torch.initial_seed() all of them
                # replayed the same noise
                model._noise_seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
            prog.set_rng(model._noise_seed, model._noise_counter)
        eps_theta = eps_f = None
    x = x if x.is_contiguous() else x.contiguous()
    y = y if y.is_contiguous() else y.contiguous()
    args = (kern.log_mean.detach(), kern.log_logvar.detach(), kern.prior_log_mean, kern.prior_log_logvar, model.z.detach(),
            model.u_mean.detach(), model.u_tril_vec.detach())
    if block:
        prog.forward(*args, *model._tn_operands(), x, y, eps_theta, eps_f, eps_u=eps_u)
    else:
        prog.forward(*args, x, y, eps_theta, eps_f)
    params = (kern.log_mean, None if kern.map_est else kern.log_logvar, model.z, model.u_mean, model.u_tril_vec)
    fwd = PendingForward(model, prog, prog.scalars, params)
    prog._ring_owner[prog._ring_i] = weakref.ref(fwd)
    return terms_of(fwd)
