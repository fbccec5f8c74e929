"""Lazy scalars for the drop-in training loop.

The reference's loop (experiments/vargp.py:29-37) does

    kl_hypers, kl_u, lik = gp.loss(x, y)
    loss = beta * kl_hypers + kl_u + (N / x.size(0)) * lik
    loss.backward()

On the native programs the three numbers come out of ONE forward and the whole backward is ONE call that takes the three
coefficients of that linear combination as its seeds.  Handing back three autograd tensors makes the caller's two lines cost
four tiny device kernels, their four autograd nodes, the engine's hand-over to its device thread and a `torch.stack` of the
incoming gradients -- ~140 us of host time per step against ~200 us of kernels for the whole step.  `ElboTerm` is what
`VARGP.loss` returns instead: a linear combination  sum_k coef_k * scalar_k + const  over the scalars of pending program
forwards, combined by plain Python arithmetic; `.backward()` hands the coefficients straight to the program's backward, which
writes the gradients into the parameters' `.grad` (accumulating when a `.grad` is already there, as autograd does).

Everything a caller may do with a 0-dim tensor beyond that still works, because anything else *materialises* the term into a
real autograd tensor (one node over the five parameters whose backward is the same program call): `torch.*` functions
(`__torch_function__`), multiplication by tensors or other terms, `backward(gradient=...)`, `.grad_fn`, ...  `.item()`,
`float()`, `.detach()`, `.cpu()`, `.tolist()` read the device scalars (one sync), without autograd.
"""
import weakref

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

_NUM = (int, float)


class PendingForward:
    """One program forward whose backward has not run: owns the program's workspace until it has, or until it dies."""

    def __init__(self, model, prog, values, params):
        self.model = model                      # the VARGP module (gradient buffers live there)
        self.prog = prog
        self.values = values                    # (3,) device view: kl_hypers, kl_u, nll of this forward (a slot of the program's ring)
        self.params = params                    # (log_mean, log_logvar or None, z, u_mean, u_tril_vec): the leaves
        self.done = False
        self.retained = False                   # the last backward asked for retain_graph: a further one re-evaluates the forward
        prog._gen += 1
        self.gen = prog._gen
        prog.busy = True
        weakref.finalize(self, _release, prog, prog._gen)

    def run_backward(self, coefs, retain=False):
        """d total / d (kl_hypers, kl_u, nll) = coefs (host floats) -> gradients into the parameters' .grad.  A second call is
        legal after `retain=True` (loss.backward(retain_graph=True), as in autograd): the program's forward is re-evaluated on
        the same operands and noise (fused.T0Program.rerun_forward) and the workspace stays with this forward until it dies."""
        if self.prog._gen != self.gen:
            from .fused import _REUSED
            raise RuntimeError(_REUSED)
        if self.done:
            if not self.retained:
                raise RuntimeError('Trying to backward through the graph a second time (VARGP.loss on the native program: the '
                                   'forward is re-evaluated for a second backward only when the first one was called with '
                                   'retain_graph=True)')
            self.prog.rerun_forward()
        self.done = True
        self.retained = bool(retain)
        m = self.model
        seeds = m._seed_tensor(coefs)
        bufs = m._grad_buffers()
        fresh = [p is None or p.grad is None for p in self.params]
        outs = [b if f else s for b, s, f in zip(bufs[0], bufs[1], fresh)]          # straight into the buffer that becomes .grad,
        self.prog.backward(seeds, *outs)                                               # or into scratch when a .grad is there already
        for p, o, f in zip(self.params, outs, fresh):
            if p is None:
                continue
            if f:
                p.grad = o
            else:
                p.grad.add_(o.view_as(p.grad))
        if not retain:
            _release(self.prog, self.gen)


def _release(prog, gen):
    if prog._gen == gen:
        prog.busy = False


class _Materialise(Function):
    """The three scalars of a pending forward as real autograd tensors (fallback for everything ElboTerm does not do itself)."""

    @staticmethod
    def forward(ctx, fwd, *params):
        ctx.fwd = fwd
        ctx.none = [p is None for p in fwd.params]
        v = fwd.values.clone()
        return v[0], v[1], v[2]

    @staticmethod
    @once_differentiable
    def backward(ctx, g0, g1, g2):
        fwd = ctx.fwd
        if fwd.prog._gen != fwd.gen:
            from .fused import _REUSED
            raise RuntimeError(_REUSED)
        if fwd.done:
            fwd.prog.rerun_forward()             # retained graph, second backward (autograd itself has let it through)
        fwd.done = True
        seeds = torch.stack([g0.reshape(()), g1.reshape(()), g2.reshape(())]).float()
        outs = [torch.empty_like(p) if p is not None else fwd.model._grad_buffers()[1][i] for i, p in enumerate(fwd.params)]
        fwd.prog.backward(seeds, *outs)
        if not fwd.retained:
            _release(fwd.prog, fwd.gen)
        return (None,) + tuple(o for o, p in zip(outs, fwd.params) if p is not None)


class ElboTerm:
    """sum_k coef * scalar_k(forward) + const, see the module docstring.  Immutable."""
    __slots__ = ('terms', 'const', 'nograd', '_real')
    __array_priority__ = 1000

    def __init__(self, terms, const=0.0, nograd=False):
        self.terms = terms            # {(PendingForward, k): coef}
        self.const = const
        self.nograd = nograd          # detached: value only
        self._real = None

    # -- arithmetic that stays lazy ---------------------------------------------------------------------------------------------
    def _lin(self, other, sa, sb):
        if isinstance(other, _NUM):
            if sa == 1.0:
                return ElboTerm(self.terms, self.const + sb * other, self.nograd)
            return ElboTerm({k: sa * c for k, c in self.terms.items()}, sa * self.const + sb * other, self.nograd)
        if isinstance(other, ElboTerm) and other.nograd == self.nograd:
            t = {k: sa * c for k, c in self.terms.items()} if sa != 1.0 else dict(self.terms)
            for k, c in other.terms.items():
                t[k] = t.get(k, 0.0) + sb * c
            return ElboTerm(t, sa * self.const + sb * other.const, self.nograd)
        return NotImplemented

    def __add__(self, o):
        r = self._lin(o, 1.0, 1.0)
        return self.tensor() + _real(o) if r is NotImplemented else r
    __radd__ = __add__

    def __sub__(self, o):
        r = self._lin(o, 1.0, -1.0)
        return self.tensor() - _real(o) if r is NotImplemented else r

    def __rsub__(self, o):
        r = self._lin(o, -1.0, 1.0)
        return _real(o) - self.tensor() if r is NotImplemented else r

    def __mul__(self, o):
        if isinstance(o, _NUM):
            return ElboTerm({k: c * o for k, c in self.terms.items()}, self.const * o, self.nograd)
        return self.tensor() * _real(o)
    __rmul__ = __mul__

    def __truediv__(self, o):
        if isinstance(o, _NUM):
            return self * (1.0 / o)
        return self.tensor() / _real(o)

    def __rtruediv__(self, o):
        return _real(o) / self.tensor()

    def __neg__(self):
        return self * -1.0

    def __pos__(self):
        return self

    # -- autograd ---------------------------------------------------------------------------------------------------------------
    def backward(self, gradient=None, retain_graph=None, create_graph=False, inputs=None):
        if self.nograd:
            raise RuntimeError('element 0 of tensors does not require grad and does not have a grad_fn')
        if gradient is not None or create_graph or inputs is not None or self._real is not None:
            for fwd, _ in self.terms:
                fwd.retained = bool(retain_graph)
            return self.tensor().backward(gradient, retain_graph, create_graph, inputs)
        per = {}
        for (fwd, k), c in self.terms.items():
            per.setdefault(fwd, [0.0, 0.0, 0.0])[k] += c
        for fwd, coefs in per.items():
            fwd.run_backward(tuple(coefs), retain=bool(retain_graph))

    @property
    def requires_grad(self):
        return not self.nograd

    @property
    def grad_fn(self):
        return self.tensor().grad_fn

    def tensor(self):
        """The term as a real 0-dim tensor (autograd-connected unless detached)."""
        if self._real is None:
            acc = None
            for fwd in {f for f, _ in self.terms}:
                if self.nograd or fwd.done or not torch.is_grad_enabled():
                    vals = fwd.values.clone()
                else:
                    if getattr(fwd, '_real_vals', None) is None:
                        fwd._real_vals = _Materialise.apply(fwd, *[p for p in fwd.params if p is not None])
                    vals = fwd._real_vals
                for (f, k), c in self.terms.items():
                    if f is fwd:
                        t = vals[k] * c if c != 1.0 else vals[k]
                        acc = t if acc is None else acc + t
            if acc is None:
                acc = torch.zeros(())
            self._real = acc + self.const if self.const != 0.0 else acc
        return self._real

    # -- value only ---------------------------------------------------------------------------------------------------------------
    def item(self):
        from . import ops
        if self._real is not None:
            v = self._real.item()
        else:
            cache = {}
            v = self.const
            for (fwd, k), c in self.terms.items():
                if fwd not in cache:
                    cache[fwd] = fwd.values.tolist()       # one read-back per forward (synchronises)
                v += c * cache[fwd][k]
        if ops._pending:                                   # 'lazy' Cholesky error mode: the read-back above waited for the step
            ops.check_linalg_errors(wait=True)
        return v

    def __float__(self):
        return float(self.item())

    def __format__(self, spec):
        return format(self.item(), spec)

    def __repr__(self):
        return f'ElboTerm({self.item()!r})'

    def tolist(self):
        return self.item()

    def detach(self):
        return ElboTerm(self.terms, self.const, True)

    @property
    def data(self):
        return self.detach().tensor()

    def clone(self):
        return self

    def cpu(self):
        return torch.tensor(self.item())

    def numpy(self):
        return self.cpu().numpy()

    def dim(self):
        return 0

    ndim = property(lambda self: 0)
    shape = property(lambda self: torch.Size([]))
    dtype = property(lambda self: torch.float32)

    @property
    def device(self):
        for fwd, _ in self.terms:
            return fwd.values.device
        return torch.device('cpu')

    def size(self, *a):
        return torch.Size([])

    def numel(self):
        return 1

    def to(self, *a, **k):
        return self.tensor().to(*a, **k)

    def __bool__(self):
        return bool(self.item())

    def _cmp(self, o, op):
        return op(self.item(), o.item() if hasattr(o, 'item') else o)

    def __lt__(self, o): return self._cmp(o, lambda a, b: a < b)
    def __le__(self, o): return self._cmp(o, lambda a, b: a <= b)
    def __gt__(self, o): return self._cmp(o, lambda a, b: a > b)
    def __ge__(self, o): return self._cmp(o, lambda a, b: a >= b)

    def __getattr__(self, name):
        # any other tensor attribute / method: on the materialised tensor
        if name.startswith('__'):
            raise AttributeError(name)
        return getattr(self.tensor(), name)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        conv = lambda a: (a.tensor() if isinstance(a, ElboTerm)
                          else type(a)(conv(v) for v in a) if isinstance(a, (list, tuple)) else a)
        return func(*conv(args), **{k: conv(v) for k, v in (kwargs or {}).items()})


def _real(o):
    return o.tensor() if isinstance(o, ElboTerm) else o


def terms_of(fwd):
    """(kl_hypers, kl_u, nll) of a pending forward as lazy terms."""
    return tuple(ElboTerm({(fwd, k): 1.0}) for k in range(3))
