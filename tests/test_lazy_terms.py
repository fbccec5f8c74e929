"""CPU: the lazy loss terms of the drop-in loop (vargp_amd/lazy.py) against a mock program -- the arithmetic that stays lazy, the
seeds handed to the program's backward, accumulation into an existing .grad, one backward per forward, value reads, and the
fall-back that materialises real autograd tensors for everything else."""
import pytest
import torch

from vargp_amd.lazy import ElboTerm, PendingForward, terms_of


class _Prog:
    """Stands in for fused.T0Program: backward(seeds, five gradient buffers) fills buffer k with seeds . (k + 1, 10, 100)."""
    def __init__(self):
        self._gen, self.busy, self.calls, self.reruns = 0, False, [], 0

    def rerun_forward(self):
        self.reruns += 1

    def backward(self, seeds, *outs):
        self.calls.append(seeds.clone())
        for k, o in enumerate(outs):
            o.fill_(float(seeds[0] * (k + 1) + seeds[1] * 10 + seeds[2] * 100))


class _Model:
    def __init__(self, params):
        self.bufs = [[torch.zeros_like(p) for p in params], [torch.zeros_like(p) for p in params]]

    def _seed_tensor(self, coefs):
        return torch.tensor(coefs, dtype=torch.float32)

    def _grad_buffers(self):
        return self.bufs


def _setup(vals=(1.0, 2.0, 3.0)):
    params = [torch.nn.Parameter(torch.zeros(3)) for _ in range(5)]
    prog, model = _Prog(), _Model(params)
    fwd = PendingForward(model, prog, torch.tensor(vals), tuple(params))
    return params, prog, fwd, terms_of(fwd)


def test_reference_loop_combine_and_backward():
    params, prog, fwd, (kl_h, kl_u, lik) = _setup()
    assert prog.busy
    loss = 10.0 * kl_h + kl_u + (12000 / 512) * lik          # experiments/vargp.py:34
    assert isinstance(loss, ElboTerm) and loss.requires_grad
    assert loss.item() == pytest.approx(10.0 + 2.0 + 3.0 * 12000 / 512)
    loss.backward()
    assert len(prog.calls) == 1 and prog.calls[0].tolist() == pytest.approx([10.0, 1.0, 12000 / 512])
    for k, p in enumerate(params):                          # the program's buffers ARE the gradients now
        assert p.grad is fwd.model.bufs[0][k]
        assert p.grad[0].item() == pytest.approx(10.0 * (k + 1) + 10.0 + 100 * 12000 / 512)
    assert not prog.busy
    with pytest.raises(RuntimeError, match='backward through the graph a second time'):      # as autograd: no retain_graph, no second backward
        loss.backward()
    assert float(loss.detach()) == pytest.approx(loss.item()) and '%.2f' % kl_u == '2.00'


def test_retained_graph_takes_a_second_backward():
    """experiments/vargp.py:35 is ordinary autograd: loss.backward(retain_graph=True) and another backward are legal there.  Here
    the second one re-evaluates the program's forward (rerun_forward) and accumulates into .grad, and the workspace stays with
    the retained loss until that is dropped; a later loss() taking the workspace over ends it with a message."""
    params, prog, fwd, (kl_h, kl_u, lik) = _setup()
    loss = 2.0 * kl_h + kl_u
    loss.backward(retain_graph=True)
    assert prog.busy and prog.reruns == 0 and len(prog.calls) == 1
    g1 = params[0].grad[0].item()
    (kl_h + 0.5 * lik).backward(retain_graph=True)               # another combination of the same forward
    assert prog.reruns == 1 and len(prog.calls) == 2 and prog.calls[1].tolist() == pytest.approx([1.0, 0.0, 0.5])
    assert params[0].grad[0].item() == pytest.approx(g1 + 1.0 * 1 + 0.5 * 100)
    loss.backward()                                              # the last one releases the workspace
    assert prog.reruns == 2 and not prog.busy
    with pytest.raises(RuntimeError, match='second time'):
        loss.backward()
    # the workspace handed on (a later forward of the program) -> a clear error, not stale numbers
    params, prog, fwd, (a, b, c) = _setup()
    a.backward(retain_graph=True)
    prog._gen += 1
    with pytest.raises(RuntimeError, match='handed to a later loss'):
        a.backward()


def test_arithmetic_stays_lazy_and_accumulates_like_autograd():
    params, prog, fwd, (a, b, c) = _setup()
    t = -(a - 2 * b) / 4 + (c + 1.5) - 0.5 + sum([a, b])       # sum() starts from the int 0
    assert isinstance(t, ElboTerm)
    coef = {k: v for (f, k), v in t.terms.items()}
    assert coef == pytest.approx({0: 0.75, 1: 1.5, 2: 1.0}) and t.const == pytest.approx(1.0)
    for p in params:
        p.grad = torch.ones_like(p)                          # no zero_grad: the new gradients are ADDED, as autograd would
    t.backward()
    want0 = 0.75 * 1 + 1.5 * 10 + 1.0 * 100
    assert params[0].grad[0].item() == pytest.approx(1.0 + want0)
    assert params[0].grad is not fwd.model.bufs[0][0]


def test_two_forwards_in_one_expression():
    p1, prog1, f1, t1 = _setup((1.0, 1.0, 1.0))
    fwd2 = PendingForward(f1.model, _Prog(), torch.tensor([2.0, 2.0, 2.0]), f1.params)
    t2 = terms_of(fwd2)
    total = sum(t1) + 2 * sum(t2)
    assert total.item() == pytest.approx(3.0 + 12.0)
    total.backward()
    assert len(prog1.calls) == 1 and len(fwd2.prog.calls) == 1 and fwd2.prog.calls[0].tolist() == [2.0, 2.0, 2.0]
    assert p1[0].grad[0].item() == pytest.approx(111.0 + 222.0)       # second forward accumulated onto the first


def test_everything_else_materialises_real_tensors():
    params, prog, fwd, (a, b, c) = _setup()
    st = torch.stack([a.detach(), b.detach(), c.detach()])           # torch function on detached terms: plain values
    assert st.tolist() == [1.0, 2.0, 3.0] and not st.requires_grad
    assert (a < b) and bool(c) and a.shape == torch.Size([]) and a.dim() == 0
    w = torch.tensor(3.0)
    loss = a * w + torch.exp(b * 0.0)                                # tensor operand / torch function: autograd takes over
    assert isinstance(loss, torch.Tensor) and loss.requires_grad
    loss.backward()
    assert len(prog.calls) == 1 and prog.calls[0].tolist() == pytest.approx([3.0, 0.0, 0.0])
    assert params[1].grad[0].item() == pytest.approx(3.0 * 2)
    with pytest.raises(RuntimeError):
        (a + b).backward()                                           # the forward has been consumed


def test_backward_with_an_explicit_gradient_goes_through_autograd():
    params, prog, fwd, (a, b, c) = _setup()
    (a + 2 * c).backward(gradient=torch.tensor(0.5))
    assert prog.calls[0].tolist() == pytest.approx([0.5, 0.0, 1.0])
    with pytest.raises(RuntimeError):
        a.detach().backward()
