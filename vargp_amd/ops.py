"""torch.autograd.Function wrappers over the C ABI: every forward AND backward below is a call into
libvargp_hip.so (hand-written HIP), torch only owns the memory and the stream.
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib
from ._lib import GemmDesc, check, lib, ptr, require_device, scratch, stream_ptr

NONE, LOWER, UPPER = 0, 1, 2
_FLIP = {NONE: NONE, LOWER: UPPER, UPPER: LOWER}
JITTER = 1e-4

# Cholesky failures: 'raise' syncs after every factorisation (what torch.cholesky does, reference
# var_gp/gp_utils.py:10); 'defer' never syncs (hipGraph-capturable): failed factors are NaN-filled and
# the flag is accumulated on device, readable with linalg_error_count().
_chol_mode = 'raise'
_info_ring = []   # 'defer' mode: most recent info tensors (device), inspected on demand


_lazy_rings = {}
_pending = []     # 'lazy' mode: (pinned host copy of an info tensor, event recorded behind the copy)


def set_cholesky_error_mode(mode):
    """What happens when a factorisation meets a matrix that is not positive definite (the reference's torch.cholesky raises,
    gp_utils.py:5-11 -- which costs it a device synchronisation per call):
      'raise'  (default) the same: every factorising call waits for its status words and raises torch.linalg.LinAlgError;
      'lazy'   no wait: the status words are copied to pinned host memory behind the call, and the error is raised by the NEXT
               factorising call, by the first host read of one of the step's values (ElboTerm.item() / float()), or by
               check_linalg_errors() -- at most one step late, never lost;
      'defer'  nothing is read back until linalg_error_count() is asked for (captured hipGraphs: no host work inside a step)."""
    global _chol_mode
    assert mode in ('raise', 'lazy', 'defer')
    check_linalg_errors()
    _chol_mode = mode


def check_linalg_errors(wait=False):
    """'lazy' mode: raise for any failed factorisation whose status words have arrived (wait: all of them, synchronising)."""
    bad = None
    while _pending:
        host, ev = _pending[0]
        if wait:
            ev.synchronize()
        elif not ev.query():
            break
        _pending.pop(0)
        n = int((host != 0).sum())
        if n and bad is None:
            bad = (n, host.numel(), int(host[host != 0][0]))
    if bad is not None:
        del _pending[:]
        raise torch.linalg.LinAlgError(f'vargp_chol_inv: {bad[0]} of {bad[1]} matrices are not positive-definite '
                                       f'(first failing leading minor of order {bad[2]}; reported lazily)')


def reset_linalg_errors():
    del _info_ring[:]


def linalg_error_count_begin():
    """'defer' mode, for callers that synchronise anyway (the driver's one sync per epoch): ENQUEUE the count of failed
    factorisations and its copy to pinned memory on the current stream -> a handle; `int(handle)` after the caller's own
    synchronisation is the count.  (linalg_error_count() after a sync starts its reductions on an idle GPU and waits for them:
    ~0.1 ms per epoch that this form hides in the pipeline.)"""
    uniq = {}
    for t in _info_ring:
        uniq[(t.data_ptr(), t.numel())] = t
    slot = _lazy_rings.get('count')
    if slot is None:
        slot = _lazy_rings['count'] = torch.zeros(1, dtype=torch.int64, pin_memory=True)
    if not uniq:
        slot.zero_()
        return slot
    counts = [torch.count_nonzero(t) for t in uniq.values()]
    total = torch.stack(counts).sum() if len(counts) > 1 else counts[0]
    slot.copy_(total.view(1), non_blocking=True)
    return slot


def linalg_error_count():
    """'defer' mode: number of failed factorisations among the most recent calls.  ONE host sync: the ring usually holds the
    same few info tensors many times over (a program's info buffer is appended on every eager call), so they are counted
    once each, and the counts are summed on the device before the single read-back."""
    uniq = {}
    for t in _info_ring:
        uniq[(t.data_ptr(), t.numel())] = t
    if not uniq:
        return 0
    counts = [torch.count_nonzero(t) for t in uniq.values()]
    return int(torch.stack(counts).sum().item()) if len(counts) > 1 else int(counts[0].item())


# ------------------------------------------------------------------------------------------------
# low-level batched GEMM (no autograd)
# ------------------------------------------------------------------------------------------------
def _mat_layout(t):
    """-> (tensor, trans, ld): trans=0 if rows are contiguous (stride(-1)==1), 1 if it is a
    transposed view (stride(-2)==1); otherwise a contiguous copy is made."""
    r, c = t.shape[-2], t.shape[-1]
    s2, s1 = t.stride(-2), t.stride(-1)
    if s1 == 1 and (s2 >= c or r == 1):
        return t, 0, max(s2, c) if r > 1 else max(c, 1)
    if s2 == 1 and (s1 >= r or c == 1):
        return t, 1, max(s1, r) if c > 1 else max(r, 1)
    t = t.contiguous()
    return t, 0, max(c, 1)


def _batch3(shape, strides):
    """collapse batch dims to exactly 3 (pad in front); -> (sizes, strides) or None if > 3 dims"""
    if len(shape) > 3:
        return None
    pad = 3 - len(shape)
    return [1] * pad + list(shape), [0] * pad + list(strides)


def bgemm(A, B, alpha=1.0, D=None, beta=0.0, triA=NONE, triB=NONE, triC=NONE, out=None):
    """C = alpha * A @ B + beta * D on the MFMA GEMM; A: (..., M, K), B: (..., K, N) with
    broadcasting over the leading dims (views with stride 0 and .mT views are consumed in place)."""
    require_device(A, B, D)
    assert A.dtype == torch.float32 and B.dtype == torch.float32
    M, K = A.shape[-2:]
    K2, N = B.shape[-2:]
    assert K == K2, (A.shape, B.shape)
    bshape = torch.broadcast_shapes(A.shape[:-2], B.shape[:-2], D.shape[:-2] if D is not None else ())
    Ae = A.expand(*bshape, M, K)
    Be = B.expand(*bshape, K, N)
    Ae, tA, lda = _mat_layout(Ae)
    Be, tB, ldb = _mat_layout(Be)
    if len(bshape) > 3:
        Ae = Ae.reshape(-1, M, K)
        Be = Be.reshape(-1, K, N)
        Ae, tA, lda = _mat_layout(Ae)
        Be, tB, ldb = _mat_layout(Be)
    C = out if out is not None else torch.empty(*bshape, M, N, dtype=torch.float32, device=A.device)
    assert tuple(C.shape) == (*bshape, M, N) and (C.stride(-1) == 1 or N == 1), 'bgemm: bad `out`'
    Cv = C.reshape(-1, M, N) if len(bshape) > 3 else C
    nbs, sA = _batch3(Ae.shape[:-2], Ae.stride()[:-2])
    _, sB = _batch3(Be.shape[:-2], Be.stride()[:-2])
    _, sC = _batch3(Cv.shape[:-2], Cv.stride()[:-2])
    d = GemmDesc()
    d.M, d.N, d.K = M, N, K
    d.transA, d.transB = tA, tB
    d.A, d.B, d.C = Ae.data_ptr(), Be.data_ptr(), Cv.data_ptr()
    d.lda, d.ldb, d.ldc = lda, ldb, (max(Cv.stride(-2), N) if M > 1 else max(N, 1))
    keep = [Ae, Be]
    if D is not None:
        De = D.expand(*bshape, M, N)
        if De.stride(-1) != 1 and N > 1:
            De = De.contiguous()
        if len(bshape) > 3:
            De = De.reshape(-1, M, N)
        _, sD = _batch3(De.shape[:-2], De.stride()[:-2])
        d.D, d.ldd = De.data_ptr(), max(De.stride(-2), N) if M > 1 else max(N, 1)
        keep.append(De)
    else:
        sD = [0, 0, 0]
        d.D, d.ldd = None, 0
    for i in range(3):
        d.nb[i] = nbs[i]
        d.sA[i], d.sB[i], d.sC[i], d.sD[i] = sA[i], sB[i], sC[i], sD[i]
    d.alpha, d.beta = float(alpha), float(beta)
    d.triA, d.triB, d.triC = triA, triB, triC
    if M > 0 and N > 0 and C.numel() > 0:
        check(lib().vargp_bgemm(d, stream_ptr()), 'vargp_bgemm')
    return C


def _reduce_to(g, shape):
    """sum a broadcast gradient back to `shape` (leading-dim broadcasts go through vargp_sum_outer)."""
    if tuple(g.shape) == tuple(shape):
        return g
    lead = g.dim() - len(shape)
    # fold size-1 leading dims of `shape` into the reduction as long as everything behind them matches
    k = 0
    while k < len(shape) and shape[k] == 1 and g.shape[lead + k] != 1:
        k += 1
    if tuple(g.shape[lead + k:]) == tuple(shape[k:]):
        g = g.contiguous()
        outer = 1
        for s in g.shape[:lead + k]:
            outer *= s
        out = torch.empty(shape, dtype=g.dtype, device=g.device)
        check(lib().vargp_sum_outer(ptr(g), ptr(out), outer, out.numel(), stream_ptr()), 'vargp_sum_outer')
        return out
    # general (interior) broadcast: rare, tiny tensors
    if lead:
        g = g.sum(dim=tuple(range(lead)))
    dims = tuple(i for i, (a, b) in enumerate(zip(g.shape, shape)) if a != b)
    return g.sum(dim=dims, keepdim=True) if dims else g


class _MatMul(Function):
    @staticmethod
    def forward(ctx, A, B, D, alpha, beta, triA, triB, triC):
        ctx.save_for_backward(A, B)
        ctx.cfg = (alpha, beta, triA, triB, triC, D.shape if D is not None else None)
        return bgemm(A, B, alpha, D, beta, triA, triB, triC)

    @staticmethod
    @once_differentiable
    def backward(ctx, gC):
        A, B = ctx.saved_tensors
        alpha, beta, triA, triB, triC, dshape = ctx.cfg
        gA = gB = gD = None
        gC = gC.contiguous() if gC.stride(-1) != 1 and gC.stride(-2) != 1 else gC
        if ctx.needs_input_grad[0]:
            gA = bgemm(gC, B.mT, alpha, triB=_FLIP[triB], triC=LOWER if triA == LOWER else NONE)
            gA = _reduce_to(gA, A.shape)
        if ctx.needs_input_grad[1]:
            gB = bgemm(A.mT, gC, alpha, triA=_FLIP[triA], triC=LOWER if triB == LOWER else NONE)
            gB = _reduce_to(gB, B.shape)
        if dshape is not None and ctx.needs_input_grad[2]:
            gD = _reduce_to(gC * beta if beta != 1.0 else gC, dshape)
        return gA, gB, gD, None, None, None, None, None


def matmul(A, B, D=None, alpha=1.0, beta=1.0, triA=NONE, triB=NONE, triC=NONE):
    """alpha * A @ B (+ beta * D).  tri* are structure hints for the LOGICAL operands; triC=LOWER is
    only meaningful (and only allowed) for lower x lower products."""
    assert triC == NONE or (triA == LOWER and triB == LOWER)
    return _MatMul.apply(A, B, D, alpha, beta if D is not None else 0.0, triA, triB, triC)


# ------------------------------------------------------------------------------------------------
# RBF gram
# ------------------------------------------------------------------------------------------------
class _RbfGram(Function):
    @staticmethod
    def forward(ctx, theta, X, Y, y_shared):
        require_device(theta, X, Y)
        theta, X = theta.contiguous(), X.contiguous()
        Y = Y.contiguous() if Y is not None else None
        S, C, M, D = theta.shape[0], X.shape[0], X.shape[1], X.shape[2]
        N = M if Y is None else Y.shape[-2]
        K = torch.empty(S, C, M, N, dtype=torch.float32, device=X.device)
        ws = scratch(lib().vargp_rbf_workspace_bytes(S, C, M, N, D, 0), X.device)
        check(lib().vargp_rbf_gram_fwd(ptr(theta), ptr(X), ptr(Y), ptr(K), S, C, M, N, D, int(y_shared), ptr(ws),
                                       ws.numel() * 4, stream_ptr()), 'vargp_rbf_gram_fwd')
        ctx.save_for_backward(theta, X, Y, K)
        ctx.y_shared = int(y_shared)
        return K

    @staticmethod
    @once_differentiable
    def backward(ctx, gK):
        theta, X, Y, K = ctx.saved_tensors
        S, C, M, D = theta.shape[0], X.shape[0], X.shape[1], X.shape[2]
        N = M if Y is None else Y.shape[-2]
        gK = gK.contiguous()
        gX = torch.empty_like(X)
        want_gY = Y is not None and ctx.needs_input_grad[2]
        gY = torch.empty_like(Y) if want_gY else None
        gtheta = torch.empty_like(theta)
        ws = scratch(lib().vargp_rbf_workspace_bytes(S, C, M, N, D, 1), X.device)
        check(lib().vargp_rbf_gram_bwd(ptr(theta), ptr(X), ptr(Y), ptr(K), ptr(gK), ptr(gX), ptr(gY), ptr(gtheta),
                                       S, C, M, N, D, ctx.y_shared, 0, ptr(ws), ws.numel() * 4, stream_ptr()),
              'vargp_rbf_gram_bwd')
        return gtheta, gX, gY, None


def rbf_gram(theta, X, Y=None, y_shared=False):
    """theta (S,D+1); X (C,M,D); Y None | (C,N,D) | (N,D) with y_shared -> K (S,C,M,N)."""
    return _RbfGram.apply(theta, X, Y, y_shared)


# ------------------------------------------------------------------------------------------------
# Cholesky (+ jitter) with inverse factor
# ------------------------------------------------------------------------------------------------
def lazy_slot(n):
    """'lazy' mode: raise what earlier calls left behind, then hand out the next (pinned int32 host buffer, event) of the ring for n
    status words.  The caller gets the status words copied into the buffer and the event recorded behind them -- `_note_chol_errors`
    with a torch copy, the native programs by themselves (info_host / info_event: no torch launch at all) -- and appends the pair
    to `_pending`.  Buffers and events are recycled (allocating them per call cost ~40 us); every event has been recorded once,
    so that its native handle exists."""
    check_linalg_errors()
    ring = _lazy_rings.setdefault((torch.cuda.current_device(), n), [[], 0])       # (events belong to a device)
    if len(ring[0]) < 16:
        ev = torch.cuda.Event()
        ev.record()
        ring[0].append((torch.empty(n, dtype=torch.int32, pin_memory=True), ev))
    host, ev = ring[0][ring[1] % len(ring[0])]
    ring[1] += 1
    if any(h is host for h, _ in _pending):        # the ring has come round to a copy that has not been looked at: wait for it
        check_linalg_errors(wait=True)
    return host, ev


def raise_slot(n):
    """(pinned int32 host buffer of n status words, event) for the 'raise' mode, recycled per size.  The event has been recorded
    once, so that its native handle exists: the first-task program records it itself right behind its factorisation launch
    (include/vargp_hip.h: info_host / info_event) and the caller waits for THAT, not for the whole forward."""
    key = ('raise', torch.cuda.current_device(), n)                                 # (events belong to a device)
    slot = _lazy_rings.get(key)
    if slot is None:
        ev = torch.cuda.Event()
        ev.record()
        slot = _lazy_rings[key] = (torch.empty(n, dtype=torch.int32, pin_memory=True), ev)
    return slot


def _raise_if_bad(host):
    nz = host[host != 0]
    if nz.numel():
        raise torch.linalg.LinAlgError(
            f'vargp_chol_inv: {int(nz.numel())} of {host.numel()} matrices are not positive-definite '
            f'(first failing leading minor of order {int(nz[0])})')


def raise_wait(host, ev):
    ev.synchronize()
    _raise_if_bad(host)


def _note_chol_errors(info):
    if _chol_mode == 'lazy':
        host, ev = lazy_slot(info.numel())
        host.copy_(info.view(-1), non_blocking=True)
        ev.record()
        _pending.append((host, ev))
        return
    if _chol_mode == 'raise':
        # one copy into a (recycled) pinned buffer + an event wait: `(info != 0).sum().item()` was two reduction launches and a
        # synchronising read on every call
        if info.is_cuda:
            host, ev = raise_slot(info.numel())
            host.copy_(info.view(-1), non_blocking=True)
            ev.record()
            raise_wait(host, ev)
        else:
            _raise_if_bad(info.view(-1))
    else:
        _info_ring.append(info)
        del _info_ring[:-64]


class _CholInv(Function):
    @staticmethod
    def forward(ctx, A, eps, want_inv):
        require_device(A)
        ctx.set_materialize_grads(False)
        n = A.shape[-1]
        Ac = A.contiguous()
        nb = Ac.numel() // (n * n)
        L = torch.empty_like(Ac)
        need_T = want_inv or ctx.needs_input_grad[0]
        T = torch.empty_like(Ac) if need_T else None
        info = torch.empty(nb, dtype=torch.int32, device=A.device)
        ws = scratch(lib().vargp_chol_workspace_bytes(nb, n, 0), A.device)
        check(lib().vargp_chol_inv_fwd(ptr(Ac), float(eps), ptr(L), ptr(T), None, ptr(info), nb, n, ptr(ws),
                                       ws.numel() * 4, stream_ptr()), 'vargp_chol_inv_fwd')
        _note_chol_errors(info)
        ctx.save_for_backward(L, T)
        ctx.want_inv = want_inv
        if want_inv:
            return L, T
        dummy = L.new_empty(0)
        ctx.mark_non_differentiable(dummy)
        return L, dummy

    @staticmethod
    @once_differentiable
    def backward(ctx, gL, gT):
        L, T = ctx.saved_tensors
        if gL is None and (gT is None or not ctx.want_inv):
            return None, None, None
        n = L.shape[-1]
        nb = L.numel() // (n * n)
        gL = gL.contiguous() if gL is not None else None
        gT = gT.contiguous() if (ctx.want_inv and gT is not None) else None
        gA = torch.empty_like(L)
        ws = scratch(lib().vargp_chol_workspace_bytes(nb, n, 1), L.device)
        check(lib().vargp_chol_inv_bwd(ptr(L), ptr(T), ptr(gL), ptr(gT), ptr(gA), nb, n, ptr(ws), ws.numel() * 4,
                                       stream_ptr()), 'vargp_chol_inv_bwd')
        return gA, None, None


def chol_inv(A, eps=JITTER):
    """-> (L, T): L = chol(A + eps I) (lower), T = L^-1."""
    return _CholInv.apply(A, eps, True)


def chol(A, eps=JITTER):
    return _CholInv.apply(A, eps, False)[0]


# ------------------------------------------------------------------------------------------------
# packed triangle
# ------------------------------------------------------------------------------------------------
class _Vec2Tril(Function):
    @staticmethod
    def forward(ctx, vec, m):
        require_device(vec)
        vec = vec.contiguous()
        nb = vec.numel() // vec.shape[-1]
        out = torch.empty(*vec.shape[:-1], m, m, dtype=torch.float32, device=vec.device)
        check(lib().vargp_vec2tril_fwd(ptr(vec), ptr(out), nb, m, stream_ptr()), 'vargp_vec2tril_fwd')
        ctx.save_for_backward(vec)
        ctx.m = m
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        vec, = ctx.saved_tensors
        nb = vec.numel() // vec.shape[-1]
        gv = torch.empty_like(vec)
        check(lib().vargp_vec2tril_bwd(ptr(vec), ptr(g.contiguous()), ptr(gv), nb, ctx.m, stream_ptr()),
              'vargp_vec2tril_bwd')
        return gv, None


def vec2tril(vec, m):
    return _Vec2Tril.apply(vec, m)


def mat2trilvec(mat):
    require_device(mat)
    mat = mat.contiguous()
    m = mat.shape[-1]
    nb = mat.numel() // (m * m)
    out = torch.empty(*mat.shape[:-2], m * (m + 1) // 2, dtype=torch.float32, device=mat.device)
    check(lib().vargp_mat2trilvec(ptr(mat), ptr(out), nb, m, stream_ptr()), 'vargp_mat2trilvec')
    return out


# ------------------------------------------------------------------------------------------------
# predictive diag, MVN KL, log-det, likelihood
# ------------------------------------------------------------------------------------------------
class _PredictiveDiag(Function):
    @staticmethod
    def forward(ctx, P, W, a, kdiag):
        require_device(P, W, a, kdiag)
        P, W, a, kdiag = P.contiguous(), W.contiguous(), a.contiguous(), kdiag.contiguous()
        M, B = P.shape[-2:]
        nb = P.numel() // (M * B)
        mu = torch.empty(*P.shape[:-2], B, dtype=torch.float32, device=P.device)
        var = torch.empty_like(mu)
        check(lib().vargp_predictive_diag_fwd(ptr(P), ptr(W), ptr(a), 1, M, ptr(kdiag), ptr(mu), ptr(var), nb, M, B,
                                              stream_ptr()), 'vargp_predictive_diag_fwd')
        ctx.save_for_backward(P, W, a)
        return mu, var

    @staticmethod
    @once_differentiable
    def backward(ctx, gmu, gvar):
        P, W, a = ctx.saved_tensors
        M, B = P.shape[-2:]
        nb = P.numel() // (M * B)
        gP, gW, ga = torch.empty_like(P), torch.empty_like(W), torch.empty_like(a)
        gk = torch.empty(P.shape[:-2], dtype=torch.float32, device=P.device)
        check(lib().vargp_predictive_diag_bwd(ptr(P), ptr(W), ptr(a), 1, M, ptr(gmu.contiguous()), ptr(gvar.contiguous()),
                                              ptr(gP), ptr(gW), ptr(ga), ptr(gk), nb, M, B, stream_ptr()),
              'vargp_predictive_diag_bwd')
        return gP, gW, ga, gk


def predictive_diag(P, W, a, kdiag):
    """P, W (..., M, B); a (..., M); kdiag (...) -> mu, var (..., B)."""
    return _PredictiveDiag.apply(P, W, a, kdiag)


class _LogdetTril(Function):
    @staticmethod
    def forward(ctx, L):
        require_device(L)
        L = L.contiguous()
        n = L.shape[-1]
        nb = L.numel() // (n * n)
        out = torch.empty(L.shape[:-2], dtype=torch.float32, device=L.device)
        check(lib().vargp_logdet_tril_fwd(ptr(L), ptr(out), nb, n, stream_ptr()), 'vargp_logdet_tril_fwd')
        ctx.save_for_backward(L)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        L, = ctx.saved_tensors
        n = L.shape[-1]
        nb = L.numel() // (n * n)
        gL = torch.empty_like(L)
        check(lib().vargp_logdet_tril_bwd(ptr(L), ptr(g.contiguous()), ptr(gL), nb, n, stream_ptr()),
              'vargp_logdet_tril_bwd')
        return gL


def logdet_tril(L):
    return _LogdetTril.apply(L)


class _MvnKl(Function):
    @staticmethod
    def forward(ctx, G, d, ldp, ldq):
        require_device(G, d, ldp, ldq)
        G, d, ldp, ldq = G.contiguous(), d.contiguous(), ldp.contiguous(), ldq.contiguous()
        M = G.shape[-1]
        nb = G.numel() // (M * M)
        kl = torch.empty(G.shape[:-2], dtype=torch.float32, device=G.device)
        check(lib().vargp_mvn_kl_fwd(ptr(G), ptr(d), ptr(ldp), ptr(ldq), ptr(kl), nb, M, stream_ptr()),
              'vargp_mvn_kl_fwd')
        ctx.save_for_backward(G, d)
        return kl

    @staticmethod
    @once_differentiable
    def backward(ctx, gkl):
        G, d = ctx.saved_tensors
        M = G.shape[-1]
        nb = G.numel() // (M * M)
        gkl = gkl.contiguous()
        gG, gd = torch.empty_like(G), torch.empty_like(d)
        check(lib().vargp_mvn_kl_bwd(ptr(G), ptr(d), ptr(gkl), ptr(gG), ptr(gd), nb, M, stream_ptr()),
              'vargp_mvn_kl_bwd')
        return gG, gd, gkl, -gkl


def mvn_kl_from_factors(G, d, logdet_p, logdet_q):
    """KL per batch element from G = Lp^-1 Lq (..., M, M), d = Lp^-1 (mu_q - mu_p) (..., M) and the two
    log-dets (...)."""
    return _MvnKl.apply(G, d, logdet_p, logdet_q)


class _SoftmaxNll(Function):
    @staticmethod
    def forward(ctx, mu, var, eps, y):
        require_device(mu, var, eps, y)
        mu, var, eps, y = mu.contiguous(), var.contiguous(), eps.contiguous(), y.contiguous()
        assert y.dtype == torch.int64
        S, F, C, B = eps.shape
        assert mu.shape == (S, C, B) and var.shape == (S, C, B), (mu.shape, eps.shape)
        nll = torch.empty((), dtype=torch.float32, device=mu.device)
        check(lib().vargp_softmax_nll_fwd(ptr(mu), ptr(var), ptr(eps), ptr(y), ptr(nll), S, F, C, B, stream_ptr()),
              'vargp_softmax_nll_fwd')
        ctx.save_for_backward(mu, var, eps, y)
        return nll

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        mu, var, eps, y = ctx.saved_tensors
        S, F, C, B = eps.shape
        gmu, gvar = torch.empty_like(mu), torch.empty_like(var)
        check(lib().vargp_softmax_nll_bwd(ptr(mu), ptr(var), ptr(eps), ptr(y), ptr(g.contiguous()), ptr(gmu),
                                          ptr(gvar), S, F, C, B, stream_ptr()), 'vargp_softmax_nll_bwd')
        return gmu, gvar, None, None


def softmax_nll(mu, var, eps, y):
    return _SoftmaxNll.apply(mu, var, eps, y)


def softmax_predict(mu, var, eps):
    require_device(mu, var, eps)
    mu, var, eps = mu.contiguous(), var.contiguous(), eps.contiguous()
    S, F, C, B = eps.shape
    probs = torch.empty(B, C, dtype=torch.float32, device=mu.device)
    check(lib().vargp_softmax_predict(ptr(mu), ptr(var), ptr(eps), ptr(probs), S, F, C, B, stream_ptr()),
          'vargp_softmax_predict')
    return probs


# ------------------------------------------------------------------------------------------------
# variational hyper-parameters
# ------------------------------------------------------------------------------------------------
class _HyperSample(Function):
    @staticmethod
    def forward(ctx, mean, logvar, eps):
        require_device(mean, logvar, eps)
        mean, logvar, eps = mean.contiguous(), logvar.contiguous(), eps.contiguous()
        S, D1 = eps.shape
        theta = torch.empty_like(eps)
        check(lib().vargp_hyper_sample_fwd(ptr(mean), ptr(logvar), ptr(eps), ptr(theta), S, D1, stream_ptr()),
              'vargp_hyper_sample_fwd')
        ctx.save_for_backward(logvar, eps)
        return theta

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        logvar, eps = ctx.saved_tensors
        S, D1 = eps.shape
        gm, gv = torch.empty_like(logvar), torch.empty_like(logvar)
        check(lib().vargp_hyper_sample_bwd(ptr(logvar), ptr(eps), ptr(g.contiguous()), ptr(gm), ptr(gv), S, D1,
                                           stream_ptr()), 'vargp_hyper_sample_bwd')
        return gm, gv, None


def hyper_sample(mean, logvar, eps):
    return _HyperSample.apply(mean, logvar, eps)


class _HyperKl(Function):
    @staticmethod
    def forward(ctx, mean, logvar, pmean, plogvar):
        require_device(mean, logvar, pmean, plogvar)
        t = [x.contiguous() for x in (mean, logvar, pmean, plogvar)]
        kl = torch.empty((), dtype=torch.float32, device=mean.device)
        check(lib().vargp_hyper_kl_fwd(*(ptr(x) for x in t), ptr(kl), mean.numel(), stream_ptr()), 'vargp_hyper_kl_fwd')
        ctx.save_for_backward(*t)
        return kl

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        t = ctx.saved_tensors
        gm, gv = torch.empty_like(t[0]), torch.empty_like(t[1])
        check(lib().vargp_hyper_kl_bwd(*(ptr(x) for x in t), ptr(g.contiguous()), ptr(gm), ptr(gv), t[0].numel(),
                                       stream_ptr()), 'vargp_hyper_kl_bwd')
        return gm, gv, None, None


def hyper_kl(mean, logvar, prior_mean, prior_logvar):
    return _HyperKl.apply(mean, logvar, prior_mean, prior_logvar)


# ------------------------------------------------------------------------------------------------
# deep-kernel feature map: Linear (+ ReLU) on the MFMA GEMM with a fused bias / activation pass
# ------------------------------------------------------------------------------------------------
class _LinearAct(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        require_device(x, weight, bias)
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        h = bgemm(x2, weight.mT)                                   # (rows, out)
        y = torch.empty_like(h)
        check(lib().vargp_bias_act_fwd(ptr(h), ptr(bias.contiguous()), ptr(y), h.shape[0], h.shape[1], int(relu),
                                       stream_ptr()), 'vargp_bias_act_fwd')
        ctx.save_for_backward(x2, weight, y)
        ctx.relu, ctx.xshape = bool(relu), x.shape
        return y.reshape(*x.shape[:-1], weight.shape[0])

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x2, weight, y = ctx.saved_tensors
        gy2 = gy.reshape(-1, weight.shape[0]).contiguous()
        gh = torch.empty_like(gy2)
        gb = torch.empty(weight.shape[0], dtype=torch.float32, device=gy.device)
        check(lib().vargp_bias_act_bwd(ptr(y), ptr(gy2), ptr(gh), ptr(gb), gy2.shape[0], gy2.shape[1], int(ctx.relu),
                                       stream_ptr()), 'vargp_bias_act_bwd')
        gx = bgemm(gh, weight).reshape(ctx.xshape) if ctx.needs_input_grad[0] else None
        gw = bgemm(gh.mT, x2) if ctx.needs_input_grad[1] else None
        return gx, gw, gb, None


def linear_act(x, weight, bias, relu):
    """act(x @ weight^T + bias) over the last dim of x; weight (out, in) as torch.nn.Linear stores it."""
    return _LinearAct.apply(x, weight, bias, relu)


# ------------------------------------------------------------------------------------------------
# triangular solve against a factor that came with its inverse (an op of its own in SURVEY §8b's list)
# ------------------------------------------------------------------------------------------------
class _TrsmLower(Function):
    @staticmethod
    def forward(ctx, L, T, B):
        require_device(L, T, B)
        T, B = T.contiguous(), B.contiguous()
        n, nrhs = B.shape[-2:]
        nb = B.numel() // (n * nrhs)
        assert T.shape[-1] == n and T.numel() == nb * n * n, 'trsm_lower: one factor per right-hand side block'
        X = torch.empty_like(B)
        check(lib().vargp_trsm_lower_fwd(ptr(T), ptr(B), ptr(X), nb, n, nrhs, stream_ptr()), 'vargp_trsm_lower_fwd')
        ctx.save_for_backward(T, X)
        return X

    @staticmethod
    @once_differentiable
    def backward(ctx, gX):
        T, X = ctx.saved_tensors
        n, nrhs = X.shape[-2:]
        nb = X.numel() // (n * nrhs)
        gB = torch.empty_like(X)
        gL = torch.empty_like(T) if ctx.needs_input_grad[0] else None
        check(lib().vargp_trsm_lower_bwd(ptr(T), ptr(X), ptr(gX.contiguous()), ptr(gB), ptr(gL), nb, n, nrhs, None, 0,
                                         stream_ptr()), 'vargp_trsm_lower_bwd')
        return gL, None, gB


def trsm_lower(L, T, B):
    """X = L^-1 B for a factor L that came with T = L^-1 from chol_inv (gradients flow to L and B; T is L's inverse,
    not an independent input)."""
    return _TrsmLower.apply(L, T, B)
