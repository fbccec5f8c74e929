"""Fused task-0 ELBO: `VARGP.loss` for a model without previous tasks as ONE autograd node whose forward
and backward are explicit sequences of C-ABI calls (same kernels as ops.py, hand-orchestrated).

What the orchestration buys over composing the per-op autograd Functions (gp_utils.py):
  * K_uu (S*C matrices) and S_u = Lu Lu^T (C matrices) are factorised by ONE vargp_chol_inv_fwd launch and
    differentiated by ONE vargp_chol_inv_bwd call (the factorisation is a latency-bound chain of M pivots);
  * the small right-hand sides that share Lz^-1 — u_mean, L_S = chol(S_u + eps I) and Lu — are packed into one
    operand, so Lz^-1 m, Lz^-1 L_S and Lz^-1 Lu (KL) come out of one GEMM; the KL and its backward read them
    in place;
  * gradient contributions to the same tensor (T = Lz^-1 gets four) are accumulated by the GEMM epilogue
    (C = alpha A B + beta D) instead of separate add kernels; the two RBF backward calls accumulate into one
    z / theta gradient.
Numerics are those of the composed path (same kernels, same order of operations inside each).
Reference lines: var_gp/vargp.py:156-194 (forward/loss for the first task), gp_utils.py:150-191.
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import ops
from ._lib import check, lib, ptr, require_device, stream_ptr, workspace
from .ops import JITTER, LOWER, UPPER, bgemm


_DEBUG_KEEP = None


_side_streams = {}


def _side_stream(device):
    """Second HIP stream per device: the K_uf kernel-matrix work (forward and backward) has no dependency on
    the K_uu -> Cholesky chain, which is a latency-bound sequence on a few CUs, so the two run concurrently.
    Forks and joins are stream waits, which a hipGraph capture records as graph edges."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _side_streams:
        _side_streams[idx] = torch.cuda.Stream(device=device)
    return _side_streams[idx]


def _rbf_ws(theta, X, Y, backward):
    S, (C, M, D) = theta.shape[0], X.shape
    N = M if Y is None else Y.shape[-2]
    return workspace(lib().vargp_rbf_workspace_bytes(S, C, M, N, D, int(backward)), X.device)


def _rbf_fwd(theta, X, Y, shared, out, ws=None):
    S, (C, M, D) = theta.shape[0], X.shape
    N = M if Y is None else Y.shape[-2]
    ws = ws if ws is not None else _rbf_ws(theta, X, Y, False)
    check(lib().vargp_rbf_gram_fwd(ptr(theta), ptr(X), ptr(Y), ptr(out), S, C, M, N, D, int(shared), ptr(ws),
                                   ws.numel() * 4, stream_ptr()), 'vargp_rbf_gram_fwd')


def _rbf_bwd(theta, X, Y, shared, K, gK, gX, gtheta, accumulate, ws=None):
    S, (C, M, D) = theta.shape[0], X.shape
    N = M if Y is None else Y.shape[-2]
    ws = ws if ws is not None else _rbf_ws(theta, X, Y, True)
    check(lib().vargp_rbf_gram_bwd(ptr(theta), ptr(X), ptr(Y), ptr(K), ptr(gK), ptr(gX), None, ptr(gtheta),
                                   S, C, M, N, D, int(shared), int(accumulate), ptr(ws), ws.numel() * 4, stream_ptr()),
          'vargp_rbf_gram_bwd')


class _ElboT0(Function):
    @staticmethod
    def forward(ctx, theta, z, u_mean, u_tril_vec, x, y, eps_f):
        require_device(theta, z, u_mean, u_tril_vec, x, y, eps_f)
        theta, z, u_mean, u_tril_vec, x, y, eps_f = (t.contiguous() for t in (theta, z, u_mean, u_tril_vec, x, y, eps_f))
        S, (C, M, D), B = theta.shape[0], z.shape, x.shape[0]
        SC, NR, dev, st = S * C, 4 + 2 * M, z.device, stream_ptr()
        f32 = dict(dtype=torch.float32, device=dev)

        # fork: K_uf = rbf(z, x) on the side stream (buffers allocated here, on the main stream, and kept
        # alive past the join)
        main, side = torch.cuda.current_stream(), _side_stream(dev)
        Kuf = torch.empty(S, C, M, B, **f32)
        ws_uf = _rbf_ws(theta, z, x, False)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            _rbf_fwd(theta, z, x, True, Kuf, ws_uf)

        Lu = torch.empty(C, M, M, **f32)
        check(lib().vargp_vec2tril_fwd(ptr(u_tril_vec), ptr(Lu), C, M, st), 'vargp_vec2tril_fwd')
        # K_uu for every (s, c) and S_u = Lu Lu^T for every c, one batch -> one factorisation
        KS = torch.empty(SC + C, M, M, **f32)
        _rbf_fwd(theta, z, None, False, KS)
        bgemm(Lu, Lu.mT, triA=LOWER, triB=UPPER, out=KS[SC:])
        LL, TT = torch.empty_like(KS), torch.empty_like(KS)
        info = torch.empty(SC + C, dtype=torch.int32, device=dev)
        check(lib().vargp_chol_inv_fwd(ptr(KS), JITTER, ptr(LL), ptr(TT), None, ptr(info), SC + C, M, None, 0, st)
              if M <= 100 else _chol_big(KS, LL, TT, info, SC + C, M), 'vargp_chol_inv_fwd')
        ops._note_chol_errors(info)
        Lz, Tz, LS = LL[:SC].view(S, C, M, M), TT[:SC].view(S, C, M, M), LL[SC:]

        R = torch.empty(C, M, NR, **f32)
        check(lib().vargp_pack_rsmall(ptr(u_mean), ptr(LS), ptr(Lu), ptr(R), C, M, st), 'vargp_pack_rsmall')
        Q = bgemm(Tz, R, triA=LOWER)                      # (S, C, M, NR) = [a | 0 0 0 | G | G2]
        main.wait_stream(side)                            # join: K_uf is needed from here on
        del ws_uf
        P = bgemm(Tz, Kuf, triA=LOWER)                    # Lz^-1 K_uf
        G = Q[..., 4:4 + M]
        W = bgemm(G.mT, P, triA=UPPER)                    # (Lz^-1 L_S)^T Lz^-1 K_uf

        kd = (2.0 * theta[:, -1]).exp().unsqueeze(1).expand(S, C).contiguous()   # gamma^2 per (s, c)
        mu, var = torch.empty(S, C, B, **f32), torch.empty(S, C, B, **f32)
        check(lib().vargp_predictive_diag_fwd(ptr(P), ptr(W), ptr(Q), NR, M * NR, ptr(kd), ptr(mu), ptr(var), SC, M, B,
                                              st), 'vargp_predictive_diag_fwd')
        F_ = eps_f.shape[1]
        nll = torch.empty((), **f32)
        check(lib().vargp_softmax_nll_fwd(ptr(mu), ptr(var), ptr(eps_f), ptr(y), ptr(nll), S, F_, C, B, st),
              'vargp_softmax_nll_fwd')
        kl_u = torch.empty((), **f32)
        check(lib().vargp_kl_t0_fwd(ptr(Q), ptr(Lz), ptr(Lu), ptr(kl_u), S, C, M, st), 'vargp_kl_t0_fwd')

        ctx.save_for_backward(theta, z, u_tril_vec, x, y, eps_f, Lu, KS, LL, TT, Kuf, R, Q, P, W, mu, var)
        return nll, kl_u

    @staticmethod
    @once_differentiable
    def backward(ctx, g_nll, g_kl):
        theta, z, u_tril_vec, x, y, eps_f, Lu, KS, LL, TT, Kuf, R, Q, P, W, mu, var = ctx.saved_tensors
        S, (C, M, D), B = theta.shape[0], z.shape, x.shape[0]
        SC, NR, dev, st = S * C, 4 + 2 * M, z.device, stream_ptr()
        f32 = dict(dtype=torch.float32, device=dev)
        F_ = eps_f.shape[1]
        Lz, Tz = LL[:SC].view(S, C, M, M), TT[:SC].view(S, C, M, M)
        g_nll, g_kl = g_nll.contiguous(), g_kl.contiguous()

        gmu, gvar = torch.empty_like(mu), torch.empty_like(var)
        check(lib().vargp_softmax_nll_bwd(ptr(mu), ptr(var), ptr(eps_f), ptr(y), ptr(g_nll), ptr(gmu), ptr(gvar),
                                          S, F_, C, B, st), 'vargp_softmax_nll_bwd')
        gP, gW = torch.empty_like(P), torch.empty_like(W)
        ga, gkd = torch.empty(S, C, M, **f32), torch.empty(S, C, **f32)
        check(lib().vargp_predictive_diag_bwd(ptr(P), ptr(W), ptr(Q), NR, M * NR, ptr(gmu), ptr(gvar), ptr(gP), ptr(gW),
                                              ptr(ga), ptr(gkd), SC, M, B, st), 'vargp_predictive_diag_bwd')
        # W = G^T P
        gQ = torch.empty_like(Q)
        G = Q[..., 4:4 + M]
        bgemm(P, gW.mT, out=gQ[..., 4:4 + M])             # gG = P gW^T  (only its lower triangle is ever used)
        bgemm(G, gW, D=gP, beta=1.0, triA=LOWER, out=gP)  # gP += G gW
        # KL: remaining columns of gQ, diagonal gradients of Lz and Lu
        gLL, gTT = torch.empty_like(LL), torch.empty_like(TT)
        gLu = torch.empty_like(Lu)
        check(lib().vargp_kl_t0_bwd(ptr(Q), ptr(Lz), ptr(Lu), ptr(ga), ptr(g_kl), ptr(gQ), ptr(gLL), ptr(gLu), S, C, M,
                                    st), 'vargp_kl_t0_bwd')
        # Q = T R and P = T K_uf : gT (lower), gR (summed over s), gK_uf
        gT = gTT[:SC].view(S, C, M, M)
        bgemm(gQ, R.mT, triC=LOWER, out=gT)                       # T is lower-triangular: so is its gradient
        bgemm(gP, Kuf.mT, D=gT, beta=1.0, triC=LOWER, out=gT)
        gTT[SC:].zero_()
        gKuf = bgemm(Tz.mT, gP, triA=UPPER)
        # fork: d K_uf -> (z, theta) on the side stream while the main stream runs the Cholesky backward chain
        main, side = torch.cuda.current_stream(), _side_stream(dev)
        gz, gtheta = torch.empty_like(z), torch.empty_like(theta)
        ws_uf = _rbf_ws(theta, z, x, True)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            _rbf_bwd(theta, z, x, True, Kuf, gKuf, gz, gtheta, accumulate=False, ws=ws_uf)
        gR = ops._reduce_to(bgemm(Tz.mT, gQ, triA=UPPER), R.shape)
        g_u_mean = gR[..., 0:1].contiguous()
        gLL[SC:].copy_(gR[..., 4:4 + M])
        gLu.add_(gR[..., 4 + M:])
        # both factorisations at once (vargp_chol_inv_bwd masks gL / gT to their lower triangles itself)
        gKS = torch.empty_like(KS)
        ws = workspace(lib().vargp_chol_workspace_bytes(SC + C, M, 1), dev)
        check(lib().vargp_chol_inv_bwd(ptr(LL), ptr(TT), ptr(gLL), ptr(gTT), ptr(gKS), SC + C, M, ptr(ws),
                                       ws.numel() * 4, st), 'vargp_chol_inv_bwd')
        # S_u = Lu Lu^T (gS_u is symmetric): gLu += 2 gS_u Lu
        bgemm(gKS[SC:], Lu, alpha=2.0, D=gLu, beta=1.0, triB=LOWER, out=gLu)
        g_vec = torch.empty_like(u_tril_vec)
        check(lib().vargp_vec2tril_bwd(ptr(u_tril_vec), ptr(gLu), ptr(g_vec), C, M, st), 'vargp_vec2tril_bwd')
        # kernel matrices -> z, theta (second call accumulates), plus the gamma^2 of the predictive variance
        main.wait_stream(side)                            # join, then accumulate the K_uu part on top
        del ws_uf
        _rbf_bwd(theta, z, None, False, KS, gKS, gz, gtheta, accumulate=True)
        check(lib().vargp_kdiag_bwd(ptr(theta), ptr(gkd), ptr(gtheta), S, C, D, st), 'vargp_kdiag_bwd')
        if _DEBUG_KEEP is not None:   # diagnostics: keep every backward intermediate alive for inspection
            _DEBUG_KEEP.update({k: v for k, v in locals().items() if isinstance(v, torch.Tensor)})
        return gtheta, gz, g_u_mean, g_vec, None, None, None


def _chol_big(KS, LL, TT, info, nb, M):
    ws = workspace(lib().vargp_chol_workspace_bytes(nb, M, 0), KS.device)
    return lib().vargp_chol_inv_fwd(ptr(KS), JITTER, ptr(LL), ptr(TT), None, ptr(info), nb, M, ptr(ws), ws.numel() * 4,
                                    stream_ptr())


def elbo_t0(theta, z, u_mean, u_tril_vec, x, y, eps_f):
    """-> (nll, kl_u) of VARGP.loss for a model with no previous tasks."""
    return _ElboT0.apply(theta, z, u_mean, u_tril_vec, x, y, eps_f)
