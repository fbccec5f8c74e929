"""ctypes binding of libvargp_hip.so (the C ABI declared in include/vargp_hip.h).

There is deliberately NO fallback: if the shared library is missing or a tensor is not on a ROCm
device, the call raises.  (The CPU restatement under oracle/ is test infrastructure only.)
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('VARGP_HIP_LIB') or os.path.join(_HERE, 'libvargp_hip.so')   # env: tuning builds only


class GemmDesc(Structure):
    _fields_ = [
        ('M', c_int32), ('N', c_int32), ('K', c_int32),
        ('transA', c_int32), ('transB', c_int32),
        ('A', c_void_p), ('B', c_void_p), ('C', c_void_p), ('D', c_void_p),
        ('lda', c_int32), ('ldb', c_int32), ('ldc', c_int32), ('ldd', c_int32),
        ('nb', c_int32 * 3),
        ('sA', c_int64 * 3), ('sB', c_int64 * 3), ('sC', c_int64 * 3), ('sD', c_int64 * 3),
        ('alpha', c_float), ('beta', c_float),
        ('triA', c_int32), ('triB', c_int32), ('triC', c_int32),
    ]


class ElboT0Desc(Structure):
    _fields_ = [
        ('S', c_int32), ('C', c_int32), ('M', c_int32), ('D', c_int32), ('B', c_int32), ('F', c_int32),
        ('map_est', c_int32), ('jitter', c_float),
        ('log_mean', c_void_p), ('log_logvar', c_void_p), ('prior_log_mean', c_void_p), ('prior_log_logvar', c_void_p),
        ('z', c_void_p), ('u_mean', c_void_p), ('u_tril_vec', c_void_p),
        ('x', c_void_p), ('y', c_void_p),
        ('eps_theta', c_void_p), ('eps_f', c_void_p),
        ('scalars', c_void_p), ('info', c_void_p),
        ('ws', c_void_p), ('ws_bytes', c_size_t),
        ('bump', c_void_p),
        ('rng_seed', ctypes.c_uint64), ('rng_counter', c_void_p), ('rng_sample_offset', c_int32),
        ('defer_hyper', c_int32), ('defer_softmax', c_int32), ('ext_lik', c_int32),
        ('info_host', c_void_p), ('info_event', c_void_p),
    ]


class HyperGradDesc(Structure):
    _fields_ = [
        ('log_mean', c_void_p), ('log_logvar', c_void_p), ('prior_log_mean', c_void_p), ('prior_log_logvar', c_void_p),
        ('eps_theta', c_void_p), ('gtheta', c_void_p), ('g2', c_void_p), ('gkd', c_void_p), ('seeds', c_void_p),
        ('S', c_int32), ('C', c_int32), ('D1', c_int32), ('map_est', c_int32),
    ]


class ElboTnDesc(Structure):
    _fields_ = [
        ('S', c_int32), ('C', c_int32), ('M', c_int32), ('D', c_int32), ('B', c_int32), ('F', c_int32), ('nblk', c_int32),
        ('map_est', c_int32), ('jitter', c_float),
        ('log_mean', c_void_p), ('log_logvar', c_void_p), ('prior_log_mean', c_void_p), ('prior_log_logvar', c_void_p),
        ('z', c_void_p), ('u_mean', c_void_p), ('u_tril_vec', c_void_p),
        ('z_all', c_void_p), ('rk_all', c_void_p),
        ('x', c_void_p), ('y', c_void_p),
        ('eps_theta', c_void_p), ('eps_f', c_void_p),
        ('scalars', c_void_p), ('info', c_void_p),
        ('ws', c_void_p), ('ws_bytes', c_size_t),
        ('bump', c_void_p),
        ('rng_seed', ctypes.c_uint64), ('rng_counter', c_void_p), ('rng_sample_offset', c_int32),
        ('forward_only', c_int32), ('defer_hyper', c_int32), ('ext_lik', c_int32),
        ('eps_u', c_void_p), ('n_v', c_int32), ('no_var_mean', c_int32),
        ('info_host', c_void_p), ('info_event', c_void_p),
    ]


_P = c_void_p
_SIGNATURES = {
    'vargp_version': (c_int, []),
    'vargp_last_error': (c_char_p, []),
    'vargp_bgemm': (c_int, [POINTER(GemmDesc), _P]),
    'vargp_sum_outer': (c_int, [_P, _P, c_int64, c_int64, _P]),
    'vargp_rbf_workspace_bytes': (c_size_t, [c_int] * 6),
    'vargp_rbf_gram_fwd': (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_size_t, _P]),
    'vargp_rbf_gram_bwd': (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P,
                                   c_size_t, _P]),
    'vargp_chol_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'vargp_chol_inv_fwd': (c_int, [_P, c_float, _P, _P, _P, _P, c_int, c_int, _P, c_size_t, _P]),
    'vargp_chol_inv_bwd': (c_int, [_P, _P, _P, _P, _P, c_int, c_int, _P, c_size_t, _P]),
    'vargp_trsm_lower_fwd': (c_int, [_P, _P, _P, c_int, c_int, c_int, _P]),
    'vargp_trsm_lower_bwd': (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, _P, c_size_t, _P]),
    'vargp_vec2tril_fwd': (c_int, [_P, _P, c_int, c_int, _P]),
    'vargp_vec2tril_bwd': (c_int, [_P, _P, _P, c_int, c_int, _P]),
    'vargp_mat2trilvec': (c_int, [_P, _P, c_int, c_int, _P]),
    'vargp_predictive_diag_fwd': (c_int, [_P, _P, _P, c_int64, c_int64, _P, _P, _P, c_int, c_int, c_int, _P]),
    'vargp_predictive_diag_bwd': (c_int, [_P, _P, _P, c_int64, c_int64] + [_P] * 6 + [c_int, c_int, c_int, _P]),
    'vargp_mvn_kl_fwd': (c_int, [_P, _P, _P, _P, _P, c_int, c_int, _P]),
    'vargp_mvn_kl_bwd': (c_int, [_P, _P, _P, _P, _P, c_int, c_int, _P]),
    'vargp_logdet_tril_fwd': (c_int, [_P, _P, c_int, c_int, _P]),
    'vargp_logdet_tril_bwd': (c_int, [_P, _P, _P, c_int, c_int, _P]),
    'vargp_softmax_nll_fwd': (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    'vargp_softmax_nll_bwd': (c_int, [_P] * 7 + [c_int, c_int, c_int, c_int, _P]),
    'vargp_softmax_predict': (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    'vargp_yogi_step_multi': (c_int, [c_int, _P, _P, _P, _P, _P] + [c_float] * 4 + [_P, c_int, _P]),
    'vargp_yogi_step_multi_hyper': (c_int, [c_int, _P, _P, _P, _P, _P] + [c_float] * 4 + [_P, c_int, POINTER(HyperGradDesc), c_int,
                                            c_int, _P]),
    'vargp_elbo_t0_hyper_desc': (c_int, [POINTER(ElboT0Desc), _P, POINTER(HyperGradDesc)]),
    'vargp_elbo_tn_hyper_desc': (c_int, [POINTER(ElboTnDesc), _P, POINTER(HyperGradDesc)]),
    'vargp_hyper_sample_fwd': (c_int, [_P, _P, _P, _P, c_int, c_int, _P]),
    'vargp_hyper_sample_bwd': (c_int, [_P, _P, _P, _P, _P, c_int, c_int, _P]),
    'vargp_hyper_kl_fwd': (c_int, [_P, _P, _P, _P, _P, c_int, _P]),
    'vargp_hyper_kl_bwd': (c_int, [_P] * 7 + [c_int, _P]),
    'vargp_elbo_t0_workspace_bytes': (c_size_t, [c_int] * 6),
    'vargp_elbo_t0_fwd': (c_int, [POINTER(ElboT0Desc), _P]),
    'vargp_elbo_t0_lik_buffers': (c_int, [POINTER(ElboT0Desc)] + [POINTER(c_void_p)] * 4),
    'vargp_elbo_tn_lik_buffers': (c_int, [POINTER(ElboTnDesc)] + [POINTER(c_void_p)] * 4),
    'vargp_elbo_t0_bwd': (c_int, [POINTER(ElboT0Desc)] + [_P] * 7),
    'vargp_bias_act_fwd': (c_int, [_P, _P, _P, c_int64, c_int, c_int, _P]),
    'vargp_bias_act_bwd': (c_int, [_P, _P, _P, _P, c_int64, c_int, c_int, _P]),
    'vargp_elbo_tn_workspace_bytes': (c_size_t, [c_int] * 7),
    'vargp_elbo_tn_workspace_bytes_fwd': (c_size_t, [c_int] * 7),
    'vargp_elbo_tn_fwd': (c_int, [POINTER(ElboTnDesc), _P]),
    'vargp_elbo_tn_bwd': (c_int, [POINTER(ElboTnDesc)] + [_P] * 7),
    'vargp_elbo_tn_begin': (c_int, [POINTER(ElboTnDesc), _P]),
    'vargp_elbo_tn_tile': (c_int, [POINTER(ElboTnDesc), _P, _P, _P, _P, c_int, _P]),
    'vargp_elbo_tn_end': (c_int, [POINTER(ElboTnDesc)] + [_P] * 7),
    'vargp_elbo_tn_moments': (c_int, [POINTER(ElboTnDesc), POINTER(c_void_p), POINTER(c_void_p)]),
    'vargp_prof_enable': (c_int, [c_int]),
    'vargp_prof_read': (c_int, [c_char_p, POINTER(ctypes.c_double), POINTER(c_int64)]),
    'vargp_prof_remember': (c_int, [c_int]),
    'vargp_prof_spans': (c_int, [c_int, POINTER(ctypes.c_uint64)]),
    'vargp_tune_gemm_tile': (c_int, [c_int]),
    'vargp_prof_replay': (c_int, [c_char_p, c_int, POINTER(ctypes.c_double), _P]),
    'vargp_yogi_step': (c_int, [_P, _P, _P, _P, c_int64] + [c_float] * 6 + [_P, _P]),
    'vargp_gather_minibatch': (c_int, [_P, _P, _P, _P, _P, c_int64, c_int, c_int, _P, _P, _P]),
}
EXPORTS = sorted(_SIGNATURES)

_lib = None


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP library has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f'{LIB_PATH} not found: build it with `make -C vargp_amd/csrc` (or __graft_entry__.build()). '
                'vargp_amd has no CPU fallback.')
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


class VargpHipError(RuntimeError):
    pass


def check(rc, what):
    if rc != 0:
        raise VargpHipError(f'{what} failed (code {rc}): {lib().vargp_last_error().decode()}')


def require_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise VargpHipError(
                'vargp_amd ops need tensors on a ROCm device (cuda:N); there is no CPU path in the product '
                '(the CPU restatement lives under oracle/ and is for tests only).')


def stream_ptr():
    # (torch.cuda.current_stream() builds a Stream object: ~20 us per call, three calls per training step on the eager route)
    raw = getattr(torch._C, '_cuda_getCurrentRawStream', None)
    if raw is not None:
        return c_void_p(raw(torch.cuda.current_device()))
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


def workspace(nbytes, device):
    return torch.empty((int(nbytes) + 3) // 4, dtype=torch.float32, device=device)


_scratch_pool = {}


def scratch(nbytes, device):
    """Pooled per-op scratch: ONE growing buffer per (device, stream).  Every C-ABI op uses its workspace only inside the
    call, and calls on a stream execute in order, so consecutive ops can share the buffer (no allocation per op; stable
    addresses under hipGraph capture once it has reached its final size)."""
    dev = torch.device(device)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(dev).cuda_stream)
    need = (int(nbytes) + 3) // 4
    if torch.cuda.is_current_stream_capturing():
        # a captured graph keeps raw pointers: give it memory of its own (the graph's private pool keeps it alive), never
        # the shared buffer, which may be re-allocated when it grows later
        return torch.empty(need, dtype=torch.float32, device=device)
    buf = _scratch_pool.get(key)
    if buf is None or buf.numel() < need:
        buf = _scratch_pool[key] = torch.empty(max(need, 2 * (buf.numel() if buf is not None else 0)),
                                               dtype=torch.float32, device=device)
    return buf


def prof_enable(on=True):
    lib().vargp_prof_enable(int(on))


def prof_read(tag):
    """-> (total_ms, launches) of the launches tagged `tag` since the last read."""
    ms, n = ctypes.c_double(0.0), c_int64(0)
    lib().vargp_prof_read(tag.encode(), ctypes.byref(ms), ctypes.byref(n))
    return ms.value, n.value


SPAN_SLOTS = {0: 't0_pro_kuu', 1: 'chol_rbf_gemm', 2: 'gemm_kernel', 3: 't0_fwd_fused', 4: 't0_bwd_mid', 5: 't0_bwdmat_gemm',
              6: 't0_puu_final', 7: 'yogi_multi', 8: 't0_bwdmat_gemm:chains_end', 9: 'chol_rbf_gemm:chains_end',
              10: 't0_pro_kuu:gram_end', 11: 't0_pro_kuu:norms_end'}


def prof_spans(mode):
    """Step time line (vargp_hip.h: vargp_prof_spans).  mode 1: clear + on, 2: off, 0: -> {slot name: (start_us, end_us)} of the
    slots stamped since the last clear (device wall clock, 10 ns ticks, as microseconds from an arbitrary origin)."""
    if mode:
        check(lib().vargp_prof_spans(int(mode), None), 'vargp_prof_spans')
        return None
    out = (ctypes.c_uint64 * 24)()
    check(lib().vargp_prof_spans(0, out), 'vargp_prof_spans')
    return {SPAN_SLOTS.get(i, str(i)): (out[2 * i] / 100.0, out[2 * i + 1] / 100.0) for i in range(12) if out[2 * i + 1]}


def prof_clocks():
    """{slot name: GHz} -- the shader clock the chip held while workgroup 0 of each slot's last stamped launch ran (vargp_prof_spans
    mode 3: s_memtime ticks over 100 MHz wall-clock ticks)."""
    out = (ctypes.c_uint64 * 24)()
    check(lib().vargp_prof_spans(3, out), 'vargp_prof_spans')
    return {SPAN_SLOTS.get(i, str(i)): 0.1 * out[2 * i] / out[2 * i + 1] for i in range(12) if out[2 * i + 1]}


def prof_remember(on=True):
    """While on, tagged launches keep a copy of their arguments for prof_replay (see vargp_hip.h)."""
    lib().vargp_prof_remember(int(on))


def prof_replay(tag, iters=50):
    """Average time (us) of the most recent launch tagged `tag`, re-launched back to back (see vargp_hip.h)."""
    us = ctypes.c_double(0.0)
    check(lib().vargp_prof_replay(tag.encode(), int(iters), ctypes.byref(us), stream_ptr()), 'vargp_prof_replay')
    return us.value
