"""vargp_amd — MI355X-native implementation of the VAR-GP ELBO hot path.

Same Python surface as the reference's `var_gp` package for this path (`vargp.VARGP`,
`kernels.RBFKernel`, `gp_utils.*`, `likelihoods.MulticlassSoftmax`), with every tensor operation of
`VARGP.loss` / `VARGP.predict` executed by hand-written HIP kernels for gfx950 behind the C ABI of
`include/vargp_hip.h`.  There is no CPU fallback: tensors must live on a ROCm device.
"""
from . import noise  # noqa: F401
from .ops import set_cholesky_error_mode, linalg_error_count, linalg_error_count_begin, reset_linalg_errors  # noqa: F401

__version__ = '0.1.0'
