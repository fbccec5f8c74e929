"""Drop-in alias of the reference's package name: `import var_gp.vargp`, `from var_gp.kernels import RBFKernel`, ... resolve
to the MI355X implementation in `vargp_amd` (same class / function names and argument meaning), so that
a caller written against uber-research/vargp (experiments/vargp.py:9-11) needs no import changes.  No code lives here."""
import sys

import vargp_amd
from vargp_amd import vargp, kernels, gp_utils, likelihoods, train_utils, datasets, vargp_retrain  # noqa: F401

for _name in ('vargp', 'kernels', 'gp_utils', 'likelihoods', 'train_utils', 'datasets', 'vargp_retrain'):
    sys.modules[f'{__name__}.{_name}'] = getattr(vargp_amd, _name)
