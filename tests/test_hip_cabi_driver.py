"""GPU: the C ABI used from plain C++ (no Python, no torch in the process): tests/native/t0_c_driver.cpp is compiled with
hipcc against include/vargp_hip.h, run on a problem written to a file, and must reproduce what the Python route computes
for the same parameters, minibatch and noise seed (the noise is drawn inside the library, so both see identical draws)."""
import os
import struct
import subprocess

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = 'cuda:0'


def test_c_driver_reproduces_python_route(tmp_path):
    from oracle import vargp_oracle as orc
    from vargp_amd.fused import T0Program
    S, F_, C, M, D, B = 2, 3, 3, 56, 40, 64
    seed = 0xC0FFEE1234
    params, prev, x, y, nz = orc.make_problem(S, F_, C, M, D, B, n_prev=0, seed=13, kind='mnist')
    order = ['log_mean', 'log_logvar', 'prior_log_mean', 'prior_log_logvar', 'z', 'u_mean', 'u_tril_vec']
    prob, res = tmp_path / 'problem.bin', tmp_path / 'result.bin'
    with open(prob, 'wb') as f:
        f.write(struct.pack('<6i', S, C, M, D, B, F_))
        f.write(struct.pack('<Q', seed))
        for k in order:
            f.write(params[k].float().contiguous().numpy().tobytes())
        f.write(x.float().contiguous().numpy().tobytes())
        f.write(y.long().contiguous().numpy().tobytes())

    exe = tmp_path / 't0_c_driver'
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O2', '-o', str(exe),
                           os.path.join(ROOT, 'tests', 'native', 't0_c_driver.cpp'),
                           '-L' + os.path.join(ROOT, 'vargp_amd'), '-lvargp_hip',
                           '-Wl,-rpath,' + os.path.join(ROOT, 'vargp_amd')])
    out = subprocess.run([str(exe), str(prob), str(res)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    raw = np.fromfile(res, dtype=np.float32)
    scal_c, bad = raw[:3], raw[3:4].view(np.int32)[0]
    assert bad == 0

    # the same step through the Python binding
    prog = T0Program(S, C, M, D, B, F_, DEV)
    counter = torch.zeros(1, dtype=torch.int32, device=DEV)
    prog.set_rng(seed, counter, 0)
    dev = [params[k].float().to(DEV).contiguous() for k in order]
    scal = prog.forward(dev[0], dev[1], dev[2], dev[3], dev[4], dev[5], dev[6], x.float().to(DEV).contiguous(),
                        y.long().to(DEV).contiguous(), None, None)
    grads = [torch.empty_like(dev[i]) for i in (0, 1, 4, 5, 6)]
    prog.backward(torch.tensor([2.0, 1.0, 7.0], device=DEV), *grads)
    torch.cuda.synchronize()
    np.testing.assert_allclose(scal_c, scal.cpu().numpy(), rtol=1e-6)
    off = 4
    for g in grads:
        n = g.numel()
        ref = g.cpu().numpy().ravel()
        got = raw[off:off + n]
        off += n
        # float atomics in a few reductions make the last bits run-dependent
        assert np.linalg.norm(got - ref) <= 1e-5 * max(np.linalg.norm(ref), 1e-30)
    assert off == raw.size
