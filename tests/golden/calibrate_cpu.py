"""Speed calibration of the CPU baseline: the REFERENCE (imported read-only from /root/reference)
and the oracle's port (`oracle.elbo_step(full_gram=True)`, what bench.py's `cpu_baseline` times on the GPU box, where the
reference cannot travel) on the same Cfg2 inputs, same thread count, same step content (loss + combine + backward).
Run in the build container only:

    python tests/golden/calibrate_cpu.py            # writes tests/golden/cpu_calibration.json

`port_vs_reference` = port steps/s / reference steps/s; bench.py prints it inside `cpu_baseline` and also reports
`value_reference_equivalent` = port rate / port_vs_reference."""
import json
import os
import sys
import time
import warnings

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
warnings.filterwarnings('ignore')

from oracle import vargp_oracle as orc  # noqa: E402
import make_golden as mg  # noqa: E402  (imports the reference, provides the noise-injection harness)


def median(ts):
    ts = sorted(ts)
    return ts[len(ts) // 2]


def main(steps=24, threads=8):
    torch.set_num_threads(threads)
    S, F_, C, M, D, B = 3, 10, 10, 100, 784, 512
    beta, n_total = 10.0, 12000
    params, prev, x, y, noise = orc.make_problem(S, F_, C, M, D, B, seed=60, kind='gauss')
    gp = mg.build_ref(params, prev, S, F_)
    t_ref, t_port = [], []
    for i in range(steps + 2):          # interleaved, so that both see the same machine state
        with mg.injected(noise):
            t0 = time.perf_counter()
            gp.zero_grad()
            kl_h, kl_u, nll = gp.loss(x, y)
            (beta * kl_h + kl_u + (n_total / B) * nll).backward()
            t_ref.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        orc.elbo_step(params, prev, x, y, noise, beta=beta, n_total=n_total, full_gram=True)
        t_port.append(time.perf_counter() - t0)
    ref_s, port_s = median(t_ref[2:]), median(t_port[2:])
    ratio = median([a / b for a, b in zip(t_ref[2:], t_port[2:])])      # per interleaved pair: robust against load drift
    out = dict(workload='Cfg2 S3 F10 C10 M100 D784 B512, gauss data, loss + combine + backward (no optimiser)',
               threads=threads, steps=steps, reference_s_per_step=ref_s, port_s_per_step=port_s,
               reference_steps_per_s=1.0 / ref_s, port_steps_per_s=1.0 / port_s, port_vs_reference=ratio,
               torch=torch.__version__, host_cpus=os.cpu_count())
    print(json.dumps(out, indent=1))
    if '--no-write' not in sys.argv:
        with open(os.path.join(HERE, 'cpu_calibration.json'), 'w') as f:
            json.dump(out, f, indent=1)
            f.write('\n')


if __name__ == '__main__':
    main()
