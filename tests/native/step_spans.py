"""Time line of one Cfg2 step as the GPU sees it (vargp_prof_spans: wall-clock stamps written by the kernels themselves, 100 MHz):
first workgroup's start and last workgroup's end of each kernel, in steady state under graph replay, to set against the
dispatch-to-completion durations rocprofv3 reports.  GPU box only; bench.py prints the same table as `timeline`."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402
from vargp_amd import ops  # noqa: E402
from vargp_amd.train import ElboTrainer  # noqa: E402

dev = torch.device('cuda', 0)
ops.set_cholesky_error_mode('defer')
gp, x, y = bench.make_model(dev)
tr = ElboTrainer(gp, lr=bench.LR, beta=bench.BETA, n_total=bench.N_TOTAL)
for _ in range(5):
    tr.step(x, y)
tr.capture(x, y)
for _ in range(300):
    tr.step_graph()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200):
    tr.step_graph()
e1.record()
torch.cuda.synchronize()
period = e0.elapsed_time(e1) / 200 * 1e3
rows = bench.step_timeline(tr.step_graph, period, reps=int(os.environ.get('REPS', '30')))
print('step period %.1f us' % period)
print('%-18s %9s %9s %9s %9s' % ('kernel', 'start', 'span', 'gap<', 'slot'))
for r in rows:
    extra = ' '.join('%s=%.1f' % (k, v) for k, v in r.items() if k.endswith('_end_us') or k.endswith('_last_start_us'))
    print('%-18s %9.2f %9.2f %9.2f %9.2f  %s' % (r['kernel'], r['start_us'], r['span_us'], r['gap_before_us'], r['slot_us'], extra))
print('sum of spans %.1f us, of gaps %.1f us' % (sum(r['span_us'] for r in rows), sum(r['gap_before_us'] for r in rows)))
