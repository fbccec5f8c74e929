"""One ELBO training step as the reference's training loop performs it
(experiments/vargp.py:29-37): zero_grad, loss, combine, backward, optimiser step — plus the
sample-parallel multi-GPU exchange (one all-reduce of [grads | kl_u | nll]).

Sample-parallel sharding: every rank holds all parameters and the same minibatch and
evaluates its `gp.n_v = S_r` of the S = sum_r S_r hyper-parameter samples; the per-rank partial means combine
linearly with the weights w_r = S_r / S, so   total = beta*kl_hypers + sum_r w_r kl_u_r + (N/B) sum_r w_r nll_r.
The S_r need not be equal (S not divisible by the number of ranks: `split_samples` hands the remainder to the first
ranks, at most one sample of imbalance), which is how a fixed sample count -- BASELINE config 4: 64 samples over 8 GPUs --
is divided without any exchange inside the step.
Each rank back-propagates its share times w_r and ONE exchange (sum) over a flat
fp32 buffer [grad(z) | grad(u_mean) | grad(u_tril_vec) | grad(log_mean) | grad(log_logvar) | kl_u | nll]
yields identical gradients on every rank.  Gradients are views into that flat buffer, so there is no
pack/unpack copy.  The exchange is either one all-reduce (`comm='allreduce'`) or a reduce-scatter + all-gather of
the same buffer (`comm='rsag'`); both are RCCL collectives on the GPU (gloo in the CPU tests).
"""
import os

import torch
import torch.distributed as dist

from . import noise
from .optim import Yogi


def split_samples(total, world):
    """Per-rank hyper-sample counts for `total` samples over `world` ranks: total // world each, the first total % world
    ranks one more."""
    return [total // world + (1 if r < total % world else 0) for r in range(world)]


def split_pairs(S, C, world):
    """One rectangle (s0, s1, c0, c1) of the S x C grid of (hyper-sample, class) problems per rank -- everything before the
    softmax is independent per (s, c) (var_gp/kernels.py:38-56, gp_utils.py:174-186).  world <= S: whole
    hyper-samples (all classes: no exchange before the likelihood), the remainder to the first ranks.  world > S: the ranks are
    dealt to the samples (world // S each, the first world % S samples one more) and every sample's classes are cut into
    contiguous ranges over its ranks -- a rank then holds ONE sample and some classes (BASELINE config 2, S = 3, C = 10 on 8
    GPUs: 4 3 3 | 4 3 3 | 5 5 pairs), and the predictive moments are all-gathered before the likelihood."""
    assert world <= S * C, f'{world} ranks for {S} x {C} (sample, class) problems'
    if world <= S:
        out, s0 = [], 0
        for n in split_samples(S, world):
            out.append((s0, s0 + n, 0, C))
            s0 += n
        return out
    out = []
    for s, nr in enumerate(split_samples(world, S)):
        assert nr <= C
        c0 = 0
        for n in split_samples(C, nr):
            out.append((s, s + 1, c0, c0 + n))
            c0 += n
    return out


class ElboTrainer:
    """`gp` is a vargp_amd VARGP module.  For tests of the exchange logic on CPU (gloo) the model can be
    replaced by any `loss_fn(x, y) -> (kl_hypers, kl_u, nll)` over an explicit `params` list."""

    def __init__(self, gp=None, lr=1e-2, beta=1.0, n_total=None, group=None, noise_seed=1234, optimizer=None,
                 params=None, loss_fn=None, native_noise=True, sample_counts=None, force_exchange=False, comm='allreduce',
                 shards=None, pair_fns=None, grid=None, pair_dims=None):
        self.gp = gp
        self.loss_fn = loss_fn if loss_fn is not None else (gp.loss if gp is not None else None)
        self.beta = float(beta)
        self.n_total = n_total
        self.group = group
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        # force_exchange: take the multi-rank path (all-reduce of the flat buffer, two-graph capture) with one rank too --
        # a smoke test of that path where only one GPU is available (bench.py, VARGP_BENCH_FORCE_DIST=1)
        self.multi = self.world > 1 or (bool(force_exchange) and dist.is_available() and dist.is_initialized())
        self.params = list(params) if params is not None else [p for p in gp.parameters() if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        # comm: how the flat buffer is summed over the ranks -- 'allreduce' (one all-reduce) or 'rsag' (reduce-scatter of
        # the buffer cut into `world` shards + all-gather of the reduced shards: on 8 fully connected GPUs every shard
        # travels over its own xGMI link).  The buffer is padded to a multiple of the world size for that.
        assert comm in ('allreduce', 'rsag'), comm
        self.comm = comm
        pad = -(n + 2) % max(self.world, 1)
        self._flat_store = torch.zeros(n + 2 + pad, dtype=torch.float32, device=dev)
        self.flat = self._flat_store[:n + 2]
        self._shard = (torch.zeros(self._flat_store.numel() // self.world, dtype=torch.float32, device=dev)
                       if comm == 'rsag' else None)
        self.comm_events = None            # list -> exchange() appends one (start, end) event pair per call (bench.py)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        self.scalars = self.flat[n:]
        self.optim = optimizer if optimizer is not None else Yogi(self.params, lr=lr)
        if callable(self.optim) and not hasattr(self.optim, 'step'):
            self.optim = self.optim(self.params)
        # hyper-samples per rank (default: gp.n_v on every rank) -> this rank's weight and first global sample index
        self.sample_counts = None if sample_counts is None else [int(c) for c in sample_counts]
        if self.sample_counts is not None:
            assert len(self.sample_counts) == self.world and min(self.sample_counts) > 0, self.sample_counts
            assert gp is None or not hasattr(gp, 'n_v') or gp.kernel.map_est or gp.n_v == self.sample_counts[self.rank]
            self.weight = self.sample_counts[self.rank] / float(sum(self.sample_counts))
            self.sample_offset = sum(self.sample_counts[:self.rank])
        else:
            self.weight = 1.0 / self.world
            self.sample_offset = None          # rank * S, S known at the first step
        # shards: one rectangle (s0, s1, c0, c1) of the (hyper-sample, class) grid per rank (split_pairs).  Rectangles that cut
        # the classes put the step on the class-sharded route: moments of the rank's pairs -> all-gather of mu, var (S, C, B)
        # -> the likelihood of ALL pairs on every rank (S F C B exps: nothing) -> backward of the rank's pairs -> the same
        # exchange of the flat buffer.  grid = (S, C) of the whole problem; pair_fns = (moments_fn, lik_fn) for models without
        # a native program (the CPU tests drive the oracle through them).
        self.shards = None if shards is None else [tuple(int(v) for v in r) for r in shards]
        self.pair_fns = pair_fns
        self._pair_dims = pair_dims            # (D + 1, F) for the noise shapes when there is no model object (pair_fns)
        self.class_split = False
        if self.shards is not None:
            assert len(self.shards) == self.world and self.multi
            self.grid = tuple(grid) if grid is not None else (gp.n_v, gp.z.size(0))
            St, Ct = self.grid
            assert sum((r[1] - r[0]) * (r[3] - r[2]) for r in self.shards) == St * Ct, 'shards must tile the (S, C) grid'
            self.class_split = any(r[3] - r[2] < Ct for r in self.shards)
            self.rect = self.shards[self.rank]
            s0, s1, c0, c1 = self.rect
            if not self.class_split:      # whole samples per rank: the sample-parallel route with these counts
                self.sample_counts = [r[1] - r[0] for r in self.shards]
                self.weight = (s1 - s0) / float(St)
                self.sample_offset = s0
            else:
                self.w_kl = (s1 - s0) / float(St)                       # kl_u = sum_r w_kl_r * (mean over the rank's samples)
                self.w_h = (s1 - s0) * (c1 - c0) / float(St * Ct)       # replicated terms (kl_hypers, nll) counted once in the sum
                self.max_pairs = max((r[1] - r[0]) * (r[3] - r[2]) for r in self.shards)
        if self.multi and self.class_split:
            # class-sharded: every rank draws the WHOLE noise tensors (same seed, same stream) and uses its rows of eps_theta
            noise.set_shard(0, 1, noise_seed, dev, None)
        elif self.multi:
            noise.set_shard(self.rank, self.world, noise_seed, dev, self.sample_counts)

        self.graph = None
        self.graph_opt = None
        self.graph_mid = None
        self._captured = {}
        # models on a native program (fused.T0Program: first task; fused.TnProgram: later tasks, ep_var_mean=True) drive it
        # directly: no autograd graph, gradients written straight into the optimiser's buffers
        is_model = gp is not None and loss_fn is None and params is None and hasattr(gp, 'draw_t0_noise')
        # which of the two programs a first-task model runs on depends on the minibatch size (VARGP.first_task_as_block):
        # decided per batch shape in _t0_fwd_bwd, exactly as VARGP.loss routes -- `_tn` describes the most recent step
        self._tn = bool(is_model and gp._use_block_program())                                             # csrc/elbo_tn.hip
        self._t0 = bool(is_model and not gp.prev_params and gp.fused_first_task
                        and type(gp.kernel).__name__ == 'RBFKernel') or self._tn                           # csrc/elbo_t0.hip
        # one program (descriptor + workspace) PER SHAPE, never freed: a captured hipGraph holds raw pointers into the
        # program it was captured with, and the ragged last minibatch of an epoch runs eagerly through another shape
        self._progs, self._prog, self._seeds, self._own_grads, self._scratch_grads = {}, None, {}, None, {}
        # with our Yogi (one parameter group) the program's first kernel also advances the optimiser's step count
        self._bump = None
        # native noise: the program draws eps_theta / eps_f itself (Philox keyed by noise_seed, device-side step
        # counter): no randn launches, and ranks see slices of one global draw by construction
        self.native_noise = bool(native_noise) and self._t0 and not self.class_split
        # ep_var_mean = False models with earlier tasks draw eps_u -- and with it eps_theta / eps_f -- from the torch generator
        # (noise.draw) on every step, native noise or not: the captures below must register that generator
        self._draws_u = bool(is_model and gp.prev_params and gp.var_mean_mask != 1.0)
        if self.class_split and self._draws_u:
            raise NotImplementedError('class-sharded steps (more ranks than hyper-samples) run ep_var_mean=True models only: the '
                                      "ablation's KL needs samples of u_<t for every class of a hyper-sample on one rank; use "
                                      'whole-sample shards (world <= n_var_samples)')
        self.noise_seed = int(noise_seed)
        self._rng_counter = torch.zeros(1, dtype=torch.int32, device=dev) if self.native_noise else None
        if self._t0 and isinstance(self.optim, Yogi) and len(self.optim.param_groups) == 1:
            self.optim.external_step = True
            self._bump = self.optim.step_counter(dev)

    # -- hipGraph capture of the step --------------------------------------------------------------
    def capture(self, x, y, warmup=3):
        """Capture the step into hipGraphs (torch.cuda.CUDAGraph).  Needs the 'defer' Cholesky error mode (no
        host sync inside the step).  One GPU: a single graph (zero-grad .. optimiser).  Sample-parallel: two
        graphs around the RCCL all-reduce, which stays an ordinary (un-captured) call:
            graph A = zero flat buffer, loss, backward, scalars   ->   all_reduce(flat)   ->   graph B = optimiser.
        Afterwards use step_graph()."""
        from . import ops
        assert ops._chol_mode == 'defer', "set_cholesky_error_mode('defer') before capturing"
        self._sx, self._sy = x.clone(), y.clone()
        self.graph_opt = None
        # the warm-up steps are real steps: snapshot parameters, optimiser state and the noise stream, restore afterwards,
        # so that the graph-mode trajectory equals the eager one (no extra updates on the first minibatch)
        snap = self._snapshot_state()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self.step(self._sx, self._sy)
        torch.cuda.current_stream().wait_stream(side)
        self._restore_state(snap)
        self.graph = torch.cuda.CUDAGraph()
        if noise._shard is not None and not (self._t0 and self.native_noise and not self._draws_u):
            # the composed (t > 0) path draws its noise from the shared torch generator inside the captured region;
            # the first-task program has its own counter-based generator and needs no generator bookkeeping per replay
            self.graph.register_generator_state(noise._shard[2])
        if not self.multi:
            with torch.cuda.graph(self.graph):
                self._sout = self.step(self._sx, self._sy)
        else:
            self.graph_mid = None
            if self.class_split:
                # three graphs: moments | all-gather | likelihood + backward | exchange | optimiser
                with torch.cuda.graph(self.graph):
                    self._pair_part1(self._sx, self._sy)
                self.graph_mid = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_mid, pool=self.graph.pool()):
                    self._sout = self._pair_part2(self._sx, self._sy)
            else:
                with torch.cuda.graph(self.graph):
                    self._sout = self._local_part(self._sx, self._sy)
            self.graph_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_opt, pool=self.graph.pool()):
                self.optim.step()
        # one captured step per minibatch size (the ragged last batch of an epoch gets its own: experiments/vargp.py); the
        # attributes graph / _sx / _sy / _sout always describe the most recent capture
        self._captured[int(x.size(0))] = (self.graph, self.graph_opt, self._sx, self._sy, self._sout, getattr(self, 'graph_mid', None))
        return self

    def capture_unrolled(self, x, y, k):
        """K consecutive steps in ONE hipGraph (one GPU only; after `capture`): a graph launch costs a few microseconds more than
        a kernel boundary inside a graph, so K steps per launch shave that off K - 1 of every K steps.  The K steps read K
        static minibatch slots (`step_graph_k(xs, ys)` with xs (K, B, D), ys (K, B); without arguments the slots keep what they
        hold -- initially K copies of x, y); noise and optimiser step counts advance on the device as in the one-step graph."""
        assert self.graph is not None and not self.multi and k >= 2 and x.size(0) == self._sx.size(0)
        self._k = int(k)
        self._sxk = x.unsqueeze(0).repeat(k, *([1] * x.dim())).contiguous()
        self._syk = y.unsqueeze(0).repeat(k, *([1] * y.dim())).contiguous()
        snap = self._snapshot_state()
        self.graph_k = torch.cuda.CUDAGraph()
        if noise._shard is not None and not (self._t0 and self.native_noise and not self._draws_u):
            self.graph_k.register_generator_state(noise._shard[2])
        with torch.cuda.graph(self.graph_k, pool=self.graph.pool()):
            for i in range(k):
                self._soutk = self.step(self._sxk[i], self._syk[i])
        self._restore_state(snap)            # (capture itself runs nothing, but keeps the eager bookkeeping honest)
        return self

    def step_graph_k(self, xs=None, ys=None):
        """Replay the K-step graph -> the last step's (kl_hypers, kl_u, nll)."""
        if xs is not None:
            self._sxk.copy_(xs, non_blocking=True)
            self._syk.copy_(ys, non_blocking=True)
        self.graph_k.replay()
        return self._soutk

    # -- device-resident epochs: the minibatch gather inside the graph, K steps per launch ------------------------------
    def capture_epoch(self, data, targets, k=32):
        """After `capture(x, y)` for the full batch size (one GPU, first-task / block program with our Yogi: the optimiser's step
        count lives on the device and is advanced by the program's first kernel): two more graphs whose steps START with the
        gather of their own minibatch out of the device-resident training set -- `vargp_gather_minibatch`: batch index =
        (device step count) - (its value at the start of the epoch), rows = `perm[index * B + r]` -- a graph of K = min(k, full
        batches per epoch) steps and one of the (full batches mod K) steps left over.  `run_epoch(perm)` then trains a whole
        epoch's full batches with ceil(full batches / K) graph launches and no other launch at all (the per-step form: two
        index_select launches + one graph launch per step).  -> self, or None when the
        trainer cannot (no device step counter, several ranks): the caller keeps `step_graph_gather`."""
        if self.graph is None or self.multi or self._bump is None or not self._t0:
            return None
        from ._lib import check, lib, ptr, stream_ptr
        B, D = self._sx.shape
        n = int(targets.size(0))
        assert data.is_contiguous() and targets.is_contiguous() and data.shape == (n, D) and targets.dtype == torch.int64
        nfull = n // int(B)
        k = max(1, min(int(k), nfull))
        self._ep = dict(data=data, targets=targets, n=n, B=int(B), k=k, r=nfull % k,
                        perm=torch.arange(n, dtype=torch.int64, device=data.device),
                        base=torch.zeros(1, dtype=torch.float32, device=data.device))
        ep = self._ep

        def gather():
            check(lib().vargp_gather_minibatch(ptr(data), ptr(targets), ptr(ep['perm']), ptr(self._bump), ptr(ep['base']), n, int(B),
                                               int(D), ptr(self._sx), ptr(self._sy), stream_ptr()), 'vargp_gather_minibatch')

        snap = self._snapshot_state()
        for name, steps in (('gk', k), ('gr', ep['r'])):
            if steps == 0:
                continue
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=self.graph.pool()):
                for _ in range(steps):
                    gather()
                    ep['out'] = self.step(self._sx, self._sy)
            ep[name] = g
        self._restore_state(snap)
        return self

    def run_epoch(self, perm):
        """One epoch's FULL minibatches of the permutation `perm` (device int64, a permutation of range(n)) -> (last step's
        (kl_hypers, kl_u, nll), number of steps run).  The ragged rest, perm[steps * B:], is the caller's (`step_graph_gather`)."""
        ep = self._ep
        ep['perm'].copy_(perm, non_blocking=True)
        ep['base'].copy_(self._bump, non_blocking=True)
        nfull = ep['n'] // ep['B']
        for _ in range(nfull // ep['k']):
            ep['gk'].replay()
        if ep['r']:
            ep['gr'].replay()
        return ep['out'], nfull

    def captured_sizes(self):
        return sorted(self._captured)

    def _select_capture(self, nb):
        if self._sx.size(0) != nb:
            self.graph, self.graph_opt, self._sx, self._sy, self._sout, self.graph_mid = self._captured[int(nb)]
        if self.class_split:
            # the un-captured all-gather between the graphs works on the pair buffers of THIS minibatch size (an eager step or a
            # capture of another size rebinds them: _pair_setup)
            self._pair_setup(self._sx)

    def _snapshot_state(self):
        snap = dict(params=[p.detach().clone() for p in self.params], rng=None, gen=None, opt=[])
        if self._rng_counter is not None:
            snap['rng'] = self._rng_counter.clone()
        if noise._shard is not None:
            snap['gen'] = noise._shard[2].get_state()
        elif self._rng_counter is None and self.params[0].is_cuda:
            # composed path on one GPU (DeepRBFKernel, the ep_var_mean=False ablation, a caller's loss_fn): noise.draw uses the
            # device's global generator, which the warm-up steps advance
            snap['cuda_rng'] = torch.cuda.get_rng_state(self.params[0].device)
        for grp in self.optim.param_groups:
            snap['opt'].append({k: v.clone() for k, v in grp.items() if torch.is_tensor(v)})
        snap['state'] = {id(p): {k: (v.clone() if torch.is_tensor(v) else v) for k, v in st.items()}
                         for p, st in self.optim.state.items()}
        return snap

    def _restore_state(self, snap):
        with torch.no_grad():
            for p, v in zip(self.params, snap['params']):
                p.copy_(v)
            if snap['rng'] is not None:
                self._rng_counter.copy_(snap['rng'])
            if snap['gen'] is not None:
                noise._shard[2].set_state(snap['gen'])
            if snap.get('cuda_rng') is not None:
                torch.cuda.set_rng_state(snap['cuda_rng'], self.params[0].device)
            for grp, saved in zip(self.optim.param_groups, snap['opt']):
                for k, v in saved.items():
                    grp[k].copy_(v)                                   # in place: the kernels hold these pointers
                if 'step' in grp and torch.is_tensor(grp['step']) and 'step' not in saved:
                    grp['step'].zero_()                               # created by the warm-up
            for p, st in self.optim.state.items():
                old = snap['state'].get(id(p))
                for k, v in st.items():
                    if not torch.is_tensor(v):
                        continue
                    if old is not None and k in old:
                        v.copy_(old[k])
                    else:                                             # moment buffers created by the warm-up
                        v.fill_(self.optim.defaults.get('initial_accumulator', 0.0))
                if old is not None:
                    for k, v in old.items():
                        if not torch.is_tensor(v):
                            st[k] = v

    def exchange(self):
        """Sum the flat buffer [grads | kl_u | nll] over the ranks (in place, identical result on every rank)."""
        ev = None
        if self.comm_events is not None and self.flat.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        if self.comm == 'rsag':
            dist.reduce_scatter_tensor(self._shard, self._flat_store, op=dist.ReduceOp.SUM, group=self.group)
            dist.all_gather_into_tensor(self._flat_store, self._shard, group=self.group)
        else:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        if ev is not None:
            ev[1].record()
            self.comm_events.append(ev)

    def step_graph(self, x=None, y=None):
        """Replay the captured step (optionally on a new minibatch of the captured shape)."""
        self._select_capture(x.size(0) if x is not None else self._sx.size(0))
        if x is not None:
            self._sx.copy_(x, non_blocking=True)
            self._sy.copy_(y, non_blocking=True)
        self.graph.replay()
        if self.multi:
            if self.class_split:
                self.gather_moments()
                self.graph_mid.replay()
            self.exchange()
            self.graph_opt.replay()
        return self._sout

    def step_graph_gather(self, data, targets, idx):
        """Replay the captured step on the minibatch data[idx], targets[idx] (device-resident data set, idx a device index
        tensor of the captured batch size): gathered straight into the graph's static inputs, no host copy."""
        self._select_capture(idx.numel())
        torch.index_select(data, 0, idx, out=self._sx)
        torch.index_select(targets, 0, idx, out=self._sy)
        return self.step_graph()

    def step(self, x, y):
        """-> (kl_hypers, kl_u, nll) as 0-dim device tensors (global values on every rank)."""
        scale = (self.n_total if self.n_total is not None else x.size(0)) / x.size(0)
        if self._t0 and not self.multi:
            if self._own_grads is None:
                self._own_grads = [torch.empty_like(p) for p in self.params]
            for p, g in zip(self.params, self._own_grads):
                p.grad = g
            # one GPU, our Yogi over one parameter group: the program's last kernel (theta-gradient -> log_mean / log_logvar)
            # is finished inside the optimiser's launch
            defer = self._defer_hyper()
            scal = self._t0_fwd_bwd(x, y, scale, 1.0, defer_hyper=defer)
            if defer:
                k = self.gp.kernel
                self.optim.step(hyper=(self._prog.hyper_desc(), k.log_mean, k.log_logvar))
            else:
                self.optim.step()
            return scal[0], scal[1], scal[2]
        if not self.multi:
            # single GPU: let autograd hand the gradients over (no zero-fill, no accumulate kernels)
            for p in self.params:
                p.grad = None
            kl_h, kl_u, nll = self.loss_fn(x, y)
            (self.beta * kl_h + kl_u + scale * nll).backward()
            self.optim.step()
            return kl_h.detach(), kl_u.detach(), nll.detach()
        out = self._local_part(x, y)
        self.exchange()
        self.optim.step()
        return out

    def _defer_hyper(self):
        """Finish the hyper-parameter backward inside the optimiser's launch?  Only when that launch updates BOTH
        variational hyper-parameter tensors (vargp_yogi_step_multi_hyper rejects a missing log_logvar unless the kernel is
        a MAP estimate, where log_logvar receives no gradient at all).  NOTE: with deferral the .grad of log_mean /
        log_logvar is only valid after optim.step()."""
        k = self.gp.kernel
        has = lambda t: any(p is t for p in self.params)
        return (isinstance(self.optim, Yogi) and len(self.optim.param_groups) == 1 and len(self.params) <= 8
                and has(k.log_mean) and (bool(k.map_est) or has(k.log_logvar))
                and os.environ.get('VARGP_DEFER_HYPER', '1') != '0')

    def _t0_fwd_bwd(self, x, y, scale, w, defer_hyper=False):
        """Native first-task program: scalars (kl_hypers, kl_u, nll) of this rank's samples, and the gradient of
        w * (beta kl_hypers + kl_u + scale nll) written into every p.grad."""
        from .fused import T0Program, TnProgram
        gp, kern = self.gp, self.gp.kernel
        x, y = x.contiguous(), y.contiguous()
        S = 1 if kern.map_est else (gp.n_v if self.shards is None else self.rect[1] - self.rect[0])
        eps_u = gp.draw_u_noise(x) if hasattr(gp, 'draw_u_noise') else None      # ep_var_mean = False models only
        if self.native_noise and not noise._injected and eps_u is None:
            eps_theta = eps_f = None
        else:
            eps_theta, eps_f = gp.draw_t0_noise(x)
            eps_theta, eps_f = None if eps_theta is None else eps_theta.contiguous(), eps_f.contiguous()
        shape = T0Program.shape_of(S, gp.z, x, gp.likelihood.n_f)
        self._tn = bool(gp._use_block_program(x.size(0)))
        if self._tn:
            shape = shape + (len(gp.prev_params) + 1,)
        if self._prog is None or self._prog.shape != shape:
            if shape not in self._progs:
                self._progs[shape] = (TnProgram if self._tn else T0Program)(*shape, x.device, kern.map_est)
                if self.native_noise:
                    self._progs[shape].set_rng(self.noise_seed, self._rng_counter,
                                               self.rank * S if self.sample_offset is None else self.sample_offset)
            self._prog = self._progs[shape]
        key = (scale, w)
        if key not in self._seeds:
            self._seeds[key] = torch.tensor([self.beta * w, w, scale * w], dtype=torch.float32, device=x.device)
        packed = gp._tn_operands() if self._tn else ()
        # (first-task program: forward and backward are issued back to back here and nll is read after both, so the
        # likelihood may be left to the backward's tile kernel -- one launch less)
        extra = dict(eps_u=eps_u) if self._tn else dict(defer_softmax=os.environ.get('VARGP_DEFER_SOFTMAX', '1') != '0')
        scal = self._prog.forward(kern.log_mean.detach(), kern.log_logvar.detach(), kern.prior_log_mean,
                                  kern.prior_log_logvar, gp.z.detach(), gp.u_mean.detach(), gp.u_tril_vec.detach(), *packed,
                                  x, y, eps_theta, eps_f, bump=self._bump, **extra)
        # (a tensor that is frozen / not among the optimiser's parameters has no .grad: the program still writes all five
        # gradients, those into scratch)
        def gbuf(t):
            if t.grad is not None:
                return t.grad
            if t not in self._scratch_grads:          # keyed by the tensor itself: holds a reference, allocated once
                self._scratch_grads[t] = torch.empty_like(t)
            return self._scratch_grads[t]
        self._prog.backward(self._seeds[key], gbuf(kern.log_mean), gbuf(kern.log_logvar), gbuf(gp.z), gbuf(gp.u_mean),
                            gbuf(gp.u_tril_vec), defer_hyper=defer_hyper)
        return scal

    def _local_part(self, x, y):
        """This rank's share: gradients of w_r (beta kl_h + kl_u_r + (N/B) nll_r) accumulated into the flat buffer, whose
        tail carries w_r kl_u_r and w_r nll_r  (w_r = S_r / S; 1 / world for equal shards)."""
        scale = (self.n_total if self.n_total is not None else x.size(0)) / x.size(0)
        if self.params[0].grad is None or self.params[0].grad.data_ptr() != self.flat.data_ptr():
            off = 0
            for p in self.params:                      # (re-)attach the gradient views of the flat buffer
                p.grad = self.flat[off:off + p.numel()].view_as(p)
                off += p.numel()
        if self.class_split:
            self._pair_part1(x, y)
            self.gather_moments()
            return self._pair_part2(x, y)
        w = self.weight
        if self._t0:
            scal = self._t0_fwd_bwd(x, y, scale, w)      # overwrites every gradient view of the flat buffer
            torch.mul(scal[1:3], w, out=self.scalars)
            return scal[0], self.scalars[0], self.scalars[1]
        self.flat.zero_()
        kl_h, kl_u, nll = self.loss_fn(x, y)
        loss = (self.beta * kl_h + kl_u + scale * nll) * w
        loss.backward()
        with torch.no_grad():
            self.scalars[0] = kl_u.detach() * w
            self.scalars[1] = nll.detach() * w
        return kl_h.detach(), self.scalars[0], self.scalars[1]

    # -- class-sharded route (split_pairs with more ranks than hyper-samples) ---------------------------------------------------
    def _pair_setup(self, x):
        """Buffers of the class-sharded route for this minibatch size: the send block (max_pairs, 2, B) of this rank's moments, the
        gathered (world, max_pairs, 2, B), the row map that puts the gathered rows into (S, C) order, the full moments / gradients."""
        B, dev = x.size(0), x.device
        if getattr(self, '_pair_B', None) == B:
            return
        bufs = self.__dict__.setdefault('_pair_bufs', {})        # per minibatch size, never freed: captured graphs hold the pointers
        if B not in bufs:
            St, Ct = self.grid
            perm = torch.empty(St * Ct, dtype=torch.int64)
            for r, (s0, s1, c0, c1) in enumerate(self.shards):
                k = 0
                for s_ in range(s0, s1):
                    for c_ in range(c0, c1):
                        perm[s_ * Ct + c_] = r * self.max_pairs + k
                        k += 1
            bufs[B] = dict(_send=torch.zeros(self.max_pairs, 2, B, device=dev),
                           _recv=torch.zeros(self.world * self.max_pairs, 2, B, device=dev), _perm=perm.to(dev),
                           _full=torch.zeros(St * Ct, 2, B, device=dev),          # [:, 0] = mu, [:, 1] = var, rows in (s, c) order
                           _mom=torch.zeros(2, St, Ct, B, device=dev),            # the same, contiguous per moment
                           _gfull=torch.zeros(2, St, Ct, B, device=dev),          # seeded d nll / d (mu, var) of ALL pairs
                           _nll=torch.zeros((), device=dev))
        self.__dict__.update(bufs[B])
        self._pair_B = B
        self.__dict__.setdefault('_lik_seed', {})

    def gather_moments(self):
        """All-gather of the ranks' predictive moments (the class-sharded route's extra exchange: 2 max_pairs B floats per rank;
        BASELINE config 2 on 8 GPUs: 20 KB each)."""
        dist.all_gather_into_tensor(self._recv, self._send, group=self.group)

    def _pair_noise(self, x):
        St, Ct = self.grid
        gp = self.gp
        if gp is not None and hasattr(gp, 'kernel'):
            D1, F_ = gp.kernel.log_mean.shape[0], gp.likelihood.n_f
        else:
            D1, F_ = self._pair_dims
        eps_theta = noise.draw('eps_theta', (St, D1), x.device)
        eps_f = noise.draw('eps_f', (St, F_, Ct, x.size(0)), x.device)
        return eps_theta, eps_f

    def _pair_part1(self, x, y):
        """Moments and KL of this rank's (sample, class) rectangle -> the send block."""
        from .fused import T0Program, TnProgram
        self._pair_setup(x)
        s0, s1, c0, c1 = self.rect
        Sl, Cl, B = s1 - s0, c1 - c0, x.size(0)
        if self.params[0].grad is None or self.params[0].grad.data_ptr() != self.flat.data_ptr():
            off = 0
            for p in self.params:
                p.grad = self.flat[off:off + p.numel()].view_as(p)
                off += p.numel()
        self.flat.zero_()                           # the program / autograd writes this rank's class rows only
        eps_theta, self._eps_f = self._pair_noise(x)
        th = eps_theta[s0:s1].contiguous()
        if self.pair_fns is not None:               # generic route (autograd): moments_fn(x, rect, eps_theta rows) ->
            kl_h, kl_u, mu, var = self.pair_fns[0](x, self.rect, th)       # (kl_hypers, kl_u of the rectangle, mu, var (Sl, Cl, B))
            self._pair_graph = (kl_h, kl_u, mu, var)
            self._send[:Sl * Cl, 0].copy_(mu.detach().reshape(Sl * Cl, B))
            self._send[:Sl * Cl, 1].copy_(var.detach().reshape(Sl * Cl, B))
            return
        gp, kern = self.gp, self.gp.kernel
        assert self._t0, 'class-sharded steps need a native program (RBFKernel, ep_var_mean=True) or pair_fns'
        x = x.contiguous()
        tn = bool(gp._use_block_program(B))
        shape = (Sl, Cl, gp.M, gp.z.size(-1), B, gp.likelihood.n_f) + ((len(gp.prev_params) + 1,) if tn else ())
        if self._prog is None or self._prog.shape != shape:
            if shape not in self._progs:
                self._progs[shape] = (TnProgram if tn else T0Program)(*shape, x.device, kern.map_est)
            self._prog = self._progs[shape]
        self._tn = tn
        packed = tuple(t[c0:c1] for t in gp._tn_operands()) if tn else ()
        self._pair_scal = self._prog.forward(kern.log_mean.detach(), kern.log_logvar.detach(), kern.prior_log_mean,
                                             kern.prior_log_logvar, gp.z.detach()[c0:c1], gp.u_mean.detach()[c0:c1],
                                             gp.u_tril_vec.detach()[c0:c1], *packed, x, y.contiguous(),
                                             None if kern.map_est else th, None, bump=self._bump, ext_lik=True)
        mu, var, _, _ = self._prog.lik_buffers()
        self._send[:Sl * Cl, 0].copy_(mu.view(Sl * Cl, B))
        self._send[:Sl * Cl, 1].copy_(var.view(Sl * Cl, B))

    def _pair_part2(self, x, y):
        """The likelihood of ALL pairs from the gathered moments, then the backward of this rank's rectangle; leaves
        [grads | w kl_u | w nll] in the flat buffer for the exchange.  -> (kl_hypers, kl_u share, nll share)."""
        from . import _lib
        St, Ct = self.grid
        s0, s1, c0, c1 = self.rect
        Sl, Cl, B = s1 - s0, c1 - c0, x.size(0)
        scale = (self.n_total if self.n_total is not None else B) / B
        torch.index_select(self._recv, 0, self._perm, out=self._full)
        if self.pair_fns is not None:
            kl_h, kl_u, mu, var = self._pair_graph
            mu_f = self._full[:, 0].reshape(St, Ct, B).clone().requires_grad_(True)
            var_f = self._full[:, 1].reshape(St, Ct, B).clone().requires_grad_(True)
            nll = self.pair_fns[1](mu_f, var_f, y, self._eps_f)
            gmu, gvar = torch.autograd.grad(scale * nll, [mu_f, var_f])
            local = (self.beta * self.w_h) * kl_h + self.w_kl * kl_u + (mu * gmu[s0:s1, c0:c1]).sum() + (var * gvar[s0:s1, c0:c1]).sum()
            local.backward()
            self._pair_graph = None
            with torch.no_grad():
                self.scalars[0] = kl_u.detach() * self.w_kl
                self.scalars[1] = nll.detach() * self.w_h
            return kl_h.detach(), self.scalars[0], self.scalars[1]
        gp, kern = self.gp, self.gp.kernel
        lib, ptr, st = _lib.lib(), _lib.ptr, _lib.stream_ptr()
        self._mom.copy_(self._full.view(St, Ct, 2, B).permute(2, 0, 1, 3))         # (static buffer: no allocation under capture)
        mu_f, var_f = self._mom[0], self._mom[1]
        eps_f, yc = self._eps_f.contiguous(), y.contiguous()
        F_ = eps_f.shape[1]
        if scale not in self._lik_seed:
            self._lik_seed[scale] = torch.tensor([scale], dtype=torch.float32, device=x.device)
        _lib.check(lib.vargp_softmax_nll_fwd(ptr(mu_f), ptr(var_f), ptr(eps_f), ptr(yc), ptr(self._nll), St, F_, Ct, B, st),
                   'vargp_softmax_nll_fwd')
        _lib.check(lib.vargp_softmax_nll_bwd(ptr(mu_f), ptr(var_f), ptr(eps_f), ptr(yc), ptr(self._lik_seed[scale]),
                                             ptr(self._gfull[0]), ptr(self._gfull[1]), St, F_, Ct, B, st), 'vargp_softmax_nll_bwd')
        _, _, gmu, gvar = self._prog.lik_buffers()
        gmu.copy_(self._gfull[0, s0:s1, c0:c1])
        gvar.copy_(self._gfull[1, s0:s1, c0:c1])
        key = ('pair', scale)
        if key not in self._seeds:
            self._seeds[key] = torch.tensor([self.beta * self.w_h, self.w_kl, 0.0], dtype=torch.float32, device=x.device)
        g = lambda t: t.grad[c0:c1]
        self._prog.backward(self._seeds[key], kern.log_mean.grad, kern.log_logvar.grad if kern.log_logvar.grad is not None
                            else self._scratch(kern.log_logvar), g(gp.z), g(gp.u_mean), g(gp.u_tril_vec))
        torch.mul(self._pair_scal[1], self.w_kl, out=self.scalars[0])
        torch.mul(self._nll, self.w_h, out=self.scalars[1])
        return self._pair_scal[0], self.scalars[0], self.scalars[1]

    def _scratch(self, t):
        if t not in self._scratch_grads:
            self._scratch_grads[t] = torch.empty_like(t)
        return self._scratch_grads[t]
