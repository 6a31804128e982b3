#!/usr/bin/env python3
"""Headline benchmark: training sessions/sec of the IntEL hot path on synthetic Tmall-shape data
(BASELINE.json: list=50, K=3 base rankers, 64-d embeddings, 1M-item table) at 1/2/4/8 MI355X.

A "step" = one pass of the hot path over one batch per GPU: forward -> BPR loss (+intent loss) ->
hand-written backward -> gradient all-reduce (N>1) -> dense Adam over all parameters, inputs resident
in HBM.  Prints ONE JSON line (rank 0).  `python bench.py --gpus N --steps K --warmup W`; for N>1 launch
with torch.distributed.run (one rank per GPU, RCCL).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12            # B/s   (MI355X_MICROARCH.md: HBM3E 8 TB/s spec)
F32_MFMA_PEAK = 157.3e12     # FLOP/s (fp32-input MFMA = fp32 vector peak)
BF16_MFMA_PEAK = 2.5e15      # FLOP/s dense bf16 MFMA
B3_KERNELS = ('gemm_rows_b3_kernel', 'gemm_rows_b3k_kernel', 'wgrad_b3_kernel', 'linear_bwd_pair_kernel', 'linear_bwd_qkv_kernel')
FUSED_KERNELS = ('tower_fwd_fused_kernel', 'tower_bwd_fused_kernel')
MFMA_KERNELS = ('gemm_rows_kernel', 'gemm_rows_w8_kernel', 'gemm_rows_w8k_kernel', 'wgrad_pipe_kernel',
                'attn_fwd_kernel', 'attn_bwd_dkv_kernel', 'attn_seq_fwd_kernel', 'attn_seq_bwd_fused_kernel', 'attn_bwd_dq_ds_kernel',
                'tw32_fwd_kernel', 'tw32_bwd_kernel', 'enc32_fwd_kernel', 'enc32_bwd_kernel')      # exact fp32 MFMAs (v_mfma_f32_16x16x4_f32)


def feed_throughput(w, cinfo, B, dev, reps=20):
    """sessions/s of intel_feed_collate (columnar corpus in HBM -> one padded batch) on a synthetic corpus
    of the workload's shape; the kernel is HBM-bound, so the achieved GB/s over the bytes it writes and reads is reported."""
    import numpy as np
    import torch
    from intel_sigir2023_amd import feed
    K, I = w['flags']['model_num'], cinfo['I']
    Lb, H = w['batch']['L'], w['batch']['H']
    n_sess = max(4 * B, 16384)
    st = feed.ColumnarStore.synthetic(n_sess, Lb, K, I, H, min(cinfo['items'], 1 << 20), min(cinfo['users'], 1 << 16), cinfo['classes'],
                                      cinfo['ctx']).to(dev)
    rs = np.random.RandomState(1)
    idx = [torch.from_numpy(rs.randint(0, n_sess, B).astype(np.int32)).to(dev) for _ in range(4)]
    shape = st.max_shape()

    def one(i):
        return st.collate(idx[i % 4], shuffle='device', seed=i, shape=shape)
    for i in range(3):
        one(i)
    torch.cuda.synchronize()
    t0 = time.time()
    for i in range(reps):
        one(i)
    torch.cuda.synchronize()
    el = (time.time() - t0) / reps
    out_bytes = B * (4 * 3 * Lb + 4 * Lb * K + 4 * I + 4 * H + 4 * H * I + 8 * H + 24)
    in_bytes = B * (Lb * (8 + 8 * K) * 2 + 4 * I + H * (8 + 4 * I) + 64)
    return {'sessions_per_s': round(B / el, 1), 'ms_per_batch': round(el * 1e3, 4), 'achieved_GBps': round((out_bytes + in_bytes) / el / 1e9, 1),
            'peak_GBps': HBM_PEAK / 1e9, 'store_MB': round(st.nbytes() / 1e6, 1), 'shuffle': 'device (counter-based RNG)'}


def algorithmic_bytes_per_session(flags, corpus, shape, train, e=4):
    """bytes a session's forward (and training extras) must move, fp32 (e=4)."""
    L, H, Hi = shape['L'], shape['H'], shape['H']
    K, I = flags['model_num'], corpus['I']
    d_id, d_im, d_u, d_c = flags['i_emb_size'], flags['im_emb_size'], flags['u_emb_size'], flags['context_emb_size']
    fwd = e * ((L + Hi) * d_id + d_u) + e * (L * d_im + (1 + H) * d_c) + 4 * (2 * L + 2 * Hi + H + 5) + 4 * L * K \
        + 4 * H * I + 4 * (L + L * K + I)
    if not train:
        return fwd
    return fwd + 4 * (L + I) + 4 * ((L + Hi) * d_id + d_u)


def cpu_baseline(args_ns, corpus, cinfo, workload, loss_name, budget_s=20.0):
    """The oracle (CPU restatement, kind='port') timed on this host's cores on a bounded sample of the same synthetic
    workload: B = 512 and B = 4096 sessions per step, each split into forward + loss / + backward / full
    training step (forward, loss, autograd, torch Adam -- the dense Adam sweep over the 1 M-row table dominates).
    `value` = full-step sessions/s at B = 512 (the published scripts' batch size)."""
    import torch
    from intel_sigir2023_amd import synth
    from intel_sigir2023_amd.model import IntEL
    from oracle import intel_oracle as O
    torch.manual_seed(0)
    cpu = torch.device('cpu')
    flags = {k: v for k, v in vars(args_ns).items() if k != 'device'}
    cfg = O.Config(**flags)
    a = argparse.Namespace(**vars(args_ns))
    a.device = cpu
    m = IntEL(a, corpus)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    del m
    opt = torch.optim.Adam(O.adam_groups(list(sd.items()), 1e-4), lr=1e-3)

    def loss_of(batch, B):
        out = O.forward(sd, batch, cfg)
        if 'BPR' in loss_name:
            noise = torch.rand(B, batch['i_id_s'].shape[1], batch['i_id_s'].shape[1])
            return O.int_bpr_loss(out, batch, cfg, noise)[0]
        return O.int_list_loss(out, batch, cfg)[0]

    def stage(batch, B, kind):
        if kind == 'fwd_loss':
            with torch.no_grad():
                return float(loss_of(batch, B))
        opt.zero_grad()
        loss = loss_of(batch, B)
        loss.backward()
        if kind == 'full_step':
            opt.step()
        return float(loss.detach())

    def timed(batch, B, kind, budget, max_n):
        stage(batch, B, kind)                  # warm-up (allocations, lazy init) at both batch sizes
        t0 = time.perf_counter()
        n = 0
        while True:
            stage(batch, B, kind)
            n += 1
            el = time.perf_counter() - t0
            if el >= budget or n >= max_n:
                return n, el
    # the host: model name, physical cores (BASELINE.md 3 quotes physical cores), logical CPUs
    model_name, phys = 'unknown', set()
    try:
        pid = cid = None
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                model_name = line.split(':', 1)[1].strip()
            elif line.startswith('physical id'):
                pid = line.split(':', 1)[1].strip()
            elif line.startswith('core id'):
                cid = line.split(':', 1)[1].strip()
                phys.add((pid, cid))
    except OSError:
        pass
    logical = os.cpu_count() or 1
    physical = len(phys) or logical
    # thread sweep: tiny GEMMs do not scale to every hardware thread (128 torch threads ran the B=512 forward 3x slower than 8 did
    # in the survey container) -- every candidate count runs one warmed B=512 full step, the fastest count is used for all stages
    t_start = time.perf_counter()
    b512 = synth.to_reference_layout(synth.make_batch(workload, 512, cpu, seed=99), cinfo['I'])
    quick = budget_s < 10.0               # (the GPU suite's contract-field test: one thread count, B = 512 only)
    cand = [min(16, logical)] if quick else sorted({c for c in (8, 16, 32, 64, physical) if 1 <= c <= logical})
    sweep = {}
    for c in cand:
        torch.set_num_threads(c)
        stage(b512, 512, 'full_step')
        t0 = time.perf_counter()
        stage(b512, 512, 'full_step')
        sweep[c] = time.perf_counter() - t0
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    total = time.perf_counter() - t_start
    budget_s = max(3.0 if quick else 8.0, budget_s - total)
    stages = {}
    for B, share, max_n in (((512, 1.0, 40),) if quick else ((512, 0.5, 40), (4096, 0.5, 6))):
        batch = b512 if B == 512 else synth.to_reference_layout(synth.make_batch(workload, B, cpu, seed=99), cinfo['I'])
        st = {}
        for kind, frac in (('fwd_loss', 0.2), ('fwd_loss_bwd', 0.3), ('full_step', 0.5)):
            n, el = timed(batch, B, kind, budget_s * share * frac, max_n)
            st[kind] = {'sessions_per_s': round(B * n / el, 2), 'ms_per_step': round(1e3 * el / n, 1), 'steps': n}
        stages['B%d' % B] = st
    total = time.perf_counter() - t_start
    return {'value': stages['B512']['full_step']['sessions_per_s'], 'unit': 'sessions/s', 'cores': int(best),
            'kind': 'port', 'stages': stages,
            'host': {'cpu': model_name, 'physical_cores': physical, 'logical_cpus': logical},
            'thread_sweep_ms_per_B512_step': {str(c): round(1e3 * v, 1) for c, v in sorted(sweep.items())},
            'sample': 'oracle/intel_oracle.py on synthetic %s sessions with the fastest torch thread count of the sweep (%d of %d physical '
                      'cores): B=512 and B=4096 per step, each warmed and timed as forward+loss, +autograd backward, full step with torch '
                      'Adam (%d / %d timed full steps), %.1f s of CPU work in total; value = full step at B=512'
                      % (workload, best, physical, stages['B512']['full_step']['steps'], stages['B4096']['full_step']['steps'] if 'B4096' in stages else 0, total)}


def pmc_stale(prof_shapes, psteps, pmc_kernels):
    """Does the committed PMC summary (tools/pmc_summary.py) describe the step THIS build runs?  The library's kernels of the live profile and their
    launches per step must equal the file's (torch's own kernels, which only the file sees, aside).  Returns None when they match, else what differs --
    `roofline.traffic` is then withheld instead of quoting bytes of a step that no longer exists."""
    if not pmc_kernels:
        return {'reason': 'no PMC summary'}
    live = {}
    for k, v in prof_shapes.items():
        base = k.split('[')[0].strip('()').split('<')[0]
        live[base] = live.get(base, 0) + v['launches']
    live = {k: round(n / psteps, 2) for k, n in live.items()}
    foreign = ('at::', 'rocprim', '__amd_rocclr', 'void at::', 'softmax_warp')
    # launch counts that depend on the SCHEDULE, not on the build: the live profile runs single-stream with the table sweep on the main stream (two dense
    # adam_kernel launches, one slab reduction per dependency round), the PMC passes run the step as the timed loop does (adam_pair_kernel, branch-local
    # reductions).  Presence still counts for them (one of the two Adam forms, the reduction kernel), the count does not
    sched = ('adam_kernel', 'adam_pair_kernel', 'slab_reduce_batch_kernel')
    filed = {k: v['launches_per_step'] for k, v in pmc_kernels.items() if not k.startswith(foreign) and 'elementwise_kernel' not in k}
    diff = {k: [live.get(k), filed.get(k)] for k in sorted(set(live) | set(filed)) if k not in sched and (live.get(k) is None or filed.get(k) is None or abs(live[k] - filed[k]) > 0.34)}
    if not (('adam_kernel' in live or 'adam_pair_kernel' in live) == ('adam_kernel' in filed or 'adam_pair_kernel' in filed)) or \
       (('slab_reduce_batch_kernel' in live) != ('slab_reduce_batch_kernel' in filed)):
        diff['optimizer / reduction kernels'] = [sorted(k for k in sched if k in live), sorted(k for k in sched if k in filed)]
    return {'reason': 'kernel set / launches per step differ from the PMC summary', 'live_vs_file': diff} if diff else None


def price_dominant_kernel(prof_shapes, psteps, pmc_kernels, pmc_source, exact_shape, planes=6.0):
    """`roofline` object for the kernel with the largest total time among the profiled launches: achieved = algorithmic bytes
    (or flops) / launch duration (HIP events on the launch stream), traffic = HBM bytes per launch from the committed PMC passes."""
    prof, split = {}, {}            # aggregate the shape-tagged records by kernel; per kernel also by row count (M of [MxNxK])
    for k, v in prof_shapes.items():
        base = k.split('[')[0].strip('()').split('<')[0]
        d = prof.setdefault(base, {'launches': 0, 'ms': 0.0, 'flops': 0.0, 'bytes': 0.0})
        for f in d:
            d[f] += v[f]
        if '[' in k:
            m = int(k.split('[')[1].split('x')[0])
            part = split.setdefault(base, {}).setdefault('rows_ge_32768' if m >= 32768 else 'rows_lt_32768', {'launches': 0, 'ms': 0.0, 'flops': 0.0, 'bytes': 0.0})
            for f in part:
                part[f] += v[f]
    tot = sum(v['ms'] for v in prof.values())
    name, dom = max(prof.items(), key=lambda kv: kv[1]['ms'])
    avg_ms = dom['ms'] / dom['launches']
    if name in B3_KERNELS:
        # fp32-accurate products on the bf16 pipe (three-plane split, 6 bf16 MFMAs per fp32 product): the matrix pipe is
        # far from its 2.5 PFLOP/s roof (reported as `bf16_mfma_frac`); the binding roof is HBM
        ach = dom['bytes'] / (dom['ms'] * 1e-3) / 1e9
        roof = {'bound': 'hbm', 'achieved': round(ach, 2), 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s', 'frac': round(ach / (HBM_PEAK / 1e9), 5),
                'traffic': None, 'bf16_mfma_frac': round(planes * dom['flops'] / (dom['ms'] * 1e-3) / BF16_MFMA_PEAK, 5),
                'fp32_equivalent_TFLOPs': round(dom['flops'] / (dom['ms'] * 1e-3) / 1e12, 2)}
        # the same kernel serves the B*L-row products of the towers (HBM-bound) and the B-row products of the session head
        # (one tile per workgroup, latency-bound): priced separately too
        roof['by_rows'] = {kk: {'launches_per_step': vv['launches'] / psteps, 'avg_launch_ms': round(vv['ms'] / vv['launches'], 5),
                                'achieved': round(vv['bytes'] / (vv['ms'] * 1e-3) / 1e9, 2), 'frac': round(vv['bytes'] / (vv['ms'] * 1e-3) / HBM_PEAK, 5)}
                           for kk, vv in sorted(split.get(name, {}).items())}
    elif name in MFMA_KERNELS:
        ach = dom['flops'] / (dom['ms'] * 1e-3) / 1e12
        roof = {'bound': 'mfma', 'achieved': round(ach, 3), 'peak': F32_MFMA_PEAK / 1e12, 'unit': 'TFLOP/s',
                'frac': round(ach / (F32_MFMA_PEAK / 1e12), 5), 'traffic': None}
    elif name in FUSED_KERNELS:
        # the one-kernel tower layer: six bf16 plane products per linear + exact fp32-MFMA attention; priced against the
        # dense bf16 MFMA peak with the six-fold plane work counted (the HBM side is reported next to it)
        eq = planes * dom['flops'] / (dom['ms'] * 1e-3) / 1e12
        roof = {'bound': 'mfma', 'achieved': round(eq, 2), 'peak': BF16_MFMA_PEAK / 1e12, 'unit': 'TFLOP/s',
                'frac': round(eq / (BF16_MFMA_PEAK / 1e12), 5), 'traffic': None,
                'fp32_equivalent_TFLOPs': round(dom['flops'] / (dom['ms'] * 1e-3) / 1e12, 2),
                'hbm_GBps': round(dom['bytes'] / (dom['ms'] * 1e-3) / 1e9, 1)}
    else:
        ach = dom['bytes'] / (dom['ms'] * 1e-3) / 1e9
        roof = {'bound': 'hbm', 'achieved': round(ach, 2), 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s',
                'frac': round(ach / (HBM_PEAK / 1e9), 5), 'traffic': None}
    if pmc_kernels and exact_shape:
        stale = pmc_stale(prof_shapes, psteps, pmc_kernels)
        if stale is None and name in pmc_kernels:
            roof['traffic'] = pmc_kernels[name]['hbm_bytes_per_launch']
            roof['traffic_source'] = pmc_source
        elif stale is not None:
            roof['traffic_stale'] = dict(stale, source=pmc_source)
    roof.update({'kernel': name, 'launches_per_step': dom['launches'] / psteps, 'avg_launch_ms': round(avg_ms, 5),
                 'share_of_kernel_time': round(dom['ms'] / tot, 4),
                 'algorithmic_per_launch': (dom['flops'] if roof['bound'] == 'mfma' else dom['bytes']) / dom['launches']})
    return roof, prof


def short_line(workload, B, over, steps, warmup, dev):
    """One short training measurement of another workload (fresh model + engine, 4 resident batches, the workload's own optimizer settings and
    table-Adam policy), same brackets as the headline's timed block."""
    import gc
    import torch
    from intel_sigir2023_amd import synth
    from intel_sigir2023_amd.engine import IntELEngine
    from intel_sigir2023_amd.model import IntEL
    w = synth.WORKLOADS[workload]
    over = dict(over)
    loss_name = over.pop('loss', 'IntBPRloss')
    args_ns = synth.make_args(workload, dev, **over)
    corpus, cinfo = synth.make_corpus(workload)
    torch.manual_seed(0)
    model = IntEL(args_ns, corpus).to(dev)
    lr, l2 = w.get('optim', (1e-3, 1e-4))
    eng = IntELEngine(model, loss_name, args_ns, lr=lr, l2=l2, lazy_table='auto')
    eng.defer_table_wait = True
    batches = [synth.make_batch(workload, B, dev, seed=50 + i) for i in range(4)]
    for bt in batches:
        bt['_intel'] = model.prepare_batch(bt)
        bt['_intel'][1]['ranking_i32'] = bt['ranking']
    for i in range(warmup):
        eng.train_step(batches[i % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        eng.train_step(batches[i % 4])
    eng.flush()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    out = {'workload': workload, 'batch': B, 'loss': loss_name, 'cal_diversity': int(args_ns.cal_diversity), 'steps': steps, 'warmup': warmup,
           'value': round(B * steps / max(el, 1e-9), 1), 'unit': 'sessions/s', 'ms_per_step': round(1e3 * el / max(1, steps), 4),
           'table_adam': 'lazy' if eng._lazy is not None else 'dense'}
    del eng, model, batches
    gc.collect()
    torch.cuda.empty_cache()
    return out


def self_launch(n, argv):
    """Start the N ranks of `bench.py --gpus N` as children (python -m torch.distributed.run, one process per GPU, rendezvous on 127.0.0.1), relay
    rank 0's JSON line and check that it really is an N-rank line.  Returns the exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:          # a free port for the rendezvous
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '8')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1', '--master-port', str(port),
           os.path.abspath(__file__)] + list(argv)
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith('{'):
            line = ln
        else:
            print(ln, file=sys.stderr)
    if p.returncode != 0:
        print('bench.py: the %d-rank launch exited with code %d' % (n, p.returncode), file=sys.stderr)
        return p.returncode or 1
    try:
        d = json.loads(line)
    except Exception:
        print('bench.py: the %d-rank launch printed no JSON line' % n, file=sys.stderr)
        return 1
    if d.get('n_gpus') != n or d.get('rccl_ranks') != n:
        print('bench.py: asked for %d GPUs, the line says n_gpus=%s rccl_ranks=%s' % (n, d.get('n_gpus'), d.get('rccl_ranks')), file=sys.stderr)
        return 1
    print(line)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', type=str, default='tmall', help='tmall | tmall_pub | tmall_pub_mse | lifedata | stress | tiny')
    ap.add_argument('--batch', type=int, default=0, help='sessions per GPU per step (weak scaling); 0: 4096, or 512 for tmall_pub')
    ap.add_argument('--loss', type=str, default='IntBPRloss')
    ap.add_argument('--cal_diversity', type=int, default=-1, help='-1: the workload default')
    ap.add_argument('--zipf', type=int, default=0, help='1: Zipf(1.05) item popularity instead of uniform')
    ap.add_argument('--dtype', type=str, default='f32', help='f32: the parity mode (headline); bf16: single bf16 product per linear')
    ap.add_argument('--adam', type=str, default='auto', help='auto: lazy when a step touches at most 1/4 of the item-id table (engine.py), dense otherwise; lazy: the item-id table\'s dense Adam in its lazy form (rows replayed when they are next read; the whole table '
                    'is settled INSIDE the timed region after the last step); dense: one sweep over the whole table every step')
    ap.add_argument('--nbatches', type=int, default=8, help='distinct resident batches cycled by the timed loop')
    ap.add_argument('--eval_steps', type=int, default=-1, help='evaluation steps timed after the training loop (-1: max(3, steps/2); 0: none)')
    ap.add_argument('--no_cpu_baseline', action='store_true')
    ap.add_argument('--no_bf16_line', action='store_true', help='skip the bf16-mode measurement appended to the fp32 line at N=1')
    ap.add_argument('--no_roofline', action='store_true')
    ap.add_argument('--no_feed', action='store_true', help='skip the device-feed (batch assembly) throughput measurement')
    ap.add_argument('--no_defer_table', action='store_true', help='every step waits for the item-id table sweep before it returns (default: the next forward starts under it)')
    ap.add_argument('--no_workloads', action='store_true', help='skip the short driver-timed lines of the other workloads (tmall_pub, lifedata, stress) appended at N=1')
    ap.add_argument('--spread_blocks', type=int, default=2, help='extra timed blocks of --steps steps after the one that defines `value` (value_spread)')
    ap.add_argument('--cpu_budget', type=float, default=24.0)
    ap.add_argument('--shapes', action='store_true', help='also report per-GEMM-shape timings')
    ap.add_argument('--encoder', type=str, default='', help='override the sequence encoder: BERT4Rec | GRU4Rec')
    ap.add_argument('--dry_launch', action='store_true', help='stop after the process group is up: rank 0 prints {"dry_launch": true, "n_gpus": N, '
                    '"rccl_ranks": N, "backend": ...}; checks the N-rank launch path without a GPU (tests/test_bench_launch_cpu.py)')
    a = ap.parse_args()

    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` on its own: become the launcher.  Nothing in this process has touched the GPU yet (torch is not even imported), the
        # N ranks are CHILD processes (never an exec), rank 0's JSON line is relayed and checked
        raise SystemExit(self_launch(a.gpus, sys.argv[1:]))

    import torch
    from intel_sigir2023_amd import _lib, parallel, synth
    from intel_sigir2023_amd.engine import IntELEngine
    from intel_sigir2023_amd.model import IntEL
    rank, world, local_rank = parallel.init_distributed()
    if world != a.gpus:
        # a line that says n_gpus = WORLD_SIZE under --gpus N would be read as an N-GPU measurement
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d (plain `python bench.py --gpus N` launches its own N ranks; under torch.distributed.run '
                         'pass the same N as --nproc-per-node)' % (a.gpus, world))
    ranks_seen = parallel.count_ranks()      # one all-reduce of ones through the group that will carry the gradients (1 without a group)
    if ranks_seen != a.gpus:
        raise SystemExit('bench.py: the process group reports %d ranks, --gpus %d' % (ranks_seen, a.gpus))
    if a.dry_launch:
        if rank == 0:
            print(json.dumps({'dry_launch': True, 'n_gpus': world, 'rccl_ranks': ranks_seen, 'backend': parallel.backend_name()}))
        parallel.barrier()
        return
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU fallback in the product path)')
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)
    _lib.lib()
    w = synth.WORKLOADS[a.workload]
    over = {}
    if a.cal_diversity >= 0:
        over['cal_diversity'] = a.cal_diversity
    if a.encoder:
        over['encoder'] = a.encoder
    if a.dtype != 'f32':
        over['dtype'] = a.dtype
    args_ns = synth.make_args(a.workload, dev, **over)
    corpus, cinfo = synth.make_corpus(a.workload)
    torch.manual_seed(0)
    model = IntEL(args_ns, corpus).to(dev)
    lr, l2 = w.get('optim', (1e-3, 1e-4))
    lazy = {'lazy': True, 'dense': False}.get(a.adam, 'auto')
    eng = IntELEngine(model, a.loss, args_ns, lr=lr, l2=l2, lazy_table=lazy)
    eng.defer_table_wait = not a.no_defer_table      # back-to-back steps: the next forward starts under the table's Adam sweep (engine.py)
    parallel.broadcast_(eng.param_buckets())
    B = a.batch or w.get('bench_batch', 4096)
    # distinct resident batches (inputs in HBM before the timed region): 8 x 73 MB of gathered item rows at the headline shape,
    # so the embedding gather is not served by the 256 MB Infinity Cache from one step to the next.  The reference-layout ->
    # ABI narrowing (model.prepare_batch: a no-op for the feed's int32 / fp32 batches) happens once per batch, here
    nbatches = max(1, a.nbatches)
    batches = [synth.make_batch(a.workload, B, dev, seed=rank * 1000 + i, zipf=bool(a.zipf)) for i in range(nbatches)]
    for bt in batches:
        bt['_intel'] = model.prepare_batch(bt)
        bt['_intel'][1]['ranking_i32'] = bt['ranking']
    Lmax = w['batch']['L']

    def one_step(i):
        return eng.train_step(batches[i % nbatches])
    loss = None
    # The synthetic corpus and the resident batches are a few million Python objects; a generation-2 collection that walks them costs the host
    # 40 - 90 ms, and when one falls into a 20-step timed block it IS the result (LifeData shape, lazy table: 3.3 -> 5.5 ms per step).  Everything
    # built so far is long-lived: collect once, then keep it out of the collector's generations.
    import gc
    gc.collect()
    gc.freeze()
    for i in range(a.warmup):
        loss = one_step(i)
    torch.cuda.synchronize()
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        loss = one_step(i)
    eng.flush()                 # lazy table Adam: every row left behind is brought up to the last step inside the timed region
    torch.cuda.synchronize()
    parallel.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    el = parallel.allreduce_max_float(el, dev)
    last_loss = float(loss[0]) if (a.steps + a.warmup) > 0 else float('nan')
    # the same timed block again (same brackets): `value` stays the FIRST block -- the driver's --steps -- and the spread shows what one sample is worth
    block_values = [world * B * a.steps / max(el, 1e-9)]
    for _ in range(max(0, a.spread_blocks) if a.steps > 0 else 0):
        torch.cuda.synchronize()
        parallel.barrier()
        torch.cuda.synchronize()
        tb = time.perf_counter()
        for i in range(a.steps):
            one_step(i)
        eng.flush()
        torch.cuda.synchronize()
        parallel.barrier()
        torch.cuda.synchronize()
        block_values.append(world * B * a.steps / max(parallel.allreduce_max_float(time.perf_counter() - tb, dev), 1e-9))

    # ---- eval throughput (forward + on-device NDCG@3), not part of `value`
    model.eval()
    ev_steps = a.eval_steps if a.eval_steps >= 0 else max(3, a.steps // 2)
    ev_el, ndcg3 = float('inf'), float('nan')
    if ev_steps > 0:
        for i in range(2 if a.warmup > 0 else 0):
            eng.eval_step(batches[i % nbatches], k=3)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(ev_steps):
            out, nd = eng.eval_step(batches[i % nbatches], k=3)
        torch.cuda.synchronize()
        ev_el = parallel.allreduce_max_float(time.perf_counter() - t1, dev)
        ndcg3 = float(nd.float().nan_to_num(0).mean())
    model.train()

    # ---- data-parallel exchange timing (N > 1, or a forced one-rank group): a few extra steps with HIP events around every collective
    exchange = None
    if parallel.active() and eng.overlap_table_update and eng.wide_backward:
        eng.time_exchange = True
        for i in range(5):
            one_step(i)
            eng.exchange_collect()
        eng.time_exchange = False
        exchange = eng.exchange_report()
        if exchange is not None:      # the slowest rank's view
            for k in ('table_exchange_ms', 'table_branch_ms', 'dense_buckets_allreduce_ms', 'exposed_ms'):
                exchange[k] = round(parallel.allreduce_max_float(exchange[k], dev), 4)
            exchange['ms_per_step'] = round(el / max(a.steps, 1) * 1e3, 4)

    # ---- per-kernel profile: EVERY rank runs the same extra steps (they contain the gradient all-reduce)
    prof_shapes, prof_eval, psteps = None, None, 3
    if not a.no_roofline:
        lib = _lib.lib()
        lib.intel_set_concurrency(model._context(), 0)     # price kernels one at a time on one stream
        ov, eng.overlap_table_update = eng.overlap_table_update, False
        lib.intel_prof_enable(1)
        for i in range(psteps):
            one_step(i)
        prof_shapes = json.loads(lib.intel_prof_collect().decode())
        model.eval()
        for i in range(psteps):
            eng.eval_step(batches[i % nbatches], k=3)
        prof_eval = json.loads(lib.intel_prof_collect().decode())
        model.train()
        lib.intel_prof_enable(0)
        eng.overlap_table_update = ov
        lib.intel_set_concurrency(model._context(), 1)
    if rank != 0:
        return
    f = w['flags']
    bf16 = a.dtype == 'bf16'
    arith = ('bf16 arithmetic: every linear / weight gradient is ONE bf16 MFMA product (operands rounded to bf16, fp32 accumulate); fp32 master '
             'weights, Adam moments, residual / LayerNorm / softmax; gated by NDCG@3 against the fp32 build (tests/test_bf16_gpu.py)') if bf16 else \
            ('fp32 storage and accumulation; the large products run as three-plane bf16 splits (hi+mid+lo, six plane products) on the bf16 '
             'MFMA pipe = fp32 accuracy, same parity thresholds as the fp32-MFMA kernels')
    res = {
        'metric': 'train sessions/sec, IntEL fwd+%s+bwd+Adam, synthetic %s list=%d K=%d, item / score tower %d / %d wide'
                  % (a.loss, {'tmall': 'Tmall-shape', 'lifedata': 'LifeData-shape'}.get(a.workload, a.workload), Lmax, f['model_num'], f['i_emb_size'] + f['im_emb_size'], f['s_emb_size']),
        'value': round(world * B * a.steps / max(el, 1e-9), 1), 'unit': 'sessions/s', 'n_gpus': world, 'steps': a.steps,
        'warmup': a.warmup, 'ms_per_step': round(1e3 * el / max(1, a.steps), 4), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'bf16' if bf16 else 'f32', 'data': 'synthetic', 'rccl_ranks': ranks_seen, 'backend': parallel.backend_name(),
        'config': {'workload': '%s: %d items, list=%d, K=%d rankers, I=%d intents, H=%d, emb %d/%d/%d/%d (id/meta/score/ctx), %s, %d heads x %d tied '
                               'layers, %s, %s loss, cal_diversity=%d, %s item ids, %d resident batches'
                               % (a.workload, cinfo['items'], Lmax, f['model_num'], cinfo['I'], w['batch']['H'], f['i_emb_size'], f['im_emb_size'],
                                  f['s_emb_size'], f['context_emb_size'], args_ns.encoder, f['num_heads'], f['num_layers'], a.dtype, a.loss,
                                  int(args_ns.cal_diversity), 'zipf(1.05)' if a.zipf else 'uniform', nbatches),
                   'global_batch': world * B, 'per_gpu_batch': B, 'parallelism': 'dp%d' % world, 'arithmetic': arith,
                   'table_adam': ('lazy: torch.optim.Adam\'s dense update of the item-id table, rows without a gradient replayed step by step when they are next '
                                  'gathered (bit-identical to the dense sweep, tests/test_lazy_adam_gpu.py); all rows settled inside the timed region after '
                                  'the last step') if eng._lazy is not None else 'dense sweep over the whole table every step'},
        'eval_sessions_per_s': round(world * B * ev_steps / ev_el, 1), 'ndcg3_random_init': round(ndcg3, 5),
        'loss_last_step': round(last_loss, 6),
        'value_spread': {'blocks': [round(v, 1) for v in block_values], 'min': round(min(block_values), 1), 'max': round(max(block_values), 1),
                         'note': '`value` is the first block (the driver\'s --steps); the others repeat it with the same brackets in the same process'},
    }
    if exchange is not None:      # data parallel: what the gradient exchange cost and how much of it was NOT hidden (slowest rank, 5 extra steps)
        res['exchange'] = exchange
    if rank == 0 and not a.no_feed:
        res['feed'] = feed_throughput(w, cinfo, B, dev)
    bytes_train = algorithmic_bytes_per_session(w['flags'], cinfo, w['batch'], True)
    res['gather_roofline'] = {'bytes_per_session': bytes_train, 'achieved_GBps': round(bytes_train * res['value'] / world / 1e9, 3),
                              'peak_GBps': HBM_PEAK / 1e9, 'frac': round(bytes_train * res['value'] / world / HBM_PEAK, 6)}
    if prof_shapes is not None:
        pmc, pmc_eval, src = None, None, None
        try:        # HBM traffic per launch from the committed rocprofv3 PMC passes (tools/pmc_summary.py)
            import glob         # the latest round's summary
            src = sorted(os.path.relpath(f, ROOT) for f in glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_pmc_traffic%s.json' % ('_bf16' if a.dtype == 'bf16' else ''))))[-1]
            j = json.load(open(os.path.join(ROOT, src)))
            pmc, pmc_eval = j.get('kernels'), j.get('kernels_eval', j.get('kernels'))
            try:
                from intel_sigir2023_amd import build as _b
                res['pmc_file'] = {'source': src, 'commit': j.get('commit'), 'csrc_stamp_matches_this_build': j.get('csrc_stamp') == _b._stamp()}
            except Exception:
                pass
            src += ' (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH x2 on gfx950), bytes per launch'
        except Exception:
            pass
        exact = a.workload == 'tmall' and B == 4096
        planes = 1.0 if bf16 else 6.0           # bf16 MFMA products per product of the path
        roof, prof = price_dominant_kernel(prof_shapes, psteps, pmc, src, exact, planes)
        res['roofline'] = roof
        eroof, eprof = price_dominant_kernel(prof_eval, psteps, pmc_eval, src, exact and not bf16, planes)
        eroof['eval_bytes_per_session'] = algorithmic_bytes_per_session(w['flags'], cinfo, w['batch'], False)
        eroof['gather_frac'] = round(eroof['eval_bytes_per_session'] * res['eval_sessions_per_s'] / world / HBM_PEAK, 6)
        res['eval_roofline'] = eroof
        res['eval_kernel_ms_per_step'] = {k: round(v['ms'] / psteps, 4) for k, v in sorted(eprof.items(), key=lambda kv: -kv[1]['ms'])[:8]}
        res['kernel_ms_per_step'] = {k: round(v['ms'] / psteps, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]['ms'])[:16]}
        res['kernel_rate'] = {k: ('%.1f TF/s' % (v['flops'] / v['ms'] / 1e9) if v['flops'] > 0 else '%.0f GB/s' % (v['bytes'] / v['ms'] / 1e6))
                              for k, v in sorted(prof.items(), key=lambda kv: -kv[1]['ms'])[:16] if v['flops'] > 0 or v['bytes'] > 0}
        if a.shapes:
            res['gemm_shapes'] = {k: '%.3f ms/step, %d launches/step, %.1f us, %.1f TF/s, %.0f GB/s' % (v['ms'] / psteps, v['launches'] // psteps, 1e3 * v['ms'] / v['launches'],
                                                                                                      v['flops'] / v['ms'] / 1e9, v['bytes'] / v['ms'] / 1e6)
                                  for k, v in sorted(prof_shapes.items(), key=lambda kv: -kv[1]['ms']) if '[' in k and v['ms'] / psteps > 0.01 and v['flops'] > 0}
            res['eval_shapes'] = {k: '%.4f ms/step, %d launches/step, %.1f us, %.0f GB/s' % (v['ms'] / psteps, v['launches'] // psteps, 1e3 * v['ms'] / max(1, v['launches']),
                                                                                             v['bytes'] / max(v['ms'], 1e-9) / 1e6)
                                  for k, v in sorted(prof_eval.items(), key=lambda kv: -kv[1]['ms']) if v['ms'] / psteps > 0.004}
            res['kernel_table'] = {k: '%d launches/step, %.4f ms/step, avg %.1f us' % (v['launches'] // psteps, v['ms'] / psteps, 1e3 * v['ms'] / max(1, v['launches']))
                                   for k, v in sorted(prof.items(), key=lambda kv: -kv[1]['ms'])}
        res['kernel_launches_per_step'] = round(sum(v['launches'] for v in prof.values()) / psteps, 1)
        # the STEP against its own floor: algorithmic bytes (SURVEY 8-d per-session figure x sessions + the optimizer's streams) and the HBM bytes the
        # PMC passes counted for a step, both over the measured step time; fp32-equivalent flops of every matrix product of the step
        n_tab = model.iid_embeddings.weight.numel()
        n_all = sum(p.numel() for p in model.parameters())
        adam_b = (24.0 * n_tab if eng._lazy is None else 0.0) + 32.0 * (n_all - n_tab)
        alg_b = float(bytes_train) * B + adam_b
        step_s = el / max(1, a.steps)
        sr = {'algorithmic_GB_per_step': round(alg_b / 1e9, 3), 'pmc_GB_per_step': None, 'traffic_ratio': None,
              'hbm_frac_algorithmic': round(alg_b / step_s / HBM_PEAK, 4), 'hbm_frac_pmc': None,
              'fp32_equivalent_TFLOPs': round(sum(v['flops'] for v in prof.values()) / psteps / step_s / 1e12, 1),
              'note': 'algorithmic = %d B/session x %d sessions + %.2f GB of Adam streams (dense table sweep: 24 B/parameter, other parameters 32)' % (bytes_train, B, adam_b / 1e9)}
        try:
            if exact and world == 1 and 'traffic_stale' not in roof:
                jj = json.load(open(os.path.join(ROOT, src.split(' ')[0])))
                sr['pmc_GB_per_step'] = jj['hbm_GB_per_train_step']
                sr['traffic_ratio'] = round(jj['hbm_GB_per_train_step'] * 1e9 / alg_b, 2)
                sr['hbm_frac_pmc'] = round(jj['hbm_GB_per_train_step'] * 1e9 / step_s / HBM_PEAK, 4)
                sr['pmc_source'] = src.split(' ')[0]
        except Exception:
            pass
        res['step_roofline'] = sr
        res['eval_kernel_launches_per_step'] = round(sum(v['launches'] for v in eprof.values()) / psteps, 1)
    if world == 1 and not bf16 and not a.no_bf16_line:
        # the same workload in the bf16 arithmetic mode (BASELINE.json configs[1] names it), measured in the same run on the same
        # resident batches: a second model + engine (own fp32 master weights and Adam state), same timing brackets
        # (the fp32 model, its optimizer state and workspace are released first: two resident 1 GB tables + workspaces cost
        # the second model 3-8 % -- measured)
        del eng, model
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        args_bf = synth.make_args(a.workload, dev, **dict(over, dtype='bf16'))
        torch.manual_seed(0)
        model_bf = IntEL(args_bf, corpus).to(dev)
        eng_bf = IntELEngine(model_bf, a.loss, args_bf, lr=lr, l2=l2, lazy_table=lazy)
        eng_bf.defer_table_wait = not a.no_defer_table
        for i in range(a.warmup):
            eng_bf.train_step(batches[i % nbatches])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.steps):
            lb = eng_bf.train_step(batches[i % nbatches])
        eng_bf.flush()
        torch.cuda.synchronize()
        el_bf = time.perf_counter() - t0
        model_bf.eval()
        for i in range(2):
            eng_bf.eval_step(batches[i % nbatches], k=3)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(max(1, ev_steps)):
            _, nd_bf = eng_bf.eval_step(batches[i % nbatches], k=3)
        torch.cuda.synchronize()
        ev_bf = time.perf_counter() - t1
        bf_roof = None
        if not a.no_roofline:
            lib = _lib.lib()
            lib.intel_set_concurrency(model_bf._context(), 0)
            ovb, eng_bf.overlap_table_update = eng_bf.overlap_table_update, False
            model_bf.train()
            lib.intel_prof_enable(1)
            for i in range(3):
                eng_bf.train_step(batches[i % nbatches])
            pbf = json.loads(lib.intel_prof_collect().decode())
            lib.intel_prof_enable(0)
            eng_bf.overlap_table_update = ovb
            lib.intel_set_concurrency(model_bf._context(), 1)
            pmc_bf, src_bf = None, None
            try:
                import glob
                src_bf = sorted(os.path.relpath(f, ROOT) for f in glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_pmc_traffic_bf16.json')))[-1]
                pmc_bf = json.load(open(os.path.join(ROOT, src_bf))).get('kernels')
            except Exception:
                pass
            bf_roof, bf_prof = price_dominant_kernel(pbf, 3, pmc_bf, src_bf, a.workload == 'tmall' and B == 4096, 1.0)
            bf_roof['kernel_ms_per_step'] = {k: round(v['ms'] / 3, 4) for k, v in sorted(bf_prof.items(), key=lambda kv: -kv[1]['ms'])[:8]}
        res['bf16_mode'] = {'value': round(B * a.steps / max(el_bf, 1e-9), 1), 'unit': 'sessions/s', 'ms_per_step': round(1e3 * el_bf / max(1, a.steps), 4),
                            'eval_sessions_per_s': round(B * max(1, ev_steps) / ev_bf, 1), 'ndcg3_random_init': round(float(nd_bf.float().nan_to_num(0).mean()), 5),
                            'loss_last_step': round(float(lb[0]), 6) if a.steps else None,
                            'roofline': bf_roof,
                            'note': 'python bench.py --dtype bf16 gives this mode its own full line (kernel table, eval roofline); not `value`'}
        del eng_bf, model_bf
        gc.collect()
        torch.cuda.empty_cache()
    if world == 1 and not bf16 and a.workload == 'tmall' and not a.no_workloads:
        # the other workloads of BASELINE.json / the published scripts, each as a short timed block in THIS process, so that their numbers are
        # driver-timed too (own full lines: python bench.py --workload W [--batch B]); never part of `value`
        res['workloads'] = [short_line(wl, bb, oo, min(20, max(1, a.steps)), min(5, a.warmup), dev)
                            for wl, bb, oo in (('tmall_pub', 512, {}), ('tmall_pub_mse', 512, {'loss': 'IntMSEloss'}), ('lifedata', 2048, {}), ('stress', 256, {}))]
    if world == 1 and not a.no_cpu_baseline:
        res['cpu_baseline'] = cpu_baseline(args_ns, corpus, cinfo, a.workload, a.loss, a.cpu_budget)
    print(json.dumps(res))


if __name__ == '__main__':
    main()
