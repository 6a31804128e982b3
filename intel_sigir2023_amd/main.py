"""CLI counterpart of the reference's ``main.py`` for the hot path:
``python -m intel_sigir2023_amd.main --model_name IntEL --loss_name IntBPRloss --workload tmall ...``

Classes are resolved by name like the reference (main.py:127-130); flags come from the same
``parse_*_args`` hooks.  Data: synthetic workloads (synth.py) by default; ``--dataset NAME --datapath DIR``
reads the reference's CSV / JSON corpus (data.SeqReader), flattens it into HBM (feed.ColumnarStore) and assembles
every batch on the device.  Needs an MI355X; there is no CPU path.
"""
import argparse
import logging
import os
import pickle
import sys

import numpy as np
import torch

from . import loss as loss_mod
from . import data as data_mod
from . import feed
from . import parallel
from . import synth
from .model import IntEL, aWELv_IntEL
from .runner import BaseRunner

MODELS = {'IntEL': IntEL, 'aWELv_IntEL': aWELv_IntEL}
LOSSES = {n: getattr(loss_mod, n) for n in ('BPRloss', 'Listloss', 'MSEloss', 'IntBPRloss', 'IntListloss', 'IntMSEloss')}
RUNNERS = {'BaseRunner': BaseRunner}


def parse_global_args(parser):
    parser.add_argument('--gpu', type=str, default='0', help='device index')
    parser.add_argument('--verbose', type=int, default=logging.INFO)
    parser.add_argument('--random_seed', type=int, default=0)
    parser.add_argument('--train', type=int, default=1)
    parser.add_argument('--workload', type=str, default='tiny', help='synthetic workload: ' + ', '.join(synth.WORKLOADS))
    parser.add_argument('--train_batches', type=int, default=8, help='synthetic batches per epoch')
    parser.add_argument('--use_engine', type=int, default=1)
    parser.add_argument('--dataset', type=str, default='', help='corpus directory name under --datapath (empty: synthetic workload)')
    parser.add_argument('--datapath', type=str, default='../data/')
    parser.add_argument('--sep', type=str, default='\t')
    parser.add_argument('--intent_note', type=str, default='')
    parser.add_argument('--max_session_len', type=int, default=40)
    parser.add_argument('--regenerate', type=int, default=0, help='Whether to regenerate intermediate files (the corpus cache).')
    return parser


def load_corpus(args, reader_cls, reader_name='SeqReader'):
    """main.py:64-72 of the reference: the parsed corpus is cached as ``<datapath>/<dataset>/<reader>_<max_session_len><note>.pkl``
    and re-read unless --regenerate.  Data parallel: rank 0 parses and writes, the other ranks wait and load."""
    path = os.path.join(args.datapath, args.dataset, '%s_%d%s.pkl' % (reader_name, args.max_session_len, args.intent_note))
    rank = parallel.rank()
    corpus = None
    if rank == 0:
        if not args.regenerate and os.path.exists(path):
            logging.info('Load corpus from {}'.format(path))
            with open(path, 'rb') as fh:
                corpus = pickle.load(fh)
            corpus.pos_types = ['c_paynum_i', 'c_favnum_i', 'c_clicknum_i']
        else:
            corpus = reader_cls(args)
            logging.info('Save corpus to {}'.format(path))
            tmp = path + '.tmp%d' % os.getpid()
            try:
                with open(tmp, 'wb') as fh:
                    pickle.dump(corpus, fh)
                os.replace(tmp, path)
            except OSError as e:          # a read-only data directory must not stop the run
                logging.info('corpus cache not written (%s)' % e)
    parallel.barrier()
    if corpus is None:
        if os.path.exists(path):
            with open(path, 'rb') as fh:
                corpus = pickle.load(fh)
            corpus.pos_types = ['c_paynum_i', 'c_favnum_i', 'c_clicknum_i']
        else:
            corpus = reader_cls(args)
    return corpus


def build_parser(argv=None):
    init = argparse.ArgumentParser(add_help=False)
    init.add_argument('--model_name', type=str, default='IntEL')
    init.add_argument('--loss_name', type=str, default='IntBPRloss')
    init.add_argument('--runner_name', type=str, default='BaseRunner')
    init_args, _ = init.parse_known_args(argv)
    for name, table in ((init_args.model_name, MODELS), (init_args.loss_name, LOSSES), (init_args.runner_name, RUNNERS)):
        if name not in table:
            raise SystemExit('unknown class %s (available: %s)' % (name, ', '.join(table)))
    parser = argparse.ArgumentParser(description='IntEL on MI355X')
    parse_global_args(parser)
    MODELS[init_args.model_name].parse_model_args(parser)
    RUNNERS[init_args.runner_name].parse_runner_args(parser)
    LOSSES[init_args.loss_name].parse_loss_args(parser)
    return init_args, parser


def main(argv=None):
    init_args, parser = build_parser(argv)
    w0, _ = parser.parse_known_args(argv)
    parser.set_defaults(**{k: v for k, v in synth.WORKLOADS[w0.workload]['flags'].items()})
    args, extras = parser.parse_known_args(argv)
    logging.basicConfig(level=args.verbose, stream=sys.stdout)
    np.random.seed(args.random_seed)
    torch.manual_seed(args.random_seed)
    if not torch.cuda.is_available():
        raise SystemExit('intel_sigir2023_amd needs an MI355X (no CPU path)')
    # data parallel (torchrun -m intel_sigir2023_amd.main ...): one process per GPU over RCCL; every global batch of
    # --batch_size sessions is split contiguously over the ranks, so N ranks follow the 1-rank trajectory
    rank, world, local_rank = parallel.init_distributed()
    args.device = torch.device('cuda', local_rank if world > 1 else int(args.gpu or 0))
    torch.cuda.set_device(args.device)
    if world > 1 and rank != 0:
        logging.getLogger().setLevel(logging.WARNING)
    if args.dataset:
        corpus = load_corpus(args, data_mod.SeqReader)
    else:
        corpus, _ = synth.make_corpus(args.workload)
    model = MODELS[init_args.model_name](args, corpus).to(args.device)
    if world > 1:         # identical replicas: rank 0's initialisation everywhere
        parallel.broadcast_([p.data for p in model.parameters()])
        model.invalidate_packed()         # raw writes do not move torch's version counters (model._params_key)
        # every rank's loss is the mean over ITS shard: a training batch must split evenly (a ragged last batch is trimmed to a
        # multiple of the world size); evaluation sets keep every session (uneven shards, weighted reductions)
        if args.batch_size % world:
            raise SystemExit('--batch_size %d is the GLOBAL batch and must be a multiple of the %d ranks' % (args.batch_size, world))
    logging.info('#params: %d' % model.count_variables())
    criterion = LOSSES[init_args.loss_name](args)
    runner = RUNNERS[init_args.runner_name](args, use_engine=bool(args.use_engine))
    if args.dataset:
        stores = {p: feed.ColumnarStore(corpus, p, model.model_num, model.intent_num, model.max_his).to(args.device)
                  for p in ('train', 'dev', 'test')}
        logging.info('columnar corpus in HBM: %s' % ', '.join('%s %d sessions / %.1f MB' % (p, s.n_sessions, s.nbytes() / 1e6)
                                                               for p, s in stores.items()))
        fixed = lambda p: list(feed.epoch_batches(stores[p], args.eval_batch_size, seed=args.random_seed + 1, shuffle_sessions=False,
                                                  rank=rank, world=world, keep_all=True))      # evaluation never drops a session
        data = {'train': lambda ep: feed.epoch_batches(stores['train'], args.batch_size, epoch=ep, seed=args.random_seed, rank=rank, world=world),
                'dev': fixed('dev'), 'test': fixed('test')}
    else:
        # synthetic workloads: every rank generates the same global batch (same seed) and keeps its contiguous shard
        mk = lambda n, seed: parallel.shard_batch(synth.make_batch(args.workload, n, args.device, seed=seed, ragged=True), rank, world)
        if args.eval_batch_size % world:
            raise SystemExit('--eval_batch_size %d must be a multiple of the %d ranks for the synthetic workloads' % (args.eval_batch_size, world))
        dev = [mk(args.eval_batch_size, 10_000 + i) for i in range(2)]
        data = {'train': lambda ep: [mk(args.batch_size, ep * 1000 + i) for i in range(args.train_batches)], 'dev': dev, 'test': dev}
    dev_curve = runner.train(model, data, criterion, init_args.loss_name) if args.train > 0 else []
    loss, res = runner.evaluate(model, data['test'], runner.topk, runner.metrics, criterion)
    logging.info('test loss= %.4f, metrics: %s' % (loss, ', '.join('%s:%.4f' % kv for kv in sorted(res.items()))))
    main.last_run = {'dev_main_metric': list(dev_curve), 'train_losses': list(getattr(runner, 'train_losses', [])), 'test': res}
    return res


if __name__ == '__main__':
    main()
