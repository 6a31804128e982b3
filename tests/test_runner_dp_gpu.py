"""Data parallelism through the CLI / runner (torchrun -m intel_sigir2023_amd.main): two ranks (gloo, both on the one test
GPU) train the CSV mini corpus through the device feed -- every global batch split contiguously over the ranks, padded to the
GLOBAL batch shape, BPR tie-breaks keyed by the global session index, dev metrics gathered over ranks, model saved by rank
0 -- and must reproduce the single-process run: the per-epoch training loss and the dev NDCG@3 trajectory
(helpers/BaseRunner.py:190-266)."""
import json
import os
import shutil
import socket
import tempfile

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(rank, world, port, root, out_dir, extra, batch='16', eval_batch='6'):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), INTEL_DIST_BACKEND='gloo', INTEL_SINGLE_DEVICE='1')
    from intel_sigir2023_amd import main as cli
    from intel_sigir2023_amd import parallel
    model_path = os.path.join(out_dir, 'w%d' % world, 'model.pt')
    cli.main(['--model_name', 'IntEL', '--loss_name', 'IntBPRloss', '--workload', 'tiny', '--dataset', 'minidata', '--datapath', root,
              '--intent_note', '_multi', '--max_session_len', '100', '--model_num', '3', '--epoch', '4', '--batch_size', batch,
              '--eval_batch_size', eval_batch, '--topk', '3,1,5', '--main_metric', 'NDCG@3', '--lr', '2e-3', '--l2', '1e-5',
              '--model_path', model_path, '--random_seed', '5'] + list(extra))
    run = cli.main.last_run
    sd = torch.load(model_path, map_location='cpu')
    with open(os.path.join(out_dir, 'w%d_r%d.json' % (world, rank)), 'w') as fh:
        json.dump({'dev': run['dev_main_metric'], 'loss': run['train_losses'], 'test': run['test'],
                   'checksum': float(sum(v.double().abs().sum() for v in sd.values()))}, fh)
    if torch.distributed.is_initialized():
        parallel.barrier()
        torch.distributed.destroy_process_group()


@pytest.mark.parametrize('extra', [(), ('--decay_lr', '0.5', '--decay_step', '2', '--test_epoch', '1')])
def test_two_rank_cli_training_follows_the_single_process_trajectory(extra):
    assert torch.cuda.is_available()
    with tempfile.TemporaryDirectory() as d:
        shutil.copytree(os.path.join(HERE, 'golden', 'minidata'), os.path.join(d, 'minidata'))
        root = d + os.sep
        mp.spawn(_run, args=(1, _free_port(), root, d, extra), nprocs=1, join=True)
        mp.spawn(_run, args=(2, _free_port(), root, d, extra), nprocs=2, join=True)
        one = json.load(open(os.path.join(d, 'w1_r0.json')))
        two = [json.load(open(os.path.join(d, 'w2_r%d.json' % r))) for r in range(2)]
    assert len(one['dev']) == 4 and len(one['loss']) == 4
    assert two[0]['dev'] == two[1]['dev'] and two[0]['loss'] == two[1]['loss']      # every rank computes the global numbers
    for e in range(4):
        assert abs(two[0]['loss'][e] - one['loss'][e]) < 2e-4, (e, one['loss'], two[0]['loss'])
        assert abs(two[0]['dev'][e] - one['dev'][e]) < 1e-3, (e, one['dev'], two[0]['dev'])
    for k, v in one['test'].items():
        assert abs(two[0]['test'][k] - v) < 1e-3, k
    assert abs(two[0]['checksum'] - one['checksum']) < 1e-3 * max(1.0, abs(one['checksum']))


def test_three_ranks_evaluate_every_session_of_uneven_batches():
    """Evaluation sets keep EVERY session under data parallelism: evaluation batches of 7 sessions split 2 / 2 / 3 over three ranks
    (and a ragged last batch with fewer sessions than ranks leaves a rank a zero-weight placeholder), losses and metrics reduced
    with the shard sizes as weights -- the dev / test numbers equal the single-process run's, which sees the same sessions."""
    assert torch.cuda.is_available()
    with tempfile.TemporaryDirectory() as d:
        for n in (1, 3):
            shutil.copytree(os.path.join(HERE, 'golden', 'minidata'), os.path.join(d, 'data%d' % n, 'minidata'))
        # the two runs side by side, each on its own copy of the corpus
        ctxs = [mp.spawn(_run, args=(n, _free_port(), os.path.join(d, 'data%d' % n) + os.sep, d, (), '18', '7'), nprocs=n, join=False) for n in (1, 3)]
        for c in ctxs:
            while not c.join():
                pass
        one = json.load(open(os.path.join(d, 'w1_r0.json')))
        three = [json.load(open(os.path.join(d, 'w3_r%d.json' % r))) for r in range(3)]
    assert three[0]['dev'] == three[1]['dev'] == three[2]['dev']
    for e in range(4):
        assert abs(three[0]['dev'][e] - one['dev'][e]) < 1e-3, (e, one['dev'], three[0]['dev'])
    for k, v in one['test'].items():
        assert abs(three[0]['test'][k] - v) < 1e-3, k
