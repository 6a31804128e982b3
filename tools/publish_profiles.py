#!/usr/bin/env python3
"""Copy the summaries `tools/collect_profiles.sh <tag>` left under gpurun_out/<tag>/ into profiles/ under their round names
(bench lines pretty-printed, kernel-stats CSVs, PMC traffic summaries, event timelines, the batch sweep as one array).
usage: python tools/publish_profiles.py <tag> [round=r03]"""
import glob
import json
import os
import shutil
import subprocess
import sys

tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else 'r06'
src = os.path.join('gpurun_out', tag)
dst = 'profiles'


def bench(name, out):
    f = os.path.join(src, name)
    if not os.path.exists(f):
        print('missing', f)
        return None
    d = json.load(open(f))
    json.dump(d, open(os.path.join(dst, out), 'w'), indent=1)
    return d


names = {'bench.json': 'bench_tmall', 'bench_bf16.json': 'bench_tmall_bf16', 'bench_lifedata.json': 'bench_lifedata', 'bench_stress.json': 'bench_stress',
         'bench_stress_b1024.json': 'bench_stress_b1024', 'bench_pl_div.json': 'bench_pl_div', 'bench_gru4rec.json': 'bench_gru4rec',
         'bench_tmall_pub.json': 'bench_tmall_pub', 'bench_gru4rec_steps.json': 'bench_gru4rec_per_step', 'bench_tmall_pub_steps.json': 'bench_tmall_pub_per_step', 'bench_stress_dense.json': 'bench_stress_dense_adam', 'bench_stress_b1024_dense.json': 'bench_stress_b1024_dense_adam',
         'bench_tmall_b1024_dense.json': 'bench_tmall_b1024_dense_adam', 'bench_tmall_b1024.json': 'bench_tmall_b1024', 'bench_tmall_lazy.json': 'bench_tmall_lazy_adam', 'bench_zipf.json': 'bench_zipf', 'bench_unfused.json': 'bench_unfused', 'bench_phased.json': 'bench_phased',
         'bench_enc_unfused.json': 'bench_enc_unfused', 'bench_enc_unfused_bwd.json': 'bench_enc_unfused_bwd', 'bench_lifedata_b4096.json': 'bench_lifedata_b4096',
         'bench_tmall_pub_long.json': 'bench_tmall_pub_300steps', 'bench_tmall_pub_mse.json': 'bench_tmall_pub_mse', 'bench_tmall_pub_mse_kernel_per_op.json': 'bench_tmall_pub_mse_kernel_per_op'}
for k, v in names.items():
    bench(k, '%s_%s.json' % (rnd, v))
sweep = []
for f in sorted(glob.glob(os.path.join(src, 'bench_b[0-9]*.json')), key=lambda p: int(os.path.basename(p)[7:-5])):
    d = json.load(open(f))
    sweep.append({'per_gpu_batch': d['config']['per_gpu_batch'], 'value': d['value'], 'ms_per_step': d['ms_per_step'],
                  'eval_sessions_per_s': d['eval_sessions_per_s'], 'bf16_mode': d.get('bf16_mode', {}).get('value')})
if sweep:
    json.dump(sweep, open(os.path.join(dst, '%s_batch_sweep.json' % rnd), 'w'), indent=1)


def one(pattern):
    fs = glob.glob(os.path.join(src, pattern), recursive=True)
    if len(fs) > 1:      # rocprofv3 names its files by process id: a second collection merged into the same scratch directory leaves BOTH runs' files side by side
        raise SystemExit('%d files match %s: stale files of an earlier collection -- rm -rf %s and collect again' % (len(fs), pattern, src))
    return fs[0] if fs else None


for sub, out in (('stats1s', 'bench_tmall_kernel_stats.csv'), ('stats', 'bench_tmall_kernel_stats_concurrent.csv'), ('stats1s_pub', 'bench_tmall_pub_kernel_stats.csv'), ('stats1s_pub_mse', 'bench_tmall_pub_mse_kernel_stats.csv'),
                 ('stats_eval', 'bench_tmall_eval_kernel_stats.csv'), ('stats1s_bf16', 'bench_tmall_bf16_kernel_stats.csv'),
                 ('stats1s_fusedbwd', 'bench_tmall_fused_tower_bwd_forced_kernel_stats.csv')):
    f = one('%s/**/*kernel_stats.csv' % sub)
    if f:
        shutil.copy(f, os.path.join(dst, '%s_%s' % (rnd, out)))
    else:
        print('missing', sub)
f, w, fe, we = (one('%s/**/*counter_collection.csv' % s) for s in ('fetch', 'write', 'fetch_eval', 'write_eval'))
if f and w:
    cmd = [sys.executable, 'tools/pmc_summary.py', f, w, '4', os.path.join(dst, '%s_pmc_traffic.json' % rnd)]
    if fe and we:
        cmd += [fe, we, '4']
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL)
fb, wb = one('fetch_bf16/**/*counter_collection.csv'), one('write_bf16/**/*counter_collection.csv')
if fb and wb:
    subprocess.run([sys.executable, 'tools/pmc_summary.py', fb, wb, '4', os.path.join(dst, '%s_pmc_traffic_bf16.json' % rnd)], check=True, stdout=subprocess.DEVNULL)
fl, wl = one('fetch_lazy/**/*counter_collection.csv'), one('write_lazy/**/*counter_collection.csv')
if fl and wl:
    subprocess.run([sys.executable, 'tools/pmc_summary.py', fl, wl, '4', os.path.join(dst, '%s_pmc_traffic_lazy_adam.json' % rnd)], check=True, stdout=subprocess.DEVNULL)
ff, wf = one('fetch_fusedbwd/**/*counter_collection.csv'), one('write_fusedbwd/**/*counter_collection.csv')
if ff and wf:
    subprocess.run([sys.executable, 'tools/pmc_summary.py', ff, wf, '4', os.path.join(dst, '%s_pmc_traffic_fused_tower_bwd_forced.json' % rnd)], check=True, stdout=subprocess.DEVNULL)
for name in ('ab_pair_bwd.txt', 'ab_table_after_flush.txt', 'ab_tower_gather.txt', 'pair_bench.txt', 'pair_pmc.txt', 'qkv_pmc.txt', 'rocprof_timeline_f32_train.txt', 'rocprof_timeline_f32_train_nopair.txt', 'ab_tower_bwd_bf16.txt', 'ab_tower_bwd_f32.txt', 'ab_defer_table.txt', 'gpu_bound_pub.txt', 'gpu_bound_tmall.txt', 'ab_pub.txt', 'ab_mse_enc32.txt', 'ab_attn_p3_lifedata.txt', 'ab_attn_p3_stress.txt', 'attn_bench_long.txt'):
    f = os.path.join(src, name)
    if os.path.exists(f):
        shutil.copy(f, os.path.join(dst, '%s_%s' % (rnd, name)))
for t in ('f32_train_fusedbwd', 'f32_train', 'bf16_train', 'f32_eval', 'pub_f32_train', 'pub_f32_eval', 'pub_mse_f32_train'):
    f = os.path.join(src, 'timeline_%s.txt' % t)
    if os.path.exists(f):
        shutil.copy(f, os.path.join(dst, '%s_timeline_%s.txt' % (rnd, t)))
for f in sorted(os.listdir(dst)):
    if f.startswith(rnd):
        print(f)
