#!/usr/bin/env python3
"""Copy a round's profile summaries from gpurun_out/prof_<tag>/ (written on the GPU box by tools/profile_round.sh) into
profiles/ under <tag>_* names and write profiles/<tag>_STAMP.json: git HEAD and the hashes of the kernel sources every file of
the round was measured on, so that a stale summary can be told from a current one.
    python tools/collect_profiles.py [tag, default r04]"""
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else 'r04'
    src = os.path.join(ROOT, 'gpurun_out', f'prof_{tag}')
    dst = os.path.join(ROOT, 'profiles')
    from bench import conv_source_stamp
    from lct_vqa_amd import build
    names = {'kt_f32': 'kernel_stats_bench_f32_b256', 'kt_bf16': 'kernel_stats_bench_bf16_b256', 'kt_lct': 'kernel_stats_bench_lct_b256',
             'kt_lct_bf16': 'kernel_stats_bench_lct_bf16_b256', 'kt_darts_enc': 'kernel_stats_darts_encoder_b16',
             'kt_darts_vqa': 'kernel_stats_darts_vqa_b64', 'kt_resnet_train_bf16': 'kernel_stats_resnet_train_bf16_b256',
             'kt_lct_train_bf16': 'kernel_stats_lct_train_bf16_b256', 'kt_darts_vqa_bf16': 'kernel_stats_darts_vqa_bf16_b64',
             'kt_lct_bf16_1s': 'kernel_stats_bench_lct_bf16_one_stream_b256'}
    copied = []
    for d, n in names.items():
        f = glob.glob(os.path.join(src, d, '**', '*kernel_stats.csv'), recursive=True)
        if f:
            shutil.copy(f[0], os.path.join(dst, f'{tag}_{n}.csv'))
            copied.append(f'{tag}_{n}.csv')
    for dt, n in (('f32', 'wino'), ('bf16', 'bf16')):
        f = os.path.join(src, f'pmc_conv_{dt}.json')
        if os.path.exists(f):
            shutil.copy(f, os.path.join(dst, f'{tag}_pmc_conv_{n}.json'))
            copied.append(f'{tag}_pmc_conv_{n}.json')
        f = os.path.join(src, f'bench_{dt}_under_rocprof.json')
        if os.path.exists(f) and os.path.getsize(f):
            shutil.copy(f, os.path.join(dst, f'{tag}_bench_{dt}_under_rocprof.json'))
            copied.append(f'{tag}_bench_{dt}_under_rocprof.json')
    for f, n in (('kt_lct.json', 'bench_lct_under_rocprof.json'), ('kt_lct_bf16.json', 'bench_lct_bf16_under_rocprof.json'),
                 ('kt_darts_vqa.json', 'bench_darts_vqa_under_rocprof.json'), ('kt_darts_enc.txt', 'darts_encoder_b16.txt'),
                 ('kt_darts_vqa_bf16.json', 'bench_darts_vqa_bf16_under_rocprof.json'), ('kt_lct_bf16_1s.json', 'bench_lct_bf16_one_stream_under_rocprof.json'),
                 ('kt_resnet_train_bf16.json', 'bench_resnet_train_bf16_under_rocprof.json'), ('kt_lct_train_bf16.json', 'bench_lct_train_bf16_under_rocprof.json'),
                 ('pmc_resnet_train_bf16.json', 'pmc_resnet_train_bf16.json')):
        p = os.path.join(src, f)
        if os.path.exists(p) and os.path.getsize(p):
            shutil.copy(p, os.path.join(dst, f'{tag}_{n}'))
            copied.append(f'{tag}_{n}')
    head = subprocess.run(['git', 'rev-parse', 'HEAD'], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    dirty = bool(subprocess.run(['git', 'status', '--porcelain', '--', 'lct_vqa_amd/csrc', 'bench.py'], cwd=ROOT, capture_output=True,
                                text=True).stdout.strip())
    stamp = {'round': tag, 'git_head_at_collection': head, 'kernel_sources_modified_since_head': dirty,
             'csrc_stamp_all_sources': build._stamp(), 'conv_source_stamp': conv_source_stamp(), 'files': sorted(copied),
             'note': 'every file listed was produced by ONE run of tools/profile_round.sh on one MI355X box from the sources these hashes name'}
    json.dump(stamp, open(os.path.join(dst, f'{tag}_STAMP.json'), 'w'), indent=1)
    print(json.dumps(stamp, indent=1))


if __name__ == '__main__':
    main()
