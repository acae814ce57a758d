#!/usr/bin/env python3
"""Copy the summaries of tools/round_check.sh (gpurun_out/round_check, gpurun_out/prof_r2) into profiles/ and refresh
profiles/pmc_traffic.json (means per dispatch of the FETCH_SIZE / WRITE_SIZE passes, stamped with the commit)."""
import csv, glob, json, os, shutil, subprocess
root = 'gpurun_out/prof_r2'

def mean(sub, counter, key):
    v = []
    for p in glob.glob(root + '/' + sub + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(p)):
            if r['Counter_Name'] == counter and key in r['Kernel_Name']:
                v.append(float(r['Counter_Value']))
    return sum(v) / len(v), len(v)

head = subprocess.check_output(['git', 'rev-parse', '--short', 'HEAD']).decode().strip()
d = json.load(open('profiles/pmc_traffic.json'))
d['commit'] = head
for name, key in (("fused_render_kernel:fp16x3+fp16mx", "fused_render_kernel<0, 2>"), ("fused_render_kernel:fp16x3", "fused_render_kernel<0, 0>"),
                  ("styled_rays_kernel:fp16x3", "styled_rays_kernel")):
    f, n = mean('fetch', 'FETCH_SIZE', key)
    w, _ = mean('write', 'WRITE_SIZE', key)
    d['kernels'][name]['fetch_bytes'], d['kernels'][name]['write_bytes'] = f * 1024, w * 1024
    d['kernels'][name]['source'] = d['kernels'][name]['source'].rsplit('n=', 1)[0] + 'n=%d dispatches' % n
    print(name, round(f * 1024 / 1e9, 3), 'GB fetched', round(w * 1024 / 1e9, 4), 'GB written', n)
json.dump(d, open('profiles/pmc_traffic.json', 'w'), indent=1)
pairs = [('gpurun_out/round_check/bench_default.json', 'profiles/r2_bench_default_b.json'), (root + '/summary.txt', 'profiles/r2_default_bench_rocprof_summary.txt'),
         (root + '/bench_stats.json', 'profiles/r2_default_bench_under_rocprof.json'),
         ('gpurun_out/round_check/bench_2rank_frames.json', 'profiles/r2_bench_2rank_gloo_frames.json'),
         ('gpurun_out/round_check/bench_2rank_rays.json', 'profiles/r2_bench_2rank_gloo_rays.json'),
         ('gpurun_out/round_check/pytest_gpu.txt', 'profiles/r2_pytest_gpu.txt')]
for a, b in pairs:
    shutil.copy(a, b)
stats = sorted(glob.glob(root + '/stats/**/*kernel_stats.csv', recursive=True), key=os.path.getmtime)[-1]
shutil.copy(stats, 'profiles/r2_default_bench_kernel_stats.csv')
b = json.loads(open('gpurun_out/round_check/bench_default.json').read().strip().splitlines()[-1])
print('headline', round(b['value']), b['unit'], round(b['ms_per_step'], 2), 'ms/step frac', round(b['roofline']['frac'], 4))
print({k: (round(v['value'], 2), v.get('unit')) for k, v in b['configs'].items()})
print({k: round(v['value']) for k, v in b['alt_precisions'].items()} if isinstance(b['alt_precisions'], dict) else [round(a['value']) for a in b['alt_precisions']])
