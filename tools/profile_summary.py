"""Turn the raw outputs of tools/profile.sh (gpurun_out/<tag>_<cfg>_*) into the committed summaries
profiles/<tag>_<cfg>_kernel_stats.csv and profiles/<tag>_<cfg>_pmc_summary.json.    usage: profile_summary.py <tag> <cfg> [--pmc-only]
(--pmc-only: counter passes of tools/profile_pmc_only.sh, <cfg> = e.g. cfg2_S16384)"""
import collections
import csv
import glob
import json
import os
import statistics
import sys

tag, cfg = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, 'gpurun_out')
sys.path.insert(0, root)


def is_opt(name):
    return ('k_cp' in name or 'cpf_jit' in name) and 'initial' not in name and 'k_reg_rows' not in name


if '--pmc-only' not in sys.argv:
    bench = json.loads(open(os.path.join(out, f'{tag}_{cfg}_bench.json')).read().strip().splitlines()[-1])
    prof = json.loads(open(os.path.join(out, f'{tag}_{cfg}_bench_under_rocprof.json')).read().strip().splitlines()[-1])
    steps, warm = prof['steps'], prof['warmup']
    rows = [r for r in csv.DictReader(open(os.path.join(out, f'{tag}_{cfg}_kernel_trace.csv'))) if is_opt(r['Kernel_Name'])]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    dur = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6 for r in rows]
    name = rows[warm]['Kernel_Name'].split('(')[0] if rows else ''
    rows = [r for r in rows if r['Kernel_Name'].split('(')[0] == name]          # the headline kernel only (no other configuration's launches)
    dur = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6 for r in rows]
    timed = dur[warm:warm + steps]          # bench.py: warm-up launches, the timed launches, one more launch for the summary
    os.makedirs(os.path.join(root, 'profiles'), exist_ok=True)
    with open(os.path.join(root, 'profiles', f'{tag}_{cfg}_kernel_stats.csv'), 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['kernel', 'launches_all', 'launches_timed', 'mean_ms_timed', 'median_ms_timed', 'min_ms_timed', 'max_ms_timed', 'mean_ms_all',
                    'bench_kernel_ms_hip_events_under_rocprof', 'bench_kernel_ms_hip_events_unprofiled', 'VGPR', 'SGPR', 'LDS_bytes', 'workgroup', 'grid'])
        r0 = rows[warm]
        ev = lambda b: float(b['roofline']['note'].split('kernel ')[1].split(' ms')[0])  # noqa: E731
        w.writerow([name, len(dur), len(timed), f'{statistics.mean(timed):.4f}', f'{statistics.median(timed):.4f}', f'{min(timed):.4f}', f'{max(timed):.4f}',
                    f'{statistics.mean(dur):.4f}', f'{ev(prof):.4f}', f'{ev(bench):.4f}', r0.get('VGPR_Count', ''), r0.get('SGPR_Count', ''),
                    r0.get('LDS_Block_Size', ''), r0.get('Workgroup_Size', r0.get('Workgroup_Size_X', '')), r0.get('Grid_Size', r0.get('Grid_Size_X', ''))])
    print(open(os.path.join(root, 'profiles', f'{tag}_{cfg}_kernel_stats.csv')).read())

# ---- counters: one optimise launch per pass (tools/gpu.py one <cfg>: warm-up launch + measured launch, both counted and averaged)
tot, n, kern = collections.defaultdict(float), collections.defaultdict(int), set()
for f in glob.glob(os.path.join(out, f'{tag}_{cfg}_pmc', '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        if is_opt(r['Kernel_Name']):
            tot[r['Counter_Name']] += float(r['Counter_Value'])
            n[r['Counter_Name']] += 1
            kern.add(r['Kernel_Name'].split('(')[0])
per = {k: tot[k] / n[k] for k in tot}
warm_line = open(os.path.join(out, f'{tag}_{cfg}_pmc', 'warm.log')).read().strip().splitlines()[-1].split()
S, T = int(warm_line[warm_line.index('S') + 1]), int(warm_line[warm_line.index('T') + 1])
variant = ' '.join(warm_line[warm_line.index('P') + 2:])
summ = {
    'config': cfg, 'samples': S, 'iterations': T, 'variant': variant, 'kernels': sorted(kern),
    'note': 'rocprofv3 --pmc passes (separate runs, counters only; tools/profile.sh -> tools/gpu.py one); values per launch. FETCH_SIZE / '
            'WRITE_SIZE in KB as reported (4-byte-per-lane accesses: the gfx950 x2 FETCH correction for 16-B streams is not applied).',
    'FETCH_SIZE_KB_per_launch': per.get('FETCH_SIZE'), 'WRITE_SIZE_KB_per_launch': per.get('WRITE_SIZE'),
    'traffic_bytes_per_launch': (per.get('FETCH_SIZE', 0) + per.get('WRITE_SIZE', 0)) * 1024 if 'FETCH_SIZE' in per else None,
    'sq_counters_per_launch': {k: v for k, v in sorted(per.items()) if k.startswith('SQ_')},
}
if 'SQ_INSTS_VALU' in per:
    summ['per_sample_step'] = {k.replace('SQ_INSTS_', '').lower(): round(per[k] / (S * T), 1) for k in per if k.startswith('SQ_INSTS_')}
if 'SQ_WAVE_CYCLES' in per and per['SQ_WAVE_CYCLES']:
    # SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* count quad-cycles summed over waves; SQ_BUSY_CYCLES per SE
    summ['valu_active_fraction_of_wave_cycles'] = round(per.get('SQ_ACTIVE_INST_VALU', 0) / per['SQ_WAVE_CYCLES'], 4)
    summ['parked_fraction_of_wave_cycles'] = round(per.get('SQ_WAIT_ANY', 0) / per['SQ_WAVE_CYCLES'], 4)
if 'SQ_INST_CYCLES_VALU' in per:
    summ['note_valu_busy'] = 'valu_busy_fraction_of_simd = SQ_INST_CYCLES_VALU (quad-cycles the VALU port is held, summed over SIMDs) / (kernel time x clock / 4 x 1024 SIMDs), clock from the bench run'
json.dump(summ, open(os.path.join(root, 'profiles', f'{tag}_{cfg}_pmc_summary.json'), 'w'), indent=1)
print(json.dumps(summ, indent=1))
