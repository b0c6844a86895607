"""Turn the counter CSVs of tools/pmc_pass.sh (gpurun_out/pmc_<tag>/) into profiles/<tag>_pmc_summary.json.
usage: python tools/pmc_summary.py <tag> [S] [T]"""
import collections, csv, glob, json, os, sys
tag = sys.argv[1]
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
T = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tot = collections.defaultdict(float); n = collections.defaultdict(int); kern = set()
for f in glob.glob(os.path.join(root, 'gpurun_out', 'pmc_' + tag, '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if ('k_cp' in k or 'cpf_jit' in k) and 'initial' not in k:
            tot[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1; kern.add(k.split('(')[0])
per = {k: tot[k] / n[k] for k in tot}
waves = 2 * S if any('split' in k for k in kern) else S
out = {
    'note': f'rocprofv3 --pmc passes (separate runs, tools/pmc_pass.sh -> tools/gpu_one.py = config-2 launches, S={S}, T={T}); '
            'values per launch (launches of a pass averaged). FETCH_SIZE / WRITE_SIZE in KB as reported (4-byte-per-lane '
            'accesses: the gfx950 x2 FETCH correction for 16-B streams is not applied).',
    'kernels': sorted(kern),
    'FETCH_SIZE_KB_per_launch': per.get('FETCH_SIZE'), 'WRITE_SIZE_KB_per_launch': per.get('WRITE_SIZE'),
    'traffic_bytes_per_launch': (per.get('FETCH_SIZE', 0) + per.get('WRITE_SIZE', 0)) * 1024 if 'FETCH_SIZE' in per else None,
    'sq_counters_per_launch': {k: v for k, v in sorted(per.items()) if k.startswith('SQ_')},
}
if 'SQ_INSTS_VALU' in per:
    ss = S * T
    out['per_sample_step'] = {k.replace('SQ_INSTS_', '').lower(): round(per[k] / ss, 1) for k in per if k.startswith('SQ_INSTS_')}
if 'SQ_WAVE_CYCLES' in per:
    ws = waves * T
    out['per_wave_step_quad_cycles'] = {'total': round(per['SQ_WAVE_CYCLES'] / ws, 1), 'parked (WAIT_ANY)': round(per.get('SQ_WAIT_ANY', 0) / ws, 1),
                                        'issue stall (WAIT_INST_ANY)': round(per.get('SQ_WAIT_INST_ANY', 0) / ws, 1),
                                        'valu': round(per.get('SQ_ACTIVE_INST_VALU', 0) / ws, 1)}
    out['valu_busy_fraction_of_simd'] = round(per.get('SQ_ACTIVE_INST_VALU', 0) * (waves / (S if S >= 1024 else 1024)) / per['SQ_WAVE_CYCLES'], 3) if S == 1024 else None
dst = os.path.join(root, 'profiles', tag + '_pmc_summary.json')
json.dump(out, open(dst, 'w'), indent=1)
print(open(dst).read())
