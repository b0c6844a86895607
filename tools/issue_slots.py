#!/usr/bin/env python3
"""What a launch asks of a SIMD's vector port, per sample-step, for the four BASELINE configurations.

    python tools/issue_slots.py <tag>            e.g. r06   (CPU only: hipcc -S of the ahead-of-time units + the committed counters)

For each configuration: the STATIC mix of its kernel's step loops (tools/isa_mix.py classes over `hipcc -S` of the ahead-of-time
translation unit: packed, DPP / permlane, transcendental, other vector, LDS, scalar, s_nop) scaled to the DYNAMIC vector-instruction
count of profiles/<tag>_<cfg>_pmc_summary.json (SQ_INSTS_VALU per sample-step: the static loops run once per step, the ratio absorbs
the set-up and epilogue), and from them

  issue_slots_per_sample_step   vector instructions + the DPP / permlane ones once more (the count VERDICT r5 asked for)
  port_cycles_per_sample_step   the same mix priced per class at the placement-verified rates of tools/micro/issue_cost2.hip
                                (profiles/r06_issue_cost2.txt, four waves per SIMD = the port saturated): other vector 1.94 cycles,
                                packed 3.29, DPP / permlane 3.7, transcendental 7.9
  port_floor_ms                 port cycles x samples x iterations / (1024 SIMDs x 2.4 GHz): what the launch would take if the vector
                                port were the only limit and never idle
  port_busy                     SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES-normalised: the fraction of the launch's SIMD time the port was
                                held (from the same counters: valu_active_fraction_of_wave_cycles x waves per SIMD, capped at 1)

Writes profiles/<tag>_issue_slots.json; bench.py puts the configuration's entry into `roofline`.
"""
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from isa_mix import classify  # noqa: E402

RATES = {'valu': 1.94, 'pk': 3.29, 'dpp': 3.7, 'perm': 3.7, 'trans': 7.9, 'mfma': 8.0}       # cycles per wave64 instruction, port saturated
TRANS = ('v_rcp_f32', 'v_rsq_f32', 'v_sqrt_f32', 'v_sin_f32', 'v_cos_f32', 'v_exp_f32', 'v_log_f32')
UNITS = {'cfg2': ('cpf_zyz3_aot.hip', []), 'cfg3': ('cpf_zyz3_aot.hip', []), 'cfg4': ('cpf_zyz_aot.hip', 'dense'), 'cfg5': ('cpf_zyz_aot.hip', 'dense')}
SAMPLES = {'cfg2': 1024, 'cfg3': 512, 'cfg4': 2048, 'cfg5': 1024}
WAVES_PER_SIMD = {'cfg2': 2, 'cfg3': 1, 'cfg4': 2, 'cfg5': 2}


def listing(unit, dense):
    from cpflow_amd import build
    out = f'/tmp/issue_slots_{os.path.basename(unit)}.s'
    flags = [f for f in build.FLAGS if f not in ('-shared',)] + (build.DENSE_FLAGS if dense else [])
    subprocess.check_call([build.HIPCC] + flags + ['-S', '--cuda-device-only', '-o', out, os.path.join(ROOT, 'cpflow_amd', 'csrc', unit)],
                          stderr=subprocess.DEVNULL)
    return open(out).read().splitlines()


def mangled_key(kernel):
    """'void cpf::k_cp_zyz3<8451, 12, 4, 1, 0, 1, 1, 2>' -> the run of Li..E tokens of its mangled name"""
    name, args = re.match(r'void cpf::(\w+)<(.*)>', kernel).groups()
    return name, ''.join(f'Li{a.strip()}E' for a in args.split(','))


def step_loops(lines, kernel):
    name, key = mangled_key(kernel)
    # the mangled name carries the template arguments as Li<n>E tokens, the layer pack between J and E
    start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\S*:', l) and name in l and ('I' + key + 'E') in l.split(':')[0].replace('J', ''))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end')) - 1      # (a kernel may hold several s_endpgm: the end of the function, not the first of them)
    # loops = backward branches; keep the outermost ones that hold an s_barrier or, failing that, the largest one
    raw = []
    body = [l.split(';')[0].strip() for l in lines[start + 1:end + 1]]
    flat = [t for t in body if t and not (t.startswith('.') and not t.startswith('.LBB')) and not (t.endswith(':') and not t.startswith('.LBB'))]
    idx, pos = {}, 0
    seq = []
    for t in flat:
        m = re.match(r'^(\.LBB\S+):', t)
        if m:
            idx[m.group(1)] = pos
        else:
            seq.append(t)
            pos += 1
    for i, t in enumerate(seq):
        parts = t.split(None, 1)
        if parts[0].startswith('s_cbranch') or parts[0] == 's_branch':
            tgt = parts[1].strip() if len(parts) > 1 else ''
            if tgt in idx and idx[tgt] <= i:
                raw.append((idx[tgt], i))
    # the step loops: loops with at least three s_barrier inside (a step's phases), largest first, none overlapping a larger one;
    # kernels without barriers (one wave per sample): the largest loop
    nbar = lambda r: sum(seq[j].startswith('s_barrier') for j in range(r[0], r[1] + 1))      # noqa: E731
    chosen = []
    for r in sorted((r for r in raw if nbar(r) >= 3), key=lambda r: r[0] - r[1]):
        if not any(not (r[1] < c[0] or c[1] < r[0]) for c in chosen):
            chosen.append(r)
    if not chosen:
        chosen = [max(raw, key=lambda r: r[1] - r[0])]
    mix = collections.Counter()
    for a, b in chosen:
        for j in range(a, b + 1):
            m = seq[j].split(None, 1)[0]
            c = classify(m)
            if c == 'valu' and m.split('_e')[0] in TRANS:
                c = 'trans'
            mix[c] += 1
    return dict(mix), [b - a + 1 for a, b in chosen]


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else 'r06'
    out, cache = {}, {}
    for cfg, (unit, dense) in UNITS.items():
        pmc_path = os.path.join(ROOT, 'profiles', f'{tag}_{cfg}_pmc_summary.json')
        if not os.path.exists(pmc_path):
            continue
        pmc = json.load(open(pmc_path))
        if unit not in cache:
            cache[unit] = listing(unit, dense)
        mix, lengths = step_loops(cache[unit], pmc['kernels'][0])
        vec_static = sum(mix.get(k, 0) for k in ('valu', 'pk', 'dpp', 'perm', 'trans', 'mfma'))
        vec_dyn = pmc['per_sample_step']['valu']
        scale = vec_dyn / vec_static
        slots = vec_dyn + scale * (mix.get('dpp', 0) + mix.get('perm', 0))
        cycles = scale * sum(mix.get(k, 0) * RATES[k] for k in RATES)
        S, T = SAMPLES[cfg], pmc.get('iterations', 2000)
        floor_ms = cycles * pmc.get('samples', S) * T / (1024 * 2.4e9) * 1e3
        out[cfg] = {'kernel': pmc['kernels'][0], 'static_step_loops': lengths, 'static_mix': mix, 'vector_per_sample_step_static': vec_static,
                    'vector_per_sample_step_pmc': vec_dyn, 'issue_slots_per_sample_step': round(slots, 1),
                    'port_cycles_per_sample_step': round(cycles, 1), 'port_floor_ms': round(floor_ms, 3),
                    'port_busy': round(min(1.0, pmc['valu_active_fraction_of_wave_cycles'] * WAVES_PER_SIMD[cfg]), 3),
                    'rates_cycles_per_instruction': RATES, 'samples': pmc.get('samples', S), 'iterations': T,
                    'source': [f'hipcc -S of cpflow_amd/csrc/{unit}', os.path.relpath(pmc_path, ROOT), 'profiles/r06_issue_cost2.txt']}
        print(cfg, json.dumps({k: out[cfg][k] for k in ('static_step_loops', 'static_mix', 'vector_per_sample_step_pmc', 'issue_slots_per_sample_step',
                                                      'port_cycles_per_sample_step', 'port_floor_ms', 'port_busy')}))
    with open(os.path.join(ROOT, 'profiles', f'{tag}_issue_slots.json'), 'w') as f:
        json.dump(out, f, indent=1)


if __name__ == '__main__':
    main()
