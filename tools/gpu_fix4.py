"""Diagnostic: 4-qubit golden fixture, pass set of each kernel variant against the stored one."""
import os, sys
import numpy as np, torch
sys.path.insert(0, '.')
from cpflow_amd.engine import TemplateEngine
g = np.load('tests/golden/raw_toff4_star_t14.npz')
S = int(g['num_samples']); idx = set(g['sample_index'].tolist())
for env in ({}, {'CPF_FORCE_GENERIC': '1'}, {'CPF_LW': '2'}):
    for k in ('CPF_FORCE_GENERIC', 'CPF_LW'): os.environ.pop(k, None)
    os.environ.update(env)
    eng = TemplateEngine(g['layer'], int(g['num_cp_gates']), g['target'], str(g['rotation_gates']), r=float(g['r']))
    init = eng.initial_angles(int(g['seed']), S)
    out = {k: v.cpu().numpy() for k, v in eng.optimize(init).items() if v is not None}
    best = out['regloss'].argmin(1)
    loss = np.take_along_axis(out['loss'], best[:, None], 1)[:, 0]
    passed = set(np.flatnonzero(out['passed']).tolist())
    diff = sorted(passed ^ idx)
    print(env, eng.last_variant(), 'passed', len(passed), 'diff', diff, [(j, float(loss[j]), j in idx) for j in diff], flush=True)
    both = [k for k, j in enumerate(g['sample_index']) if j in passed]
    same = out['cz'][g['sample_index'][both]] == g['cz'][both]
    print('   cz same', same.mean(), 'loss 428', float(loss[428]), 'regloss', out['regloss'][428], flush=True)
