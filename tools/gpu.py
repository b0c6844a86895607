#!/usr/bin/env python3
"""GPU-side diagnostics in one CLI (none of this is a test or a product path).  Run from anywhere:

    python tools/gpu.py perf [cfg2 cfg3 cfg4 cfg5 n5 ...]      kernel time (HIP events) of the optimise launch; env T, FIRST=1
    python tools/gpu.py one <cfg> [S] [T]                      one warm-up + one launch (for rocprofv3 passes)
    python tools/gpu.py knobs <cfg> <S> "K=V,..." "K=V,..."    same-process A/B over the library's tuning switches (cpf_set_tuning)
    python tools/gpu.py ablib libA.so libB.so [cfg]            same-box A/B of two builds (alternating processes, CPF_LIB)
    python tools/gpu.py agree <KNOB> [cfg2|cfg3]               two settings of a knob: agreement on short runs + timing over S
    python tools/gpu.py fuzz [n_cases] [seed]                  random layers / block counts / rotation strings vs the f64 oracle
                                                               (env FUZZ_NS=4,5,6: qubit counts to draw from; FUZZ_KMAX: block counts for n >= 4)
    python tools/gpu.py fix4                                   4-qubit golden fixture: pass set of each kernel variant
    python tools/gpu.py stamps                                 per-phase cycle shares (library built with -DCPF_STAMPS)
    python tools/gpu.py static-e2e                             wall time of Synthesize.static() end to end
    python tools/gpu.py host-api                               end-to-end timing of the host-pointer entry cpf_run_static
"""
import itertools
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rand_u(n, seed):
    rng = np.random.default_rng(seed) if not isinstance(seed, np.random.Generator) else seed
    a = rng.normal(size=(2 ** n, 2 ** n)) + 1j * rng.normal(size=(2 ** n, 2 ** n))
    q, r = np.linalg.qr(a)
    return q * (np.diag(r) / np.abs(np.diag(r)))


def configs():
    """name -> (layer, K, target, sample counts); cfg2..cfg5 are BASELINE.json's (cpflow_amd/workloads.py)."""
    from cpflow_amd import workloads as w
    c = {k: (v.layer, v.num_cp_gates, v.target, [v.samples_per_gpu]) for k, v in w.CONFIGS.items()}
    c['cfg2'] = c['cfg2'][:3] + ([1024, 2048, 4096, 16384],)
    c['cfg3'] = c['cfg3'][:3] + ([512, 4096],)
    c['n5'] = ([[0, 1], [1, 2], [2, 3], [3, 4]], 36, rand_u(5, 0), [1000])
    c['n4star'] = ([[0, 1], [0, 2], [0, 3]], 23, rand_u(4, 1), [500])
    c['n5k86'] = ([[0, 1], [1, 2], [2, 3], [3, 4]], 86, rand_u(5, 2), [1000])          # the shape of paper/results/benchmarks/res_4gt13_91
    c['n5conn'] = ([[i, j] for i in range(5) for j in range(i + 1, 5)], 57, rand_u(5, 3), [2000])   # ... res_4gt10-v1_81 ('xz' there)
    c['n3k14'] = ([[0, 1], [1, 2]], 14, rand_u(3, 4), [100])                            # the adaptive tutorial's shape (toff3_chain)
    c['n3cz8'] = ([[0, 1], [0, 2], [1, 2]], 8, rand_u(3, 6), [1000])                     # paper/results/local_minimums: Ansatz(3, 'cz', ...)
    c['n3k24'] = ([[0, 1], [1, 2]], 24, rand_u(3, 7), [100])
    c['n3c18'] = ([[0, 1], [0, 2], [1, 2]], 18, rand_u(3, 5), [100])
    return c


TUNING = ('CPF_SPLIT', 'CPF_PSTORE', 'CPF_MSTORE', 'CPF_KSPEC', 'CPF_PAIRS', 'CPF_LONE', 'CPF_PRIO', 'CPF_LW', 'CPF_GROUP', 'CPF_DENSE', 'CPF_DENSE_LW',
          'CPF_DENSE_SPECIALISED', 'CPF_DENSE_FOLD', 'CPF_ZYZ', 'CPF_ZYZL', 'CPF_ZYZL_SAVE', 'CPF_RELABEL', 'CPF_FORCE_GENERIC', 'CPF_JIT', 'CPF_JIT_K', 'CPF_NO_AOT3', 'CPF_ZYZ3', 'CPF_ZYZ3_M', 'CPF_ZYZ3_PAIRS', 'CPF_ZYZ3_RES', 'CPF_ZYZ3_STAMPS', 'CPF_ZYZ3_DEFS')


def sync_tuning():
    """This tool keeps its A/B switches in os.environ; the release library does not read them from there, so they are
    handed over through cpf_set_tuning before every plan creation / launch."""
    from cpflow_amd import _ffi
    for k in TUNING:
        try:
            _ffi.set_tuning(k, os.environ.get(k))
        except Exception:
            # a release library does not know the diagnostics switches (CPF_ZYZ3_DEFS, CPF_ZYZ3_STAMPS, CPF_FORCE_GENERIC, ...):
            # build one with cpflow_amd.build.build_variant('diag', ['CPF_DIAGNOSTICS']) and select it with CPF_LIB
            if os.environ.get(k) is not None:
                raise SystemExit(f'{k} needs a library built with -DCPF_DIAGNOSTICS: python -c "from cpflow_amd import build; '
                                 f'build.build_variant(\'diag\', [\'CPF_DIAGNOSTICS\'])" and CPF_LIB=cpflow_amd/libcpflow_amd.diag.so')


def engine(cfg, T=None, **kw):
    from cpflow_amd.engine import TemplateEngine
    sync_tuning()
    layer, K, tgt, Ss = configs()[cfg]
    if cfg == 'n3cz8':
        kw.setdefault('entangler', 'cz')
    return TemplateEngine(layer, K, tgt, kw.pop('rot', 'xyz'), num_gd_iterations=int(os.environ.get('T', '2000')) if T is None else T, **kw), Ss


def timed(eng, init, reps=3):
    import torch
    out = eng.optimize(init)
    torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        out = eng.optimize(init)
        ms.append(eng.last_kernel_ms())
    return out, min(ms)


def cmd_perf(args):
    for name in (args or ['cfg2']):
        eng, Ss = engine(name)
        T = eng.num_gd_iterations
        for S in (Ss[:1] if os.environ.get('FIRST') else Ss):
            out, ms = timed(eng, eng.initial_angles(0, S))
            ok = out['passed']
            print(f'{name} S={S} T={T} kernel {ms:.2f} ms  {S * T / ms * 1e3:.3e} sample-steps/s  '
                  f'{eng.flops_per_sample_step * S * T / ms * 1e3 / 1e12:.2f} TFLOP/s  [{eng.last_variant()}] passed {int(ok.sum())} '
                  f'minCZ {int(out["cz"][ok].min()) if ok.any() else -1}', flush=True)


def cmd_one(args):
    import torch
    cfg = args[0] if args else 'cfg2'
    eng, Ss = engine(cfg, T=int(args[2]) if len(args) > 2 and args[2] else None)
    S = int(args[1]) if len(args) > 1 and args[1] else Ss[0]
    init = eng.initial_angles(0, S)
    for _ in range(2):
        eng.optimize(init)
    torch.cuda.synchronize()
    print('kernel ms', eng.last_kernel_ms(), 'cfg', cfg, 'S', S, 'T', eng.num_gd_iterations, 'P', eng.num_angles, '[' + eng.last_variant() + ']')


def cmd_knobs(args):
    import torch
    cfg, S, combos = args[0], int(args[1]), args[2:]
    eng, _ = engine(cfg)
    T = eng.num_gd_iterations
    init = eng.initial_angles(0, S)
    res, var, stats = {c: [] for c in combos}, {}, {}
    keys = {kv.split('=', 1)[0] for c in combos for kv in c.split(',')}
    for _ in range(int(os.environ.get('REPS', '5'))):
        for c in combos:
            for k in keys:                       # a switch one combination sets must not leak into the next
                os.environ.pop(k, None)
            for kv in c.split(','):
                k, v = kv.split('=', 1)
                os.environ[k] = v
            sync_tuning()
            out = eng.optimize(init)
            torch.cuda.synchronize()
            res[c].append(eng.last_kernel_ms())
            var[c] = eng.last_variant()
            stats[c] = (int(out['passed'].sum()), int(out['cz'][out['passed']].min()) if out['passed'].any() else -1)
    for c in combos:
        print(f'{cfg} S={S} {c:40s} min {min(res[c]):.3f} median {sorted(res[c])[len(res[c]) // 2]:.3f} mean {sum(res[c]) / len(res[c]):.3f} ms  {S * T / min(res[c]) * 1e3:.3e}/s  [{var[c]}] passed/minCZ {stats[c]}', flush=True)


def cmd_ablib(args):
    libs, cfg = args[:2], (args[2] if len(args) > 2 else 'cfg2')
    res = {lib: [] for lib in libs}
    for _ in range(3):
        for lib in libs:
            env = dict(os.environ, CPF_LIB=os.path.abspath(lib), FIRST='1')
            out = subprocess.run([sys.executable, os.path.abspath(__file__), 'perf', cfg], env=env, capture_output=True, text=True).stdout
            line = [x for x in out.splitlines() if x.startswith(cfg)][0]
            res[lib].append(float(line.split('kernel')[1].split('ms')[0]))
    for lib in libs:
        print(os.path.basename(lib), cfg, 'S=first', 'min %.2f ms' % min(res[lib]), res[lib])


def cmd_agree(args):
    """e.g. agree CPF_SPLIT cfg2 (two-ended vs one-wave), agree CPF_PSTORE cfg3 (stored states vs un-applying)."""
    import torch
    knob, cfg = args[0], (args[1] if len(args) > 1 else 'cfg2')

    def run(v, S, T, rot='xyz'):
        os.environ[knob] = str(v)
        eng, _ = engine(cfg, T=T, rot=rot)
        out, ms = timed(eng, eng.initial_angles(0, S), reps=3 if T >= 1000 else 1)
        return {k: x.cpu().numpy() for k, x in out.items() if torch.is_tensor(x)}, ms, eng.last_variant()
    for T in (0, 1, 2, 10, 50):
        a, _, va = run(1, 64, T)
        b, _, vb = run(0, 64, T)
        d = {k: float(np.max(np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)))) for k in ('regloss', 'loss', 'reg', 'params')}
        print(f'T={T:3d} [{va}] vs [{vb}] max|d|', {k: f'{v:.2e}' for k, v in d.items()}, 'cz equal', bool((a['cz'] == b['cz']).all()), flush=True)
    T = int(os.environ.get('T', '2000'))
    for S in (512, 1024, 2048, 4096):
        a, ms1, va = run(1, S, T)
        b, ms0, vb = run(0, S, T)
        print(f'{cfg} S={S}: {knob}=1 {ms1:.2f} ms  =0 {ms0:.2f} ms  ratio {ms0 / ms1:.3f}  passed {int(a["passed"].sum())}/{int(b["passed"].sum())} '
              f'[{va}] vs [{vb}]', flush=True)


def cmd_fuzz(args):
    from cpflow_amd.engine import TemplateEngine
    from oracle import Oracle
    N = int(args[0]) if args else 24
    rng = np.random.default_rng(int(args[1]) if len(args) > 1 else 0)
    ROTS = ['xyz', 'xz', 'yz', 'zxz', 'xyz', 'z', 'yxz', 'xy', 'zyx']
    # 3 qubits: the default (the two-ended ZYZ-form kernel where it serves the case), that kernel at a random cut, and the
    # gate-by-gate families behind their switches
    VARIANTS = [dict(), dict(CPF_ZYZ3='1', CPF_ZYZ3_M='?'), dict(CPF_PSTORE='1', CPF_KSPEC='1', CPF_JIT_K='1'),
                dict(CPF_PSTORE='0', CPF_KSPEC='0', CPF_JIT_K='0'), dict(CPF_SPLIT='0')]
    bad, t0 = 0, time.time()
    for _case in range(N):
        n = int(rng.choice([int(x) for x in os.environ['FUZZ_NS'].split(',')] if os.environ.get('FUZZ_NS') else [2, 3, 3, 3, 4, 4, 5, 6]))
        pairs = [list(p) for p in itertools.permutations(range(n), 2)]
        LL = min(int(rng.integers(1, 4)), len(pairs))
        layer = [pairs[i] for i in rng.choice(len(pairs), LL, replace=False)]
        if max(max(p) for p in layer) != n - 1:
            layer[0] = [int(rng.integers(n - 1)), n - 1]     # the layer defines n (topology.py:36-38)
        if n == 3 and rng.random() < 0.5:                    # the reference's own layers: block-count-specialised kernels from the shipped cache
            layer = [[[0, 1], [1, 2]], [[0, 1], [0, 2], [1, 2]]][int(rng.integers(2))]
        K = int(rng.integers(2, 27)) if n == 3 else int(rng.integers(2, int(os.environ.get('FUZZ_KMAX', '9'))))
        rot = ROTS[int(rng.integers(len(ROTS)))]
        ent = str(rng.choice(['cp', 'cp', 'cp', 'cz', 'cx']))
        target = rand_u(n, rng)
        T = 6
        S = 1100 if n in (4, 5) and rng.random() < 0.5 else 3          # 4 / 5 qubits: one wave per sample (relabelled qubits) or four
        if n == 6 and rng.random() < 0.5:
            S = 600                                                    # 6 qubits: two waves per sample (above 512 samples) or four
        if n == 3:
            S = int(rng.choice([3, 300, 700, 2100]))                   # 3 qubits: one sample per CU / stored states / saturated
        r = float(rng.choice([0.0, 0.004]))
        loss = str(rng.choice(['hst', 'hst', 'state', 'relative_phase']))
        if loss == 'relative_phase' and set(rot) == {'z'}:
            # every block gate is diagonal and this loss reads |U| only: the block angles' gradient is exactly zero, Adam walks
            # them on rounding noise (the float32 and float64 oracles part by 0.24 rad within 6 steps) and a CP angle that
            # leaves a penalty plateau that way changes reg (7e-4 between the two oracles) - compare without the penalty
            r = 0.0
        if loss == 'state':
            target = np.concatenate([target[:, :1], np.zeros((1 << n, (1 << n) - 1))], axis=1)
        mask = (rng.random(3 * n + (2 * len(rot) + (ent == 'cp')) * K) < 0.3).astype(np.uint8) if rng.random() < 0.4 else None
        orc = Oracle(layer, K, target, rot, num_gd_iterations=T, r=r, entangler=ent, loss_kind={'hst': 0, 'state': 1, 'relative_phase': 2}[loss],
                     frozen_mask=mask)
        init = orc.initial_angles(int(rng.integers(1 << 20)), S)
        ref = orc.run(init[:3], np.float64, keep_history=True)
        frozen = None if mask is None else np.broadcast_to(mask, (S, len(mask))).copy()
        for v in (VARIANTS if n == 3 else VARIANTS[:1]):
            for k in ('CPF_PSTORE', 'CPF_KSPEC', 'CPF_JIT_K', 'CPF_SPLIT', 'CPF_ZYZ3', 'CPF_ZYZ3_M'):
                os.environ.pop(k, None)
            v = {k: (str(int(rng.integers(1, K))) if x == '?' else x) for k, x in v.items()}
            os.environ.update(v)
            sync_tuning()
            eng = TemplateEngine(layer, K, target, rot, num_gd_iterations=T, r=r, entangler=ent, loss=loss)
            out = {k: x.cpu().numpy() for k, x in eng.optimize(init, frozen=frozen, keep_history=True).items() if x is not None}
            e0 = np.abs(out['history_regloss'][:3, 0] - ref['history_loss'][:, 0]).max()
            eh = np.abs(out['history_regloss'][:3] - ref['history_loss']).max()
            ep = np.abs(out['history_params'][:3, :3] - ref['history_params'][:, :3]).max()
            # A gradient component at rounding level (null directions of the state-preparation / relative-phase losses, e.g. the
            # first Rz on |0>; the odd accidental one with HST) is turned by Adam into a step of ~lr whose sign is noise, without
            # touching the loss: the loss history is the criterion, angles and CZ counts are compared where the angles agree
            same_angles = ep < 2e-4
            ok = e0 < 1e-5 and eh < 5e-5 and (not same_angles or (out['cz'][:3] == ref['cz']).all()) and (same_angles or ep < 0.35)
            bad += not ok
            print(f"{'ok ' if ok else 'BAD'} n={n} layer={layer} K={K} rot={rot} {ent} {loss} r={r} S={S} frozen={mask is not None} {v} [{eng.last_variant()}] first {e0:.1e} hist {eh:.1e} "
                  f"params {ep:.1e}", flush=True)
    print('cases', N, 'failures', bad, 'time %.0fs' % (time.time() - t0))
    return 1 if bad else 0


def cmd_fix4(_args):
    from cpflow_amd.engine import TemplateEngine
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'raw_toff4_star_t14.npz'))
    S, idx = int(g['num_samples']), set(g['sample_index'].tolist())
    for env in ({}, {'CPF_FORCE_GENERIC': '1'}, {'CPF_LW': '2'}, {'CPF_DENSE': '0'}):
        for k in ('CPF_FORCE_GENERIC', 'CPF_LW', 'CPF_DENSE'):
            os.environ.pop(k, None)
        os.environ.update(env)
        sync_tuning()
        eng = TemplateEngine(g['layer'], int(g['num_cp_gates']), g['target'], str(g['rotation_gates']), r=float(g['r']))
        out = {k: v.cpu().numpy() for k, v in eng.optimize(eng.initial_angles(int(g['seed']), S)).items() if v is not None}
        passed = set(np.flatnonzero(out['passed']).tolist())
        both = [k for k, j in enumerate(g['sample_index']) if j in passed]
        same = out['cz'][g['sample_index'][both]] == g['cz'][both]
        print(env, eng.last_variant(), 'passed', len(passed), 'symmetric difference', sorted(passed ^ idx), 'cz same', same.mean(), flush=True)


def cmd_stamps(args):
    cfg = args[0] if args else 'cfg2'
    eng, Ss = engine(cfg, T=500)
    T = 500
    out = eng.optimize(eng.initial_angles(0, int(os.environ.get('S', str(Ss[0])))), keep_history=True)
    split = eng.last_variant().startswith(('split', 'zyz3'))
    h = out['history_regloss'][0, :16].cpu().numpy().astype(np.float64)
    names = (['forward half', 'wait for other wave', 'loss/track', 'backward half', 'wait for other wave', 'owner', 'wait (owner)', 'loop top'] if split else
             ['surface fwd', 'blocks fwd', 'loss/track', 'blocks bwd', 'surface bwd', 'owner (Adam, sincos, penalty)', '-', 'loop top'])
    if eng.last_variant().startswith('zyz3'):
        names = ['owner / build', 'wait (tables)', 'forward half', 'wait for other wave', 'loss/track', 'gradient half', 'wait for other wave', 'row sums + chain']
    print(eng.last_variant())
    for w in ((0, 1) if split else (0,)):
        st = h[8 * w:8 * w + 8]
        print('wave', 'AB'[w], 'total', st.sum() / T, 'cycles/step')
        for n, v in zip(names, st):
            print(f'  {n:32s} {v / T:9.1f} cycles/step  {100 * v / st.sum():5.1f} %')


def cmd_static_e2e(_args):
    import contextlib
    import io
    from cpflow_amd import StaticOptions, Synthesize, workloads
    for S in (10, 1024):
        syn = Synthesize([[0, 1], [1, 2]], target_unitary=workloads.CCZ, label='ccz')
        opts = StaticOptions(num_cp_gates=12, accepted_num_cz_gates=10, num_samples=S)
        for rep in range(2):
            t0 = time.perf_counter()
            with contextlib.redirect_stdout(io.StringIO()):
                t1 = time.perf_counter()
                syn._generate_raw(opts)
                t2 = time.perf_counter()
                res = syn.static(opts, save_results=False)
            dt = time.perf_counter() - t0
            print(f'S={S} rep {rep}: raw stage {1e3 * (t2 - t1):.1f} ms, whole static() {1e3 * (dt - (t2 - t1)):.1f} ms, '
                  f'{len(res.decompositions)} decompositions, best {min(d.cz_count for d in res.decompositions)} CZ')


def cmd_host_api(_args):
    import ctypes as C
    from cpflow_amd import _ffi, workloads
    from cpflow_amd.engine import placements_for
    lib = _ffi.lib()
    layer, K, S, T, P = [[0, 1], [1, 2]], 12, 1024, 2000, 93
    pl, tgt, rot = placements_for(layer, K), np.ascontiguousarray(workloads.CCZ), b'xyz'
    pb = _ffi.Problem(3, K, pl.ctypes.data, 3, rot, 0, 0, tgt.ctypes.data, _ffi.Penalty(0, np.pi / 2, 2, .05, .05, .05), 1, 0.00055,
                      0.1, 0.9, 0.999, 1e-8, T, 0.2, 1e-3)
    out = dict(params=np.empty((S, 2, P), np.float32), regloss=np.empty((S, 2), np.float32), loss=np.empty((S, 2), np.float32),
               reg=np.empty((S, 2), np.float32), cz=np.empty(S, np.int32), passed=np.empty(S, np.uint8))
    res = _ffi.Result(*[out[k].ctypes.data for k in ('params', 'regloss', 'loss', 'reg', 'cz', 'passed')])
    for rep in range(4):
        t0 = time.perf_counter()
        rc = lib.cpf_run_static(C.byref(pb), 0, 0, S, 0, S, 0, None, None, C.byref(res))
        dt = time.perf_counter() - t0
        assert rc == 0, lib.cpf_last_error()
        print(f'cpf_run_static rep {rep}: {dt * 1e3:.2f} ms end to end -> {S * T / dt:.3e} sample-steps/s; passed {int(out["passed"].sum())}')


COMMANDS = {'perf': cmd_perf, 'one': cmd_one, 'knobs': cmd_knobs, 'ablib': cmd_ablib, 'agree': cmd_agree, 'fuzz': cmd_fuzz, 'fix4': cmd_fix4,
            'stamps': cmd_stamps, 'static-e2e': cmd_static_e2e, 'host-api': cmd_host_api}

if __name__ == '__main__':
    if len(sys.argv) < 2 or sys.argv[1] not in COMMANDS:
        sys.exit(__doc__)
    sys.exit(COMMANDS[sys.argv[1]](sys.argv[2:]) or 0)
