#!/usr/bin/env python3
"""Headline benchmark: sample-steps/s of the CP-template raw stage (BASELINE.json metric).

One "step" = one full pass of the hot path over one batch of synthetic input: draw the
initial angles from the seed on the device, run num_gd_iterations Adam iterations for every
sample (one persistent HIP kernel), evaluate CZ counts / entry-loss flags, and (N > 1)
all-gather the per-sample summaries.  value = (samples x iterations x steps) / wall time.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2|cfg3|cfg4|cfg5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
    python bench.py --config cfg3 --gpus 8      # north_star's 8-GPU configurations: cfg3 (4096 samples), cfg5 (8192)
    python bench.py --config cfg5 --gpus 8

Started WITHOUT torchrun and with --gpus N > 1, this process only launches N fresh worker
processes (one rank per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set per child) before
any GPU call of its own, and relays rank 0's JSON line.

Default workload (BASELINE.json configs[1]): CCZ target, 3 qubits, chain layer [[0,1],[1,2]],
num_cp_gates=12, num_samples=1024 per GPU, 2000 iterations, lr 0.1, r 0.00055, seed 0.
Weak scaling: every rank runs the configuration's per-GPU sample count (global sample rows
rank*count ...).  --config selects another BASELINE configuration (cpflow_amd/workloads.py).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: peak FP32 vector (packed v_pk_fma_f32 issue) = f32 MFMA dense peak
HBM_PEAK_GBS = 8000.0


def launch_ranks(n):
    """--gpus N without torchrun: start N workers (this file again) and relay rank 0's stdout.
    Nothing in this process has touched the GPU (torch is not even imported yet)."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    # a rank that fails or hangs must not hang the parent: poll all ranks, end everyone as soon as one exits non-zero or
    # the deadline passes (CPF_BENCH_RANK_TIMEOUT seconds, default 1500), and return a non-zero code
    import threading
    deadline = time.time() + float(os.environ.get('CPF_BENCH_RANK_TIMEOUT', '1500'))
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    codes, failed = [None] * n, None
    while any(c is None for c in codes) and failed is None:
        for r, p in enumerate(procs):
            if codes[r] is None:
                codes[r] = p.poll()
                if codes[r] not in (None, 0):
                    failed = f'rank {r} exited with code {codes[r]}'
        if time.time() > deadline:
            failed = f'ranks still running after the deadline: {[r for r, c in enumerate(codes) if c is None]}'
        time.sleep(0.2)
    if failed:
        print(f'[bench] {failed}; stopping the other ranks', file=sys.stderr)
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
    reader.join(timeout=10)
    out = b''.join(chunks)
    for line in out.decode().splitlines():                # the one JSON line to stdout; library chatter (gloo) to stderr
        print(line, file=sys.stdout if line.startswith('{') else sys.stderr)
    sys.stdout.flush()
    return 1 if failed else 0


def usable_cores():
    """The cores this process may really use: the affinity mask, capped by the cgroup's CPU quota (a GPU box hands a job a share of
    its host: 128 hardware threads visible, a fraction of them usable - running one OpenMP thread per VISIBLE core on such a box
    oversubscribes the share several times over and was what round 5's `cores: 128` line did)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith('cpu.max'):
                quota, period = (None if parts[0] == 'max' else float(parts[0])), float(parts[1])
            else:
                quota = float(parts[0])
                with open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as f2:
                    period = float(f2.read())
                quota = None if quota <= 0 else quota
            if quota:
                n = max(1, min(n, int(quota / period + 0.5)))
            break
        except Exception:
            continue
    return n


def cpu_baseline(wl, layer, K, target, rot, T, r, budget_s=10.0):
    """Time the CPU restatement (oracle/, a C port of the JAX path, OpenMP over samples) on the host cores, on a bounded sample of
    the same workload: the scalar float32 build (`kind: "port"`, the line's `cpu_baseline`), the same run vectorised over 16
    samples at a time (`simd`: kind "port-simd", oracle/cpf_oracle_simd.h), and BASELINE config 1 - the reference's own
    CPU-runnable case (README.md:35: "from one to five minutes" for its static() call) - as wall seconds of its raw stage
    (`config1`).  Reported, never shipped, never the target."""
    import oracle
    from cpflow_amd import workloads
    lib_path = None
    try:
        oracle.build(native=True)
        lib_path = os.path.join(ROOT, 'oracle', '_build', 'libcpf_oracle_native.so')
    except Exception:
        pass
    flags = '-march=native' if lib_path else 'portable -O3'
    cores = usable_cores()

    def timed(run, budget):
        """~budget seconds of work: all T iterations on whole multiples of the core count (x 16 lanes for the SIMD build) when such a
        batch fits the budget, otherwise one batch over as many iterations as fit (rows of the same seed-0 angle matrix)."""
        simd = run == 'run_simd'
        batch = cores * (16 if simd else 1)
        cal = oracle.Oracle(layer, K, target, rot, r=r, num_gd_iterations=min(T, 20), lib_path=lib_path)
        init = cal.initial_angles(0, batch)
        for _ in range(2):              # the first call pays for the thread pool and the page faults
            t0 = time.time()
            getattr(cal, run)(init, nthreads=cores) if simd else cal.run(init, np.float32, nthreads=cores)
            per_step = max(time.time() - t0, 1e-5) / (min(T, 20) + 1)
        reps = int(budget / (per_step * T))
        S, T_run = (int(min(batch * reps, 1 << 20)), T) if reps >= 1 else (batch, max(20, int(budget / per_step)))
        orc = oracle.Oracle(layer, K, target, rot, r=r, num_gd_iterations=T_run, lib_path=lib_path)
        init = orc.initial_angles(0, S)
        t0 = time.time()
        getattr(orc, run)(init, nthreads=cores) if simd else orc.run(init, np.float32, nthreads=cores)
        dt = time.time() - t0
        return S, T_run, dt

    S, T_run, dt = timed('run', budget_s)
    out = {'value': S * T_run / dt, 'unit': 'sample-steps/s', 'cores': cores, 'kind': 'port',
           'sample': f'{S} samples (seed-0 initial angles) x {T_run} iterations of the same workload ({wl.name}: {wl.title}), float32 C '
                     f'restatement of the JAX path (oracle/cpf_oracle.c, scalar, OpenMP over samples on the {cores} cores this job may '
                     f'use of {os.cpu_count()} visible, {flags}), {dt:.1f} s'}
    try:
        S2, T2, dt2 = timed('run_simd', budget_s)
        out['simd'] = {'value': S2 * T2 / dt2, 'unit': 'sample-steps/s', 'cores': cores, 'kind': 'port-simd',
                       'sample': f'{S2} samples x {T2} iterations, the same restatement over 16 samples at a time (oracle/cpf_oracle_simd.h: '
                                 f'structure of arrays, the scalar build\'s operations in its order, libmvec sinf / cosf; {flags}), {dt2:.1f} s'}
    except NotImplementedError as e:
        out['simd'] = {'value': None, 'kind': 'port-simd', 'sample': f'not covered: {e}'}
    except Exception as e:
        out['simd'] = {'value': None, 'kind': 'port-simd', 'sample': f'failed: {e}'}
    try:
        w1 = workloads.CONFIGS['cfg1']
        o1 = oracle.Oracle(w1.layer, w1.num_cp_gates, w1.target, w1.rotation_gates, r=w1.r, num_gd_iterations=w1.num_gd_iterations, lib_path=lib_path)
        init = o1.initial_angles(w1.seed, w1.samples_per_gpu)
        t0 = time.time()
        o1.run(init, np.float32, nthreads=min(cores, w1.samples_per_gpu))
        d1 = time.time() - t0
        t0 = time.time()
        o1.run_simd(init, nthreads=1)
        d1s = time.time() - t0
        out['config1'] = {'workload': f'cfg1: {w1.title}, num_cp_gates={w1.num_cp_gates}, num_samples={w1.samples_per_gpu}, '
                                      f'num_gd_iterations={w1.num_gd_iterations} (BASELINE configs[0], the README run; raw stage only)',
                          'wall_s': d1, 'cores': min(cores, w1.samples_per_gpu), 'kind': 'port',
                          'wall_s_simd': d1s, 'cores_simd': 1, 'sample_steps_per_s': w1.samples_per_gpu * w1.num_gd_iterations / d1,
                          'reference_says': 'README.md:35 "from one to five minutes" for the whole static() call (JIT, verification and qiskit '
                                            'post-processing included; hardware not stated)'}
    except Exception as e:
        out['config1'] = {'error': str(e)[:200]}
    return out


def executed_flops(variant, num_qubits, num_blocks, algorithmic):
    """Approximate flops the kernel EXECUTES per sample-step, for the ZYZ-form kernels only (None otherwise: the gate-by-gate
    kernels execute SURVEY's count and more), a packed multiply counted as 2 and a packed FMA as 4.  zyz/, zyzl/: the generated
    group routines (cpf_zyz_asm.h) spend 15 + 51 packed instructions per four amplitudes and block = 13 + 47 flop per amplitude.
    zyz3/ (cpf_zyz3.h): per amplitude and block 14 forward (one complex multiply, two real stages), 34 in the gradient half (the
    same on the other matrix, the three products behind the 7 generator sums) and 7 adds in the row sums = 55.  The surface gates
    and per-step phases are left at SURVEY's figure."""
    if not variant.startswith(('zyz/', 'zyzl/', 'zyz3/')):
        return None
    n2 = float(1 << (2 * num_qubits))
    per_amp = 55.0 if variant.startswith('zyz3/') else 60.0
    return algorithmic - 3.0 * n2 * 32.0 * num_blocks + n2 * per_amp * num_blocks


def other_configs(workloads, TemplateEngine, torch, launches=3):
    """The other BASELINE configurations' per-GPU shares (cfg3: 512 samples, cfg4: 2048, cfg5: 1024) measured in this same
    process after the headline run: one warm-up and `launches` timed launches each (HIP events on the launch stream), so
    that the driver's own run records them too.  They never enter `value`."""
    rows = {}
    for name in ('cfg3', 'cfg4', 'cfg5'):
        wl = workloads.CONFIGS[name]
        try:
            eng = TemplateEngine(wl.layer, wl.num_cp_gates, wl.target, wl.rotation_gates, r=wl.r, num_gd_iterations=wl.num_gd_iterations)
            S = wl.samples_per_gpu
            init = eng.initial_angles(wl.seed, S * wl.gpus, 0, S)
            eng.optimize(init)
            torch.cuda.synchronize()
            eng.event_log = []
            for _ in range(launches):
                out = eng.optimize(init)
            ms, all_ms = eng.mean_kernel_ms()
            tf = eng.flops_per_sample_step * S * wl.num_gd_iterations / (ms * 1e-3) / 1e12
            rows[name] = {'workload': f'{wl.title}, num_cp_gates={wl.num_cp_gates}, num_samples={S} (one GPU\'s share of {S * wl.gpus})',
                          'kernel': eng.last_variant(), 'kernel_ms': ms, 'kernel_ms_all': all_ms,
                          'sample_steps_per_s': S * wl.num_gd_iterations / (ms * 1e-3), 'tflops': tf, 'frac': tf / FP32_PEAK_TFLOPS,
                          'flop_per_sample_step': eng.flops_per_sample_step, 'passed_entry_loss': int(out['passed'].sum())}
            ex = executed_flops(eng.last_variant(), wl.num_qubits, wl.num_cp_gates, eng.flops_per_sample_step)
            if ex is not None:          # 'frac' prices SURVEY's dense-4x4 count; the ZYZ form executes fewer flops for the same result
                rows[name]['flop_executed_per_sample_step'] = ex
                rows[name]['frac_executed'] = rows[name]['frac'] * ex / eng.flops_per_sample_step
            slots, _ = stored_issue_slots(name)
            if slots:
                rows[name].update(slots)
                rows[name]['port_floor_frac'] = slots['port_floor_ms'] / ms if slots.get('port_floor_ms') else None
            eng.close()
        except Exception as e:          # reported, never required
            rows[name] = {'error': str(e)[:300]}
    return rows


def stored_traffic(config, samples_per_gpu):
    """HBM bytes per launch from the newest committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE summary of this
    configuration (profiles/r*_<config>_pmc_summary.json; counters need their own profiler passes, so they cannot be
    read inside this run).  Returns (bytes or None, file name or None)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_summary.json')), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
        except Exception:
            continue
        if d.get('config', 'cfg2') == config and d.get('samples', 1024) == samples_per_gpu and d.get('traffic_bytes_per_launch'):
            return d['traffic_bytes_per_launch'], os.path.relpath(path, ROOT)
    return None, None


def stored_issue_slots(config):
    """What the configuration's kernel asks of a SIMD's vector port (tools/issue_slots.py: the static mix of the step loops scaled to
    the counters' dynamic instruction count, priced at the placement-verified rates of tools/micro/issue_cost2.hip), from the newest
    committed profiles/r*_issue_slots.json.  Returns (dict or None, file name or None)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_issue_slots.json')), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
        except Exception:
            continue
        if config in d:
            e = d[config]
            return {k: e.get(k) for k in ('issue_slots_per_sample_step', 'port_cycles_per_sample_step', 'port_floor_ms', 'port_busy')}, os.path.relpath(path, ROOT)
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)      # 0.6 s of back-to-back launches at config 2: long enough for a utilisation sampler
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', default='cfg2', choices=['cfg2', 'cfg3', 'cfg4', 'cfg5'],
                    help='BASELINE.json configuration (default: configs[1], the one the metric is quoted on)')
    ap.add_argument('--samples-per-gpu', type=int, default=None, help="override the configuration's per-GPU sample count")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-other-configs', action='store_true', help='skip the extra.other_configs block of the cfg2 line')
    ap.add_argument('--strong-ref', action='store_true',
                    help='N > 1: after the timed region rank 0 also runs ALL samples of the job on its one GPU (the strong-scaling '
                         'reference), so that the line carries weak AND strong efficiency')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(args.gpus))

    import torch
    import torch.distributed as dist
    from cpflow_amd import distributed, workloads
    from cpflow_amd.engine import TemplateEngine

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1:
        # CPF_DIST_BACKEND=gloo + CPF_ONE_DEVICE=1 emulate N ranks on a single GPU (contract smoke test only)
        one_device = os.environ.get('CPF_ONE_DEVICE') == '1'
        if one_device:
            local_rank = 0
        elif torch.cuda.device_count() < world:
            sys.exit(f'[bench] {world} ranks but {torch.cuda.device_count()} visible GPU(s); on a 1-GPU box the N-rank '
                     f'path can be rehearsed with CPF_ONE_DEVICE=1 CPF_DIST_BACKEND=gloo')
        torch.cuda.set_device(local_rank)
        backend = os.environ.get('CPF_DIST_BACKEND', 'gloo' if one_device else 'nccl')
        import datetime
        pg_timeout = datetime.timedelta(seconds=float(os.environ.get('CPF_DIST_TIMEOUT', '600')))
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank), timeout=pg_timeout)
        else:
            dist.init_process_group(backend, timeout=pg_timeout)
    n_gpus = world
    if args.gpus != n_gpus and rank == 0:
        print(f'[bench] --gpus {args.gpus} but WORLD_SIZE={world}: running {n_gpus} rank(s)', file=sys.stderr)

    wl = workloads.CONFIGS[args.config]
    layer, K, rot, T, r, seed, target = wl.layer, wl.num_cp_gates, wl.rotation_gates, wl.num_gd_iterations, wl.r, wl.seed, wl.target
    S_local = args.samples_per_gpu or wl.samples_per_gpu
    S_total = S_local * n_gpus
    eng = TemplateEngine(layer, K, target, rot, r=r, num_gd_iterations=T, device=local_rank if world > 1 else None)
    first, count = distributed.shard_range(S_total, rank, n_gpus)
    assert count == S_local

    gather_ms = []                      # wall time of the ONE collective of a step, per timed step (N > 1)

    def step():
        init = eng.initial_angles(seed, S_total, first, count)
        out = eng.optimize(init)
        if world > 1:
            # the collective on its own clock: drain this rank's kernel first, so that launch skew between the ranks (the wait inside
            # the all-gather for the slowest one) and the kernel's own time separate in the line (`distributed.per_rank`)
            torch.cuda.synchronize()
            tg = time.perf_counter()
        summary = distributed.gather_samples(
            {k: out[k] for k in ('regloss', 'loss', 'cz', 'passed')}, S_total)
        if world > 1:
            torch.cuda.synchronize()
            gather_ms.append((time.perf_counter() - tg) * 1e3)
        return out, summary

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    eng.event_log = []                  # HIP events on the launch stream around every optimize kernel
    gather_ms.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    k_ms, k_all = eng.mean_kernel_ms()  # average launch duration over the timed region
    eng.event_log = None
    my_gather = list(gather_ms)
    out, summary = step()
    fence()

    # strong-scaling reference (opt-in): the whole job's samples on rank 0's GPU alone, same step, three timed passes
    strong_ref = None
    if args.strong_ref and world > 1:
        if rank == 0:
            def step_all():
                return eng.optimize(eng.initial_angles(seed, S_total, 0, S_total))
            step_all()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                step_all()
            torch.cuda.synchronize()
            strong_ref = {'samples': S_total, 'ms_per_step_one_gpu': (time.perf_counter() - t1) / 3 * 1e3, 'kernel': eng.last_variant()}
        dist.barrier()

    local_elapsed = elapsed
    t = torch.tensor([elapsed], dtype=torch.float64, device=eng.device if world == 1 or dist.get_backend() == 'nccl' else 'cpu')
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # what torch.distributed saw: backend, world size, and the distinct devices behind the ranks (every rank contributes
    # its device's UUID / PCI bus id), so that a reader can tell N ranks on N devices from N ranks on one
    dist_info = distributed.describe_job(eng.device)
    if world > 1:
        # per rank: its own mean kernel time, its own mean time inside the collective (a rank that arrives early waits there for the
        # slowest one) and its own wall time of the timed region - launch skew, RCCL latency and kernel time in one line
        mine = {'rank': rank, 'kernel_ms': k_ms, 'collective_ms': sum(my_gather) / max(1, len(my_gather)),
                'collective_ms_min': min(my_gather) if my_gather else None, 'collective_ms_max': max(my_gather) if my_gather else None,
                'elapsed_ms_per_step': local_elapsed / args.steps * 1e3}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        dist_info['per_rank'] = per_rank
        dist_info['collective'] = {'what': 'one all-gather per step of a [samples per rank, 6] float32 slab (regloss[2], loss[2], cz, passed: '
                                           'cpflow_amd/distributed.py gather_samples), timed per rank around the call with the rank\'s own '
                                           'kernel drained first',
                                   'bytes_per_rank': int(S_local * 6 * 4)}

    if rank == 0:
        value = S_total * T * args.steps / elapsed
        flops = eng.flops_per_sample_step                      # 3 N^2 (32 K + 16 n) + 20 P (84,804 at config 2)
        achieved_tf = flops * S_local * T / (k_ms * 1e-3) / 1e12
        P, N = eng.num_angles, 1 << wl.num_qubits
        alg_bytes = S_local * (8 * P + 28) + 8 * N * N         # persistent model: theta0 in, best theta + scalars out
        traffic, traffic_src = stored_traffic(args.config, S_local)
        line = {
            'metric': 'optimizer steps/sec x num_samples (sample-steps/s), 3q CCZ synth' if args.config == 'cfg2'
                      else f'optimizer steps/sec x num_samples (sample-steps/s), BASELINE {args.config}',
            'value': value, 'unit': 'sample-steps/s', 'n_gpus': n_gpus, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'{args.config}: {wl.title}, num_cp_gates={K}, num_samples={S_local}/GPU '
                                   f'({S_total} total), num_gd_iterations={T}, lr=0.1, r={r}, seed={seed}, {rot}',
                       'parallelism': f'sample-sharded x{n_gpus}', 'passed_entry_loss': int(summary['passed'].sum())},
            'roofline': {'bound': 'valu', 'achieved': achieved_tf, 'peak': FP32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': achieved_tf / FP32_PEAK_TFLOPS, 'traffic': traffic, 'traffic_source': traffic_src,
                         'note': 'FP32 vector-ALU issue bound (no MFMA, no HBM traffic inside the run); peak = packed-FMA '
                                 f'vector peak; kernel {k_ms:.3f} ms per launch (mean of {len(k_all)} timed launches, HIP '
                                 f'events on the launch stream; median {sorted(k_all)[len(k_all) // 2]:.3f}), '
                                 f'{flops:.0f} flop/sample-step; traffic = HBM bytes per launch from the stored rocprofv3 '
                                 '--pmc passes named in traffic_source (not measured in this run); flop/sample-step is SURVEY '
                                 '8(d)\'s algorithmic count (one dense 4x4 per block and pass) whatever algebra the kernel runs - the '
                                 'ZYZ-form kernels (zyz/, zyzl/, zyz3/) execute fewer: frac_executed (DESIGN.md section 4)',
                         'kernel': eng.last_variant()},
            'hbm': {'algorithmic_bytes_per_launch': alg_bytes,
                    'achieved_GBps': alg_bytes / (k_ms * 1e-3) / 1e9, 'peak_GBps': HBM_PEAK_GBS,
                    'frac': alg_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
        }
        ex = executed_flops(eng.last_variant(), wl.num_qubits, K, flops)
        if ex is not None:          # what the kernel executes (the ZYZ form needs fewer flops than SURVEY's dense-4x4 count behind `frac`)
            rf = line['roofline']
            line['roofline'] = {'bound': rf['bound'], 'achieved': rf['achieved'], 'peak': rf['peak'], 'unit': rf['unit'], 'frac': rf['frac'],
                                'frac_executed': rf['frac'] * ex / flops, 'flop_executed_per_sample_step': ex,
                                **{k: v for k, v in rf.items() if k not in ('bound', 'achieved', 'peak', 'unit', 'frac')}}
        slots, slots_src = stored_issue_slots(args.config)
        if slots and S_local == wl.samples_per_gpu:
            # the port's own view: instructions, not flops (stored, like `traffic`: counters need their own profiler passes)
            line['roofline'].update(slots)
            line['roofline']['port_floor_frac'] = slots['port_floor_ms'] / k_ms if slots.get('port_floor_ms') else None
            line['roofline']['issue_slots_source'] = slots_src
        line['distributed'] = dist_info
        if strong_ref:
            # strong efficiency = t(1 GPU, all samples) / (N x t(N GPUs, all samples)); the weak one is the driver's from its per-N values
            strong_ref['strong_scaling_efficiency'] = strong_ref['ms_per_step_one_gpu'] / (n_gpus * line['ms_per_step'])
            line['strong_ref'] = strong_ref
        if n_gpus == 1 and args.config == 'cfg2' and not args.no_other_configs:
            line['extra'] = {'other_configs': other_configs(workloads, TemplateEngine, torch)}
        if not args.no_cpu_baseline and n_gpus == 1:
            try:
                line['cpu_baseline'] = cpu_baseline(wl, layer, K, target, rot, T, r)
                line['speedup_vs_cpu_port'] = value / line['cpu_baseline']['value']
            except Exception as e:      # the baseline is reported, never required
                line['cpu_baseline'] = {'value': None, 'unit': 'sample-steps/s', 'cores': 0, 'kind': 'port',
                                        'sample': f'failed: {e}'}
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
