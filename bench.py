#!/usr/bin/env python3
"""Headline benchmark: sample-steps/s of the CP-template raw stage (BASELINE.json metric).

One "step" = one full pass of the hot path over one batch of synthetic input: draw the
initial angles from the seed on the device, run num_gd_iterations Adam iterations for every
sample (one persistent HIP kernel), evaluate CZ counts / entry-loss flags, and (N > 1)
all-gather the per-sample summaries.  value = (samples x iterations x steps) / wall time.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): CCZ target, 3 qubits, chain layer [[0,1],[1,2]],
num_cp_gates=12, num_samples=1024 per GPU, 2000 iterations, lr 0.1, r 0.00055, seed 0.
Weak scaling: every rank runs 1024 samples (global sample rows rank*1024 ...).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: peak FP32 vector = f32 MFMA dense peak
HBM_PEAK_GBS = 8000.0


def cpu_baseline(layer, K, target, rot, T, r, budget_s=12.0):
    """Time the CPU restatement (oracle/, a C port of the JAX path, OpenMP over samples) on the
    host cores, on a bounded sample of the same workload.  Reported, never shipped."""
    import oracle
    lib_path = None
    try:
        oracle.build(native=True)
        lib_path = os.path.join(ROOT, 'oracle', '_build', 'libcpf_oracle_native.so')
    except Exception:
        pass
    orc = oracle.Oracle(layer, K, target, rot, r=r, num_gd_iterations=T, lib_path=lib_path)
    cores = int(orc.lib.cpfo_max_threads())
    init = orc.initial_angles(0, cores)
    t0 = time.time()
    orc.run(init, np.float32, nthreads=cores)          # calibration: one sample per core
    dt = max(time.time() - t0, 1e-3)
    # ~budget_s seconds of work: whole multiples of the core count, rows of the same seed-0 angle matrix
    S = int(min(max(cores, cores * int(budget_s / dt)), 65536))
    init = orc.initial_angles(0, S)
    t0 = time.time()
    orc.run(init, np.float32, nthreads=cores)
    dt = time.time() - t0
    return {'value': S * T / dt, 'unit': 'sample-steps/s', 'cores': cores, 'kind': 'port',
            'sample': f'{S} samples (seed-0 initial angles) x {T} iterations of the same 3q CCZ workload, float32 C '
                      f'restatement of the JAX path (oracle/cpf_oracle.c, OpenMP, '
                      f'{"-march=native" if lib_path else "portable -O3"}), {dt:.1f} s'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--samples-per-gpu', type=int, default=1024)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from cpflow_amd import distributed
    from cpflow_amd.engine import TemplateEngine

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1:
        # CPF_DIST_BACKEND=gloo + CPF_ONE_DEVICE=1 emulate N ranks on a single GPU (contract smoke test only)
        if os.environ.get('CPF_ONE_DEVICE') == '1':
            local_rank = 0
        torch.cuda.set_device(local_rank)
        backend = os.environ.get('CPF_DIST_BACKEND', 'nccl')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend)
    n_gpus = world if world > 1 else 1
    if args.gpus != n_gpus and rank == 0:
        print(f'[bench] --gpus {args.gpus} but WORLD_SIZE={world}: running {n_gpus} rank(s)', file=sys.stderr)

    layer, K, rot, T, r, seed = [[0, 1], [1, 2]], 12, 'xyz', 2000, 0.00055, 0
    target = np.diag([1, 1, 1, 1, 1, 1, 1, -1]).astype(np.complex128)       # CCZ
    S_local = args.samples_per_gpu
    S_total = S_local * n_gpus
    eng = TemplateEngine(layer, K, target, rot, r=r, num_gd_iterations=T, device=local_rank if world > 1 else None)
    first, count = distributed.shard_range(S_total, rank, n_gpus)
    assert count == S_local

    def step():
        init = eng.initial_angles(seed, S_total, first, count)
        out = eng.optimize(init)
        summary = distributed.gather_samples(
            {k: out[k] for k in ('regloss', 'loss', 'cz', 'passed')}, S_total)
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
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    k_ms, _all = eng.mean_kernel_ms()   # average launch duration over the timed region
    eng.event_log = None
    out, summary = step()
    fence()

    t = torch.tensor([elapsed], dtype=torch.float64, device=eng.device if world == 1 or dist.get_backend() == 'nccl' else 'cpu')
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        value = S_total * T * args.steps / elapsed
        flops = eng.flops_per_sample_step                      # 3 N^2 (32 K + 16 n) + 20 P = 84,804
        achieved_tf = flops * S_local * T / (k_ms * 1e-3) / 1e12
        P = eng.num_angles
        alg_bytes = S_local * (8 * P + 28) + 8 * 8 * 8         # persistent model: theta0 in, best theta + scalars out
        traffic = None      # HBM bytes per launch from the separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
        try:
            import glob
            newest = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_summary.json')))[-1]
            with open(newest) as f:
                traffic = json.load(f)['traffic_bytes_per_launch'] if S_local == 1024 else None
        except Exception:
            pass
        line = {
            'metric': 'optimizer steps/sec x num_samples (sample-steps/s), 3q CCZ synth',
            'value': value, 'unit': 'sample-steps/s', 'n_gpus': n_gpus, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'CCZ 3q, chain [[0,1],[1,2]], num_cp_gates={K}, num_samples={S_local}/GPU '
                                   f'({S_total} total), num_gd_iterations={T}, lr=0.1, r={r}, seed={seed}, xyz',
                       'parallelism': f'sample-sharded x{n_gpus}', 'passed_entry_loss': int(summary['passed'].sum())},
            'roofline': {'bound': 'mfma', 'achieved': achieved_tf, 'peak': FP32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': achieved_tf / FP32_PEAK_TFLOPS, 'traffic': traffic,
                         'note': 'fp32 vector ALU work (no MFMA used); peak = f32 vector = f32 MFMA dense peak; '
                                 f'kernel {k_ms:.3f} ms per launch (HIP events), {flops:.0f} flop/sample-step',
                         'kernel': eng.last_variant()},
            'hbm': {'algorithmic_bytes_per_launch': alg_bytes,
                    'achieved_GBps': alg_bytes / (k_ms * 1e-3) / 1e9, 'peak_GBps': HBM_PEAK_GBS,
                    'frac': alg_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
        }
        if not args.no_cpu_baseline and n_gpus == 1:
            try:
                line['cpu_baseline'] = cpu_baseline(layer, K, target, rot, T, r)
                line['speedup_vs_cpu_port'] = value / line['cpu_baseline']['value']
            except Exception as e:      # the baseline is reported, never required
                line['cpu_baseline'] = {'value': None, 'unit': 'sample-steps/s', 'cores': 0, 'kind': 'port',
                                        'sample': f'failed: {e}'}
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
