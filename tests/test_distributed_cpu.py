"""CPU: the N>1 path (sample sharding + the one all-gather) with gloo, world sizes 2, 3, 4 and 8."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cpflow_amd import distributed


def test_shard_ranges_partition_all_samples():
    for S in (0, 1, 7, 10, 1024, 4097):
        for R in (1, 2, 3, 8):
            seen = []
            for r in range(R):
                first, count = distributed.shard_range(S, r, R)
                assert 0 <= count <= distributed.shard_capacity(S, R)
                seen += list(range(first, first + count))
            assert seen == list(range(S))


def _worker(rank, size, port, S, q, P=5):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=size)
    try:
        first, count = distributed.shard_range(S, rank, size)
        g = torch.Generator().manual_seed(1234)
        full = {'params': torch.rand(S, 2, P, generator=g), 'regloss': torch.rand(S, 2, generator=g),
                'cz': torch.randint(0, 30, (S,), generator=g, dtype=torch.int32),
                'passed': torch.rand(S, generator=g) > 0.5}
        local = {k: v[first:first + count] for k, v in full.items()}
        out = distributed.gather_samples(local, S)
        ok = all(out[k].dtype == full[k].dtype and torch.equal(out[k], full[k]) for k in full)
        # the job description bench.py prints: every rank calls it, every rank gets the same picture
        info = distributed.describe_job(None)
        ok = ok and info['backend'] == 'gloo' and info['world_size'] == size and [e['rank'] for e in info['ranks']] == list(range(size))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


# (3, 2), (4, 5): some ranks own no sample (shard_range hands out empty shards whenever ceil(S/R)(R-1) >= S,
# e.g. the README's num_samples=10 on 8 ranks); they must still take part in the collective
# world size 8 = the north star's node: the README's example (10 samples: ranks 5, 6, 7 own nothing, rank 4 two), and the
# splits of BASELINE configs 3 and 5 (4096 = 8 x 512 rows of 121 angles, 8192 = 8 x 1024 rows of 438: the real slab widths,
# rows gathered in global sample order)
@pytest.mark.parametrize('size,S,P', [(2, 10, 5), (2, 7, 5), (3, 8, 5), (3, 2, 5), (4, 5, 5), (8, 10, 5), (8, 4096, 121), (8, 8192, 438)])
def test_gather_equals_single_rank(size, S, P):
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    if size == 8:
        per = [distributed.shard_range(S, r, 8)[1] for r in range(8)]
        assert per == ([2, 2, 2, 2, 2, 0, 0, 0] if S == 10 else [S // 8] * 8)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, size, port, S, q, P)) for r in range(size)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, True) for r in range(size)]


def test_bench_parent_fails_when_a_rank_fails():
    """`python bench.py --gpus 2` started without torchrun on a box with fewer than 2 GPUs (here: none): every rank exits
    with an error before rendezvous, and the parent must return non-zero promptly instead of hanging or printing a line."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'CPF_ONE_DEVICE')}
    env['CPF_BENCH_RANK_TIMEOUT'] = '120'
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                         env=env, capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip('this box has two GPUs: the ranks would run')
    assert out.returncode != 0 and not out.stdout.strip().startswith('{') and time.time() - t0 < 200
    assert 'visible GPU' in out.stderr or 'exited with code' in out.stderr
