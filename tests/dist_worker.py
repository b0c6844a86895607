"""Worker of tests/test_gpu_synthesis.py::test_two_ranks_on_one_gpu_equal_single_process: one rank of an N-rank job
(gloo rendezvous from the environment, every rank on cuda:0) running the real front end; rank 0 writes what it gathered."""
import contextlib
import io
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(out_path, num_samples):
    import torch
    import torch.distributed as dist
    from cpflow_amd import StaticOptions, Synthesize, workloads
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world > 1:
        torch.cuda.set_device(0)
        dist.init_process_group('gloo')
    opts = StaticOptions(num_cp_gates=12, accepted_num_cz_gates=10, num_samples=num_samples, num_gd_iterations_at_verification=1500)
    syn = Synthesize([[0, 1], [1, 2]], target_unitary=workloads.CCZ, label='ccz')
    raw = syn._generate_raw(opts)
    with contextlib.redirect_stdout(io.StringIO()):
        res = syn.static(opts, save_results=False)
    if world == 1 or dist.get_rank() == 0:
        np.savez(out_path, params=np.stack([r['params'] for r in raw]), regloss=np.stack([r['regloss'] for r in raw]),
                 loss=np.stack([r['loss'] for r in raw]), cz=raw.cz, passed=raw.passed,
                 dec_cz=np.array([d.cz_count for d in res.decompositions]), dec_loss=np.array([d.loss for d in res.decompositions]))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]))
