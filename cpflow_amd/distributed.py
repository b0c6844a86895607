"""Sample sharding across the GPUs of one node (one process per GPU, torch.distributed).

The reference is single-device: `jit(vmap(...))` over all samples (cpflow/optimization.py:361-363).
Samples never interact, so here rank k runs the contiguous block of global sample indices
[k*ceil(S/R), ...) — initial-angle row j is drawn from the global index j, so the result is
independent of R — and ONE all-gather after the loop collects what the entry_loss /
accepted_num_cz_gates filter reads (cpflow/main.py:594-603, 660): regloss, loss, reg, cz, pass,
plus the {initial, best} angle rows.  No per-step traffic.  Backend "nccl" is RCCL on ROCm
(xGMI); "gloo" is used by the CPU tests.
"""
import torch
import torch.distributed as dist


def world():
    """(rank, world_size) — (0, 1) when torch.distributed is not initialised."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(num_samples, rank, world_size):
    """[first, first+count) of global sample indices owned by `rank` (contiguous, ceil-sized blocks)."""
    per = -(-int(num_samples) // int(world_size))
    first = min(rank * per, num_samples)
    return first, min(per, num_samples - first)


def shard_capacity(num_samples, world_size):
    return -(-int(num_samples) // int(world_size))


def gather_samples(local, num_samples, group=None, force_collective=False):
    """All-gather per-sample tensors.

    `local` maps name -> tensor whose first dimension is this rank's sample count.  Every
    rank returns name -> tensor with first dimension `num_samples`, rows in global sample
    order.  All fields travel in ONE collective: they are packed as float32 columns of a
    [capacity, width] slab (int32 / bool fields are exactly representable).
    """
    rank, size = world()
    if size == 1 and not force_collective:
        return local
    names = list(local)
    cap = shard_capacity(num_samples, size)
    first, count = shard_range(num_samples, rank, size)
    cols, widths = [], []
    for name in names:
        t = local[name]
        assert t.shape[0] == count, (name, t.shape, count)
        # explicit width: reshape(0, -1) is ambiguous on a rank whose shard is empty (e.g. 10 samples on 8 ranks),
        # and such a rank must still enter the collective
        width = 1
        for d in t.shape[1:]:
            width *= int(d)
        flat = t.reshape(count, width).to(torch.float32)
        cols.append(flat)
        widths.append(width)
    slab = torch.cat(cols, dim=1) if cols else torch.empty(count, 0)
    pad = torch.zeros(cap, slab.shape[1], dtype=torch.float32, device=slab.device)
    pad[:count] = slab
    device = slab.device
    if dist.get_backend(group) == 'gloo' and pad.is_cuda:      # CPU tests / single-GPU emulation of N ranks
        pad = pad.cpu()
    full = torch.empty(size * cap, slab.shape[1], dtype=torch.float32, device=pad.device)
    dist.all_gather_into_tensor(full, pad, group=group) if _has_into_tensor(pad) else _all_gather_list(full, pad, size, group)
    full = full.to(device)
    rows = torch.cat([full[r * cap:r * cap + shard_range(num_samples, r, size)[1]] for r in range(size)], dim=0)
    out, col = {}, 0
    for name, w in zip(names, widths):
        t = local[name]
        piece = rows[:, col:col + w].reshape((num_samples,) + tuple(t.shape[1:]))
        out[name] = piece.to(t.dtype)
        col += w
    return out


def _has_into_tensor(t):
    # gloo (CPU tests) lacks all_gather_into_tensor on some builds; nccl/RCCL has it
    return t.is_cuda


def _all_gather_list(full, pad, size, group):
    chunks = [torch.empty_like(pad) for _ in range(size)]
    dist.all_gather(chunks, pad, group=group)
    full.copy_(torch.cat(chunks, dim=0))


def describe_job(device=None):
    """What torch.distributed sees of this job, gathered from every rank: backend, world size, and per rank the device index,
    UUID, PCI bus id and host - so that a benchmark line can show N ranks on N distinct GPUs (bench.py `distributed`).
    Every rank must call it (one object all-gather); without a process group it describes the single process."""
    import socket
    me = {'rank': world()[0], 'device_index': None, 'uuid': '', 'pci_bus_id': '', 'name': 'cpu', 'host': socket.gethostname()}
    if device is not None and torch.cuda.is_available():
        dev = torch.device(device)
        props = torch.cuda.get_device_properties(dev)
        me.update(device_index=dev.index, uuid=str(getattr(props, 'uuid', '')), name=props.name,
                  pci_bus_id=f'{getattr(props, "pci_domain_id", 0):04x}:{getattr(props, "pci_bus_id", 0):02x}:{getattr(props, "pci_device_id", 0):02x}')
    rank, size = world()
    if size == 1 and not (dist.is_available() and dist.is_initialized()):
        return {'backend': None, 'world_size': 1, 'ranks': [me], 'distinct_devices': 1}
    everyone = [None] * size
    dist.all_gather_object(everyone, me)
    distinct = len({(e['host'], e['uuid'] or e['pci_bus_id'], e['device_index']) for e in everyone})
    return {'backend': dist.get_backend(), 'world_size': dist.get_world_size(), 'ranks': everyone, 'distinct_devices': distinct}
