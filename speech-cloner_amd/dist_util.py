"""One-process-per-GPU helpers (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" on CPU).

The conversion path shards by utterance/window with no data-path collective (SURVEY.md section
8e): every rank owns a contiguous slice of the batch.  The process group is only used for the
timing barrier, the max-over-ranks reduction of the timed region and gathering small results.
"""
import os


def env_world():
    """(rank, local_rank, world_size) from the torch.distributed.run environment."""
    return (int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0')),
            int(os.environ.get('WORLD_SIZE', '1')))


def init(backend=None, device_index=None):
    """Initialise the default process group when WORLD_SIZE > 1.  Returns (rank, world)."""
    import torch
    rank, local, world = env_world()
    if world > 1 and not torch.distributed.is_initialized():
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        kw = {}
        if backend == 'nccl':
            kw['device_id'] = torch.device('cuda', local if device_index is None else device_index)
        torch.distributed.init_process_group(backend, **kw)
    return rank, world


def shard_range(n_items, rank, world):
    """Contiguous slice [lo, hi) of ``n_items`` owned by ``rank``: sizes differ by at most one,
    earlier ranks take the larger shares, every item is owned exactly once."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def barrier():
    import torch
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.barrier()


def max_over_ranks(value, device='cpu'):
    """MAX all-reduce of a python float."""
    import torch
    if not (torch.distributed.is_available() and torch.distributed.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    return float(t.item())


def gather_concat(array, device='cpu'):
    """all_gather of equally-shaped numpy float32 arrays, concatenated on axis 0 in rank order."""
    import numpy as np
    import torch
    if not (torch.distributed.is_available() and torch.distributed.is_initialized()):
        return array
    t = torch.from_numpy(np.ascontiguousarray(array)).to(device)
    out = [torch.empty_like(t) for _ in range(torch.distributed.get_world_size())]
    torch.distributed.all_gather(out, t)
    return torch.cat(out, 0).cpu().numpy()


def finalize():
    import torch
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
