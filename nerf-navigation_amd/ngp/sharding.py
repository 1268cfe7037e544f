"""Ray / view sharding across the GPUs of one node: replicas of the model, independent rays, and no
collective on the data path.  The helpers are backend-agnostic (nccl = RCCL on the GPUs, gloo in the CPU tests)."""
import torch
import torch.distributed as dist


# True: take every collective even in a group of ONE rank (bench.py --force-dist, tests/test_gpu_rccl_single_rank.py): the only way a one-GPU box
# can execute the RCCL calls of the N > 1 path.  A process group must be initialised.
FORCE_COLLECTIVES = False


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def collectives_on():
    """whether the collectives of the N > 1 path run: more than one rank, or forced on an initialised single-rank group"""
    rank, ws = world()
    return ws > 1 or (FORCE_COLLECTIVES and dist.is_available() and dist.is_initialized())


def barrier():
    if collectives_on():
        dist.barrier()


def pose_indices(rank, world_size, n_poses):
    """Views rank `rank` renders at steps 0..n_poses-1: the orbit of n_poses*world_size views, phase-shifted by the rank,
    so that no two ranks ever render the same view at the same step and all views are covered."""
    total = n_poses * world_size
    return [(k * world_size + rank) % total for k in range(n_poses)]


def row_band(rank, world_size, height, align=8):
    """Contiguous band of image rows [lo, hi) for strong-scaling one frame; bands are multiples of `align` rows
    (the 8x8 tile order of the fused kernel) except possibly the last."""
    rows = -(-height // world_size)
    rows = -(-rows // align) * align
    lo = min(rank * rows, height)
    return lo, min(lo + rows, height)


def reduce_throughput(samples, seconds, device):
    """(sum over ranks of samples, max over ranks of seconds): the two scalars bench.py reports from."""
    if not collectives_on():
        return float(samples), float(seconds)
    s = torch.tensor([float(samples)], dtype=torch.float64, device=device)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(s.item()), float(t.item())


def gather_rows(local, device=None):
    """All-gather a [rows, ...] tensor of per-rank row bands into the full image on every rank (optional: the
    reference's eval loop gathers predictions the same way, nerf/utils.py:872-882)."""
    rank, ws = world()
    if not collectives_on():
        return local
    sizes = [torch.zeros(1, dtype=torch.int64, device=local.device) for _ in range(ws)]
    dist.all_gather(sizes, torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device))
    m = int(max(int(s.item()) for s in sizes))
    pad = torch.zeros((m,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(ws)]
    dist.all_gather(parts, pad)
    return torch.cat([p[: int(s.item())] for p, s in zip(parts, sizes)], dim=0)
