"""Band sharding across the GPUs of one node (one process per GPU, ``torch.distributed``;
backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for the tests).

Bands are independent units of work — this is exactly how the reference parallelises
(joblib over bands, narrow_band_least_squares.py:285) — so there is no data-path collective:
each rank runs the whole hot path for its own bands and ONE all-gather of the padded result
grids at the end puts every band on every rank.  torch is plumbing here (process group and
the collective); no kernel of the path runs through it.
"""
import numpy as np


def dist_info():
    """(rank, world_size, backend) of the default process group, or (0, 1, None)."""
    try:
        import torch.distributed as td
    except Exception:
        return 0, 1, None
    if td.is_available() and td.is_initialized():
        return td.get_rank(), td.get_world_size(), td.get_backend()
    return 0, 1, None


def band_costs(npts, fs, winlens, winover, npairs):
    """Relative cost per band: windows x pairs x W^2 (the cross-correlation dominates)."""
    costs = []
    for wl in winlens:
        W = int(wl * fs)
        inc = max(1, int(np.round((1 - winover) * W)))
        nwin = max(0, -(-(npts - W) // inc))
        costs.append(float(nwin) * npairs * float(W) * float(W))
    return costs


def shard_bands(costs, world):
    """Longest-processing-time greedy partition -> list (per rank) of ascending band indices.
    Deterministic: ties go to the lower rank."""
    order = sorted(range(len(costs)), key=lambda b: (-costs[b], b))
    load = [0.0] * world
    out = [[] for _ in range(world)]
    for b in order:
        r = min(range(world), key=lambda i: (load[i], i))
        out[r].append(b)
        load[r] += costs[b]
    return [sorted(x) for x in out]


class _DevArray:
    """Expose a raw device pointer through ``__cuda_array_interface__`` so that torch can wrap
    HBM owned by libnbls_hip.so without a copy."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {'shape': tuple(shape), 'typestr': typestr,
                                         'data': (int(ptr), False), 'version': 2, 'strides': None}


def all_gather_arrays(arr, device_index=None):
    """All-gather one equal-shaped numpy array per rank -> list of numpy arrays (rank order)."""
    import torch
    import torch.distributed as td
    rank, world, backend = dist_info()
    if backend is None:
        return [arr]
    t = torch.from_numpy(np.ascontiguousarray(arr))
    if backend == 'nccl':
        t = t.cuda(device_index if device_index is not None else torch.cuda.current_device())
    outs = [torch.empty_like(t) for _ in range(world)]
    td.all_gather(outs, t)
    return [o.cpu().numpy() for o in outs]


def all_gather_device_grid(ptr, shape, dtype=np.float64, device_index=0):
    """All-gather a result grid straight from HBM (RCCL reads the library's buffer): -> list of
    numpy arrays, rank order.  Falls back to a staged copy if torch cannot wrap the pointer."""
    import torch
    import torch.distributed as td
    rank, world, backend = dist_info()
    typestr = np.dtype(dtype).str
    t = None
    if backend == 'nccl':
        try:
            t = torch.as_tensor(_DevArray(ptr, shape, typestr), device='cuda:%d' % device_index)
        except Exception:
            t = None
    if t is None:
        raise RuntimeError('device-pointer gather needs the nccl backend and __cuda_array_interface__ support')
    outs = [torch.empty_like(t) for _ in range(world)]
    td.all_gather(outs, t)
    return [o.cpu().numpy() for o in outs]


def all_gather_device_grids(ptrs, nbytes, shape, device_index=0):
    """All-gather the four result grids (vel, baz, mdccm, sigma_tau) of every rank with ONE collective
    when they are one contiguous block in HBM (the library allocates them so), else one per grid.
    -> list over ranks of (4, *shape) float64 arrays."""
    import torch
    import torch.distributed as td
    rank, world, backend = dist_info()
    if backend != 'nccl':
        raise RuntimeError('device-pointer gather needs the nccl backend')
    cells = int(np.prod(shape))
    contiguous = all(int(ptrs[i + 1]) - int(ptrs[i]) == nbytes for i in range(3)) and nbytes == cells * 8
    if not contiguous:
        per = [all_gather_device_grid(p, shape, np.float64, device_index) for p in ptrs[:4]]
        return [np.stack([per[g][r] for g in range(4)]) for r in range(world)]
    t = torch.as_tensor(_DevArray(ptrs[0], (4 * cells,), np.dtype(np.float64).str), device='cuda:%d' % device_index)
    key = (world, 4 * cells, device_index)
    bufs = _GATHER_BUFFERS.get(key)
    if bufs is None:        # device destination + pinned host landing zone, reused by every call of this shape
        bufs = (torch.empty((world, 4 * cells), dtype=torch.float64, device=t.device),
                torch.empty((world, 4 * cells), dtype=torch.float64, pin_memory=True))
        _GATHER_BUFFERS.clear()
        _GATHER_BUFFERS[key] = bufs
    out, pinned = bufs
    td.all_gather_into_tensor(out, t)
    pinned.copy_(out, non_blocking=True)
    torch.cuda.current_stream(t.device).synchronize()
    host = pinned.numpy()
    return [host[r].reshape((4,) + tuple(shape)).copy() for r in range(world)]


_GATHER_BUFFERS = {}
