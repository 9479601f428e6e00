"""Sharding of one call over the GPUs of a node, and the ONE RCCL operation that collects the results.

Bands are independent units of work — exactly how the reference parallelises (joblib over bands,
narrow_band_least_squares.py:285) — so there is no data-path collective: every GPU runs the whole hot
path for its own bands (or, with fewer bands than GPUs, for its slice of every band's windows) and one
grouped RCCL operation inside ``libnbls_hip.so`` (``nbls_comm_gather``: grouped send/recv to a root, or
an all-gather) moves the result blocks over xGMI.  No PyTorch: the communicator lives in the library.

Two ways to drive several GPUs, chosen from the environment:

* one process, all GPUs (default when more than one device is visible and no launcher set a rank):
  one library handle per device, ``ncclCommInitAll``, per-device host threads only to queue the work;
* one process per GPU (``RANK`` / ``WORLD_SIZE`` / ``LOCAL_RANK`` set by ``torch.distributed.run``,
  ``mpirun`` or anything else): ``ncclCommInitRank``; the 128-byte communicator id goes from rank 0 to
  the others over a TCP socket on ``MASTER_ADDR`` (port ``NBLS_COMM_PORT``, default ``MASTER_PORT`` + 73).
"""
import ctypes as C
import os
import socket
import threading
import time

import numpy as np


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


def shard_bands_contiguous(costs, world):
    """Partition into ``world`` CONTIGUOUS band ranges of about equal cost (cut where the running cost passes k / world of
    the total) -> list (per rank) of ascending band indices; with at least as many bands as ranks nobody is left empty."""
    nb = len(costs)
    cum = np.concatenate(([0.0], np.cumsum(np.asarray(costs, dtype=np.float64))))
    cuts = [0]
    for r in range(1, world):
        target = cum[-1] * r / world
        b = int(np.searchsorted(cum, target))
        if b > 0 and b <= nb and target - cum[b - 1] < cum[b] - target:
            b -= 1                                   # the nearer of the two band boundaries
        b = min(max(b, cuts[-1] + (1 if nb >= world else 0)), nb - (world - r if nb >= world else 0))
        cuts.append(max(b, cuts[-1]))
    cuts.append(nb)
    return [list(range(cuts[r], cuts[r + 1])) for r in range(world)]


def plan_shards(costs, world):
    """The band shares of a sharded call: contiguous ranges when they balance about as well as the LPT partition (within
    5 % of its largest share) — rank order is then band order, and the process that drives all GPUs can build the
    dropped-element dictionary from each GPU's streamed rows while the passes are still running — else LPT.
    Deterministic: every rank of a launcher job computes the same shares.  -> (shards, contiguous flag)."""
    lpt = shard_bands(costs, world)
    con = shard_bands_contiguous(costs, world)
    load = lambda sh: max(sum(costs[b] for b in s) for s in sh) if sh else 0.0      # noqa: E731
    if load(con) <= 1.05 * load(lpt):
        return con, True
    return lpt, False


def env_rank():
    """(rank, world, local_rank) as a launcher exported them, else (0, 1, 0)."""
    try:
        world = int(os.environ.get('WORLD_SIZE', '1'))
        rank = int(os.environ.get('RANK', '0'))
        local = int(os.environ.get('LOCAL_RANK', str(rank)))
    except ValueError:
        return 0, 1, 0
    return (rank, world, local) if world > 1 else (0, 1, 0)


def visible_devices():
    """Device indices one process may drive: ``NBLS_DEVICES`` ("0,1,2"), else every HIP device."""
    env = os.environ.get('NBLS_DEVICES')
    if env:
        return [int(x) for x in env.split(',') if x.strip() != '']
    from ._hip import load_library
    return list(range(load_library().nbls_device_count()))


def set_transport_library(path, allow_shared_device=False):
    """Rehearsal on a one-GPU box (``nbls_comm_set_library``): resolve RCCL's entry points from ``path`` — the tests'
    loopback stand-in — and let one device carry several ranks.  Before the first communicator of the process."""
    from ._hip import load_library
    rc = load_library().nbls_comm_set_library(path.encode() if path else None, int(bool(allow_shared_device)))
    if rc != 0:
        raise RuntimeError('nbls_comm_set_library: RCCL has already been resolved from another library (%d)' % rc)


class Group:
    """The ranks (= GPUs) one call is sharded over.  ``handles`` are the library handles THIS process
    drives (all of them in the one-process form, one in the process-per-GPU form); ``ranks`` their ranks."""

    def __init__(self, handles, ranks, world, root):
        self.handles, self.ranks, self.world, self.root = handles, ranks, world, root

    def gather(self, block_bytes, status=0):
        """One grouped RCCL operation: -> (world, block_bytes) uint8 array of every rank's block (the last
        8 bytes of a block are that rank's status word), or None on a process that does not drive the root."""
        lib = self.handles[0].lib
        hs = (C.c_void_p * len(self.handles))(*[h._h for h in self.handles])
        deliver = self.root < 0 or self.root in self.ranks
        out = np.empty((self.world, block_bytes), dtype=np.uint8) if deliver else None
        rc = lib.nbls_comm_gather(hs, len(self.handles), self.root, block_bytes, int(status),
                                  out.ctypes.data if deliver else None, out.nbytes if deliver else 0)
        self.handles[0]._chk(rc)
        return out


_groups = {}
_group_override = None       # tests install a stand-in here (CPU rehearsal of the host logic)


def _exchange_unique_id(lib, rank, world, timeout=300.0):
    addr = os.environ.get('MASTER_ADDR', '127.0.0.1')
    port = int(os.environ.get('NBLS_COMM_PORT', str(int(os.environ.get('MASTER_PORT', '29500')) + 73)))
    if rank == 0:
        uid = (C.c_char * 128)()
        rc = lib.nbls_comm_unique_id(uid, 128)
        if rc != 0:
            raise RuntimeError('RCCL is not available (nbls_comm_unique_id -> %d)' % rc)
        payload = bytes(uid)
        with socket.create_server(('', port)) as srv:
            srv.settimeout(timeout)
            for _ in range(world - 1):
                conn, _ = srv.accept()
                with conn:
                    conn.sendall(payload)
        return payload
    deadline = time.time() + timeout
    while True:
        try:
            with socket.create_connection((addr, port), timeout=5.0) as s:
                buf = b''
                while len(buf) < 128:
                    chunk = s.recv(128 - len(buf))
                    if not chunk:
                        raise ConnectionError('short read')
                    buf += chunk
                return buf
        except OSError:
            if time.time() > deadline:
                raise RuntimeError('rank %d could not fetch the RCCL id from %s:%d' % (rank, addr, port))
            time.sleep(0.05)


def get_group():
    """The Group of this process, or None when a single GPU does the whole call.  Communicators are
    created once per process and reused."""
    if _group_override is not None:
        return _group_override
    from . import engine
    rank, world, local = env_rank()
    if world > 1:                                  # one process per GPU
        key = ('rank', rank, world, os.getpid())
        g = _groups.get(key)
        if g is None:
            h = engine.get_handle()
            uid = _exchange_unique_id(h.lib, rank, world)
            h._chk(h.lib.nbls_comm_init_rank(h._h, uid, world, rank))
            g = Group([h], [rank], world, root=-1)          # all-gather: every rank returns the results
            _groups[key] = g
        return g
    devs = visible_devices()
    if len(devs) < 2 and os.environ.get('NBLS_FORCE_DIST_PATH') != '1':
        return None
    key = ('all', tuple(devs), os.getpid())
    g = _groups.get(key)
    if g is None:
        seen = {}
        hs = []
        for d in devs:                               # (a device listed twice — loopback rehearsal on one GPU — gets two handles)
            k = seen.get(d, 0)
            seen[d] = k + 1
            hs.append(engine.get_handle(d, k))
        arr = (C.c_void_p * len(hs))(*[h._h for h in hs])
        hs[0]._chk(hs[0].lib.nbls_comm_init_all(arr, len(hs)))
        g = Group(hs, list(range(len(hs))), len(hs), root=0)
        _groups[key] = g
    return g


def run_on_handles(fn, handles):
    """fn(index, handle) for every local handle — concurrently (the library calls release the GIL) when
    there are several, so that the GPUs start together.  Exceptions are collected, not raised:
    -> list of (exception or None) per handle."""
    errs = [None] * len(handles)

    def work(i):
        try:
            fn(i, handles[i])
        except BaseException as e:      # noqa: BLE001 - reported through the gather's status word
            errs[i] = e
    if len(handles) == 1:
        work(0)
        return errs
    ths = [threading.Thread(target=work, args=(i,)) for i in range(len(handles))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    return errs
