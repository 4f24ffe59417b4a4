"""1-D block-cyclic distributed Cholesky for one large GP likelihood.

Used only when a single N is too large / too slow for one GPU; smaller problems shard over theta
instead (distributed.sharded_map, no collective).  One process per GPU; outer panel j (`panel`
128-tile columns = 512 wide by default) belongs to rank j % world.  Per step k:

    owner(k+1):  panel stream   LA(k): apply panel k to its own panel k+1; factor panel k+1; pack it
    everyone:    comm stream    broadcast(packed panel k+1)   <- the ONE real exchange step of the path
    everyone:    main stream    REST(k): apply panel k to the owned panels j >= k+2 (+ corner tile on rank 0);
                                owner(k+2) does panel k+2 FIRST (own event): LA(k+1) waits for that piece only

so the broadcast of panel k+1 (<= 134 MB at N=32768, received over one xGMI link, ~1 ms) and the
owner's panel factorisation both fly under the trailing update of panel k (look-ahead).  The
scalars (sum log L_ii, |z|^2, info) are merged with one tiny all-reduce at the end.

All device memory for packed panels comes from torch, all ordering is expressed with torch
streams / events, and the collective is torch.distributed.broadcast (backend "nccl" = RCCL over
xGMI).  The compute steps are the C-ABI calls gphip_dist_* (same HIP kernels as the single-GPU
path).  `LoopbackComm` runs several virtual ranks inside ONE process on ONE GPU (device-to-device
copies instead of RCCL): that is how the schedule is verified on a 1-GPU box.
"""
from __future__ import annotations

import math

import numpy as np

LOG_TWO_PI = math.log(2.0 * math.pi)


class TorchDistComm:
    """SPMD: this process is one rank; collectives through torch.distributed (RCCL on GPUs)."""

    def __init__(self, dist):
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.local_ranks = [self.rank]

    def broadcast(self, bufs: dict, src: int):
        """bufs: {local rank: tensor}; async broadcast enqueued on the current stream's order."""
        return self.dist.broadcast(bufs[self.rank], src=src, async_op=True)

    def allreduce_scalars(self, vals: dict):
        import torch
        dev = "cuda" if self.dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor(vals[self.rank], dtype=torch.float64, device=dev)
        s = t.clone()
        self.dist.all_reduce(s[:2], op=self.dist.ReduceOp.SUM)
        self.dist.all_reduce(s[2:], op=self.dist.ReduceOp.MAX)
        return s.cpu().numpy()


class LoopbackComm:
    """All `world` virtual ranks live in this process on one GPU; broadcast = device copies."""

    def __init__(self, world: int):
        self.world = world
        self.local_ranks = list(range(world))

    def broadcast(self, bufs: dict, src: int):
        for r, t in bufs.items():
            if r != src:
                t.copy_(bufs[src], non_blocking=True)
        return None

    def allreduce_scalars(self, vals: dict):
        arr = np.array([vals[r] for r in self.local_ranks], dtype=np.float64)
        return np.concatenate([arr[:, :2].sum(axis=0), arr[:, 2:].max(axis=0)])


class _HostStream:
    """Stand-in for torch.cuda.Stream when the step backend is synchronous (CPU schedule tests):
    every step has completed when its call returns, so waits are no-ops."""
    cuda_stream = 0

    def wait_event(self, ev):
        pass

    def wait_stream(self, st):
        pass


class _HostEvent:
    def record(self, stream=None):
        pass


class _GpuRuntime:
    def __init__(self, torch, device):
        self.torch, self.device = torch, device

    def stream(self, high_priority=False):
        return self.torch.cuda.Stream(device=self.device, priority=-1 if high_priority else 0)

    def event(self):
        return self.torch.cuda.Event()

    def on(self, stream):
        return self.torch.cuda.stream(stream)


class _HostRuntime:
    def stream(self, high_priority=False):
        return _HostStream()

    def event(self):
        return _HostEvent()

    def on(self, stream):
        import contextlib
        return contextlib.nullcontext()


class DistributedCholesky:
    """handles: {rank: step backend} for the ranks local to this process (one in SPMD mode).  A step
    backend is a `_lib.Handle` (HIP kernels; device="cuda") or any object with the same dist_* methods
    taking torch tensors for the packed panels (the CPU schedule tests use a numpy one; device="cpu")."""

    def __init__(self, handles: dict, comm, device=None):
        import torch
        self.torch = torch
        self.handles = handles
        self.comm = comm
        self.world = comm.world
        if device == "cpu":
            self.device = torch.device("cpu")
            self.rt = _HostRuntime()
        else:
            self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
            self.rt = _GpuRuntime(torch, self.device)
        any_h = next(iter(handles.values()))
        self.N = any_h.N
        self.nouter = any_h.dist_num_panels()
        self.shapes = [any_h.dist_panel_shape(k) for k in range(self.nouter)]
        self.main, self.panel, self.commst = {}, {}, {}
        for r, h in handles.items():
            self.main[r] = self.rt.stream()
            self.panel[r] = self.rt.stream(high_priority=True)
            self.commst[r] = self.rt.stream()
            h.set_streams(self.main[r].cuda_stream, self.panel[r].cuda_stream)
        nmax = max(r_ * c_ for r_, c_ in self.shapes)
        # three rotating packed-panel buffers per local rank: panel k+1 lands while panel k is read
        tdtype = torch.float64 if any_h.dtype == 64 else torch.float32
        self.bufs = {r: [torch.empty(nmax, dtype=tdtype, device=self.device) for _ in range(3)]
                     for r in handles}

    def owner(self, k: int) -> int:
        return k % self.world

    def loglik(self, theta):
        """theta -> (loglik, logdet, quad, info); identical on every rank."""
        torch = self.torch
        H, nouter = self.handles, self.nouter
        ev_rest = {r: [None] * (nouter + 1) for r in H}        # REST(k) done on rank r's main stream
        ev_bcast = {r: [None] * (nouter + 1) for r in H}       # packed panel k ready in bufs[r][k % 3]
        ev_first = {r: [None] * (nouter + 1) for r in H}       # REST(k)'s piece on outer panel k+2 done (its owner only)

        def view(r, k):
            rows, cols = self.shapes[k]
            return self.bufs[r][k % 3][: rows * cols]

        def factor_and_broadcast(k):
            """owner factors + packs panel k (panel stream); everyone receives it (comm stream)."""
            o = self.owner(k)
            packed_ev = None
            if o in H:
                with self.rt.on(self.panel[o]):
                    H[o].dist_factor_panel(k, view(o, k))
                    packed_ev = self.rt.event()
                    packed_ev.record(self.panel[o])
            # every local rank's comm stream: wait until its copy of the buffer is free (REST(k-3)
            # was its last reader) and, on the owner, until the pack has finished
            for r in H:
                if k >= 3 and ev_rest[r][k - 3] is not None:
                    self.commst[r].wait_event(ev_rest[r][k - 3])
                if r == o:
                    self.commst[r].wait_event(packed_ev)
                elif o in H:                                   # loopback: the copy reads the owner's buffer
                    self.commst[r].wait_event(packed_ev)
            if len(H) == 1:
                r = next(iter(H))
                with self.rt.on(self.commst[r]):
                    work = self.comm.broadcast({r: view(r, k)}, o)
                    if work is not None:
                        work.wait()                            # comm stream waits for RCCL; host does not block
                    ev = self.rt.event()
                    ev.record(self.commst[r])
                    ev_bcast[r][k] = ev
            else:                                              # loopback: copies issued from each receiver's stream
                for r in H:
                    with self.rt.on(self.commst[r]):
                        if r != o:
                            view(r, k).copy_(view(o, k), non_blocking=True)
                        ev = self.rt.event()
                        ev.record(self.commst[r])
                        ev_bcast[r][k] = ev

        for r, h in H.items():
            with self.rt.on(self.main[r]):
                h.dist_begin(theta, r, self.world)
                built = self.rt.event()
                built.record(self.main[r])
            self.panel[r].wait_event(built)
        factor_and_broadcast(0)
        for k in range(nouter):
            if k + 1 < nouter:
                o = self.owner(k + 1)
                if o in H:                                     # LA(k) on the owner's panel stream
                    self.panel[o].wait_event(ev_bcast[o][k])
                    if k >= 1:
                        self.panel[o].wait_event(ev_first[o][k - 1])
                    with self.rt.on(self.panel[o]):
                        H[o].dist_update(k, view(o, k), k + 1, k + 2, True)
                factor_and_broadcast(k + 1)
            for r, h in H.items():                             # REST(k) on everyone's main stream
                self.main[r].wait_event(ev_bcast[r][k])
                with self.rt.on(self.main[r]):
                    # (the last panel has no look-ahead step: its REST starts at the corner tile)
                    first = k + 2 if k + 1 < nouter else k + 1
                    if self.world > 1 and k + 2 < nouter and r == self.owner(k + 2):
                        h.dist_update(k, view(r, k), k + 2, k + 3, False)
                        ev_first[r][k] = self.rt.event()
                        ev_first[r][k].record(self.main[r])
                        first = k + 3
                    h.dist_update(k, view(r, k), first, nouter + 1, False)
                    ev = self.rt.event()
                    ev.record(self.main[r])
                    ev_rest[r][k] = ev
                    if ev_first[r][k] is None:
                        ev_first[r][k] = ev
        vals = {}
        for r, h in H.items():
            self.main[r].wait_stream(self.panel[r])
            logdet_part, quad, info = h.dist_end()
            vals[r] = [logdet_part, quad, float(info)]
        tot = self.comm.allreduce_scalars(vals)
        logdet, quad, info = float(tot[0]), float(tot[1]), int(tot[2])
        ll = -0.5 * (self.N * LOG_TWO_PI + logdet + quad)
        if info == 0 and not math.isfinite(ll):
            info = 2
        return ll, logdet, quad, info
