"""Multi-GPU partitioning of the GP path: one process per GPU.

The path shards over *independent units* -- hyper-parameter points theta (the nested-sampling
sweep BS:902-916 and the replicas of parallelNestedSampling BS:1349-1357), posterior samples and
test points (prediction, BGP:355-372) -- so units are dealt round-robin to ranks and there is NO
data-path collective.  torch.distributed (RCCL on GPUs, gloo in the CPU tests) only merges the
small result vectors on the host side, the way the reference merges ParallelTable results
(combineRuns, BS:1293-1315).  Training data are replicated: every rank regenerates / receives the
same X, y (N*d*8 bytes, 2 MB at N=32768).

`block_cyclic_owner` is the column map of the 1-D block-cyclic distributed Cholesky (§8e(3)),
kept here so the host logic is unit-tested on CPU.
"""
from __future__ import annotations

import numpy as np


def shard_indices(n_items: int, rank: int, world: int) -> np.ndarray:
    """Round-robin deal: unit i goes to rank i % world (keeps cheap/expensive theta mixed)."""
    return np.arange(rank, n_items, world)


def block_cyclic_owner(block_col: int, world: int) -> int:
    """Owner of 128-tile block column j in the 1-D block-cyclic layout."""
    return block_col % world


def local_block_columns(n_block_cols: int, rank: int, world: int) -> np.ndarray:
    return np.arange(rank, n_block_cols, world)


def sharded_map(evaluate, items: np.ndarray, dist=None):
    """Evaluate `evaluate(items[idx]) -> (values[len(idx)], info[len(idx)])` on this rank's shard and
    return the merged full-length (values, info) on every rank.  `dist` is torch.distributed (or
    None for a single process).  The merge is one all_gather of a padded [ceil(n/world), 2] block."""
    items = np.asarray(items, dtype=np.float64)
    n = len(items)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        vals, info = evaluate(items)
        return np.asarray(vals, dtype=np.float64), np.asarray(info, dtype=np.int64)
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    idx = shard_indices(n, rank, world)
    vals, info = evaluate(items[idx]) if len(idx) else (np.zeros(0), np.zeros(0, dtype=np.int64))
    per = (n + world - 1) // world
    buf = torch.zeros(per, 2, dtype=torch.float64)
    buf[:len(idx), 0] = torch.as_tensor(np.asarray(vals, dtype=np.float64))
    buf[:len(idx), 1] = torch.as_tensor(np.asarray(info, dtype=np.float64))
    backend = dist.get_backend()
    if backend == "nccl":
        buf = buf.cuda()
    gathered = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(gathered, buf)
    out_v = np.zeros(n)
    out_i = np.zeros(n, dtype=np.int64)
    for r in range(world):
        ridx = shard_indices(n, r, world)
        g = gathered[r].cpu().numpy()
        out_v[ridx] = g[:len(ridx), 0]
        out_i[ridx] = g[:len(ridx), 1].astype(np.int64)
    return out_v, out_i


def shard_slice(n_items: int, rank: int, world: int) -> slice:
    """Contiguous shard (for test points: keeps each rank's chunk coalesced)."""
    per = (n_items + world - 1) // world
    return slice(min(rank * per, n_items), min((rank + 1) * per, n_items))


def sharded_predict(predict, Xs: np.ndarray, dist=None):
    """Test-point sharding of the prediction: every rank holds the full factor -- either each
    rank fitted on its own (small N: replicas) or the ranks fitted ONE factorisation together through per-rank
    handles (`_lib.Handle(.., rank=, world=, comm_id=)`: the collective gphip_fit unpacks every received panel, so
    L ends up replicated with no extra traffic) -- predicts its contiguous shard of the test points with
    `predict(Xs_shard) -> (mean, var)` and the shards are concatenated on every rank with one all_gather of the
    RESULTS.  No collective inside the solve.  (A multi-device handle in ONE process shards test points by itself
    inside gphip_predict.)"""
    Xs = np.atleast_2d(np.asarray(Xs, dtype=np.float64))
    m = len(Xs)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return predict(Xs)
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    sl = shard_slice(m, rank, world)
    mean, var = predict(Xs[sl]) if sl.stop > sl.start else (np.zeros(0), np.zeros(0))
    per = (m + world - 1) // world
    buf = torch.zeros(per, 2, dtype=torch.float64)
    buf[:len(mean), 0] = torch.as_tensor(np.asarray(mean, dtype=np.float64))
    buf[:len(mean), 1] = torch.as_tensor(np.asarray(var, dtype=np.float64))
    if dist.get_backend() == "nccl":
        buf = buf.cuda()
    gathered = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(gathered, buf)
    out_m, out_v = np.zeros(m), np.zeros(m)
    for r in range(world):
        rs = shard_slice(m, r, world)
        g = gathered[r].cpu().numpy()
        out_m[rs] = g[:rs.stop - rs.start, 0]
        out_v[rs] = g[:rs.stop - rs.start, 1]
    return out_m, out_v
