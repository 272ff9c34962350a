"""Multi-GPU driver: independent units (cells or hyperparameter-grid points) sharded across
the GPUs of one node, one process per GPU.

The path has NO data-path collective: the only communication is one broadcast of the shared
stimulus matrix X from rank 0 (RCCL over xGMI; ``backend="nccl"`` is RCCL on ROCm) before the
work starts and one small all-gather of ``(loss, grad[6])`` per unit afterwards.  X is
N*d*8 B (16 MiB at N=8192, d=256): latency-bound, a flat broadcast.

The same code runs on CPU with the ``gloo`` backend (tests/test_distributed_cpu.py)."""
from __future__ import annotations

from typing import Callable, List, Sequence

import torch
import torch.distributed as dist

RESULT_WIDTH = 7  # loss + 6 gradients (theta dict order)


def partition(n_units: int, world: int, rank: int) -> List[int]:
    """Static cyclic assignment u -> rank (u mod world): 64 cells on 8 GPUs = 8 each."""
    return [u for u in range(n_units) if u % world == rank]


def _single(force_collectives: bool) -> bool:
    """True when there is nothing to communicate: no process group, or a group of one rank -- unless the caller
    asks for the collectives anyway (``force_collectives``: a world-size-1 ``nccl`` group on a one-GPU box then
    runs exactly the RCCL calls of the multi-GPU job, which is how tests/test_gpu_rccl.py exercises them)."""
    if not dist.is_initialized():
        return True
    return dist.get_world_size() == 1 and not force_collectives


def broadcast_stimuli(X: torch.Tensor | None, shape, device, src: int = 0, force_collectives: bool = False) -> torch.Tensor:
    """Rank ``src`` passes X; every other rank passes None and receives a copy."""
    if _single(force_collectives):
        assert X is not None
        return X
    if dist.get_rank() != src:
        X = torch.empty(shape, dtype=torch.float64, device=device)
    else:
        X = X.to(device=device, dtype=torch.float64).contiguous()
    dist.broadcast(X, src=src)
    return X


def broadcast_state(r, m, V, n: int, device, dtype=torch.float64, src: int = 0, force_collectives: bool = False):
    """Per-cell state shared by every unit of a hyperparameter grid (BASELINE configs[4]: 512 theta
    points, one cell): ``r``, ``m`` (n each) and ``V`` (n x n) from rank ``src`` to all ranks.

    ``r`` and ``m`` are KiB-sized: one flat broadcast each.  ``V`` is bandwidth-bound (256 MiB at
    N=8192 fp32, 512 MiB fp64) and xGMI is point-to-point, so a ring / tree broadcast would be
    bound by one ~153 GB/s link; instead rank ``src`` SCATTERS ``world`` row blocks (each of its
    seven links carries 1/world of the payload) and the ranks ALL-GATHER them (every link carries
    one block in each direction): SURVEY.md section 5 / 8(e).  Ranks other than ``src`` pass None."""
    if _single(force_collectives):
        assert r is not None and m is not None and V is not None
        return r.to(device=device, dtype=dtype), m.to(device=device, dtype=dtype), V.to(device=device, dtype=dtype)
    world, rank = dist.get_world_size(), dist.get_rank()
    vecs = torch.empty((2, n), dtype=dtype, device=device)
    if rank == src:
        vecs[0], vecs[1] = r.to(device=device, dtype=dtype), m.to(device=device, dtype=dtype)
    dist.broadcast(vecs, src=src)
    rows = (n + world - 1) // world                      # row block per rank (the last one zero padded)
    block = torch.empty((rows, n), dtype=dtype, device=device)
    if rank == src:
        Vp = torch.zeros((rows * world, n), dtype=dtype, device=device)
        Vp[:n] = V.to(device=device, dtype=dtype)
        dist.scatter(block, [Vp[i * rows:(i + 1) * rows] for i in range(world)], src=src)
        del Vp
    else:
        dist.scatter(block, None, src=src)
    full = torch.empty((rows * world, n), dtype=dtype, device=device)
    if dist.get_backend() == "nccl":
        dist.all_gather_into_tensor(full, block)
    else:
        parts = [full[i * rows:(i + 1) * rows] for i in range(world)]
        gathered = [torch.empty_like(block) for _ in range(world)]
        dist.all_gather(gathered, block)
        for dst, g in zip(parts, gathered):
            dst.copy_(g)
    return vecs[0].contiguous(), vecs[1].contiguous(), full[:n]


def evaluate_units(units: Sequence[int], eval_fn: Callable[[int], Sequence[float]], device) -> torch.Tensor:
    """Run ``eval_fn(u) -> (loss, g0..g5)`` for the local units, back to back."""
    out = torch.zeros((len(units), RESULT_WIDTH), dtype=torch.float64, device=device)
    for i, u in enumerate(units):
        vals = eval_fn(u)
        out[i] = torch.as_tensor(list(vals), dtype=torch.float64, device=device)
    return out


def evaluate_units_pipelined(units: Sequence[int], submit_fn: Callable[[int, int], object],
                             collect_fn: Callable[[object, int], Sequence[float]], device, depth: int = 2,
                             lockstep: bool = False) -> torch.Tensor:
    """Keep ``depth`` independent units in flight: ``submit_fn(u, slot)`` enqueues unit u on the
    engine / stream of ``slot`` and returns a ticket, ``collect_fn(ticket, slot)`` waits for it and
    returns ``(loss, g0..g5)``.  Units are independent, so one unit's latency-bound
    Cholesky leaves and small panels overlap the next unit's large GEMMs.  Results are identical
    to :func:`evaluate_units` (each unit still runs alone on its own context)."""
    out = torch.zeros((len(units), RESULT_WIDTH), dtype=torch.float64, device=device)
    if lockstep:
        # groups of `depth` units submitted together and collected together: units of the same size
        # then run in phase (their leaf phases and their large GEMMs coincide) instead of staggered
        for g0 in range(0, len(units), depth):
            tickets = [(i, submit_fn(units[i], i - g0), i - g0) for i in range(g0, min(g0 + depth, len(units)))]
            for j, t, sl in tickets:
                out[j] = torch.as_tensor(list(collect_fn(t, sl)), dtype=torch.float64, device=device)
        return out
    inflight = []  # (index, ticket, slot)
    for i, u in enumerate(units):
        slot = i % depth
        if len(inflight) == depth:
            j, t, sl = inflight.pop(0)
            out[j] = torch.as_tensor(list(collect_fn(t, sl)), dtype=torch.float64, device=device)
        inflight.append((i, submit_fn(u, slot), slot))
    for j, t, sl in inflight:
        out[j] = torch.as_tensor(list(collect_fn(t, sl)), dtype=torch.float64, device=device)
    return out


def evaluate_units_grouped(units: Sequence[int], group_fn: Callable[[Sequence[int]], Sequence[Sequence[float]]], device,
                           group: int, begin_fn=None, finish_fn=None, sets: int = 1) -> torch.Tensor:
    """Evaluate the local units ``group`` at a time: ``group_fn(us) -> [(loss, g0..g5) for u in us]`` runs one
    grouped call (``engine.fit_eval_group`` / ``gpfit_fit_eval_batch``: the Cholesky recursions of the whole group
    in lock step, the products behind them as batched launches).  Units are independent and every
    unit's numbers are bit-identical to its own evaluation, so the table equals :func:`evaluate_units`'.

    With ``begin_fn(us, slot) -> handle`` / ``finish_fn(handle, slot) -> rows`` and ``sets`` >= 2 sets of engines
    (``slot`` = which set) the groups are pipelined one deep: group k + 1 is enqueued before group k is collected,
    so the host's share of a group (argument marshalling, result collection: about 5 ms per group of 16) runs
    beside the previous group instead of between two groups.  The table is assembled on the host and moved to the
    device once."""
    if group < 1:
        raise ValueError("group must be at least 1")
    table = [None] * len(units)

    def store(g0, n_us, rows):
        if len(rows) != n_us:
            raise RuntimeError(f"the group call returned {len(rows)} results for {n_us} units")
        for j, row in enumerate(rows):
            table[g0 + j] = [float(v) for v in row]

    split = begin_fn is not None and finish_fn is not None
    if group_fn is None:
        if not split:
            raise ValueError("evaluate_units_grouped needs group_fn, or begin_fn together with finish_fn")
        group_fn = lambda us: finish_fn(begin_fn(us, 0), 0)      # one set of engines: begin and collect back to back
    pipelined = split and sets >= 2
    pending = None
    try:
        for gi, g0 in enumerate(range(0, len(units), group)):
            us = list(units[g0:g0 + group])
            if not pipelined:
                store(g0, len(us), group_fn(us))
                continue
            handle = begin_fn(us, gi % sets)
            done, pending = pending, (handle, g0, len(us), gi % sets)
            if done is not None:
                store(done[1], done[2], finish_fn(done[0], done[3]))
        if pending is not None:
            done, pending = pending, None
            store(done[1], done[2], finish_fn(done[0], done[3]))
    finally:
        # an exception above (begin_fn, finish_fn or a malformed result) must not leave a group enqueued: its
        # contexts would refuse every later call ("an asynchronous evaluation is pending")
        if pending is not None:
            try:
                finish_fn(pending[0], pending[3])
            except Exception:
                pass
    if not table:
        return torch.zeros((0, RESULT_WIDTH), dtype=torch.float64, device=device)
    return torch.tensor(table, dtype=torch.float64).to(device)


def gather_results(local: torch.Tensor, n_units: int, force_collectives: bool = False) -> torch.Tensor:
    """All ranks receive the [n_units, 7] table in unit order."""
    if _single(force_collectives):
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    per = (n_units + world - 1) // world
    pad = torch.zeros((per, RESULT_WIDTH), dtype=torch.float64, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    table = torch.zeros((n_units, RESULT_WIDTH), dtype=torch.float64, device=local.device)
    for rk in range(world):
        for i, u in enumerate(partition(n_units, world, rk)):
            table[u] = parts[rk][i]
    return table


def run_sharded(n_units: int, eval_fn: Callable[[int], Sequence[float]], device, submit_fn=None, collect_fn=None,
                depth: int = 2, lockstep: bool = False, group_fn=None, group: int = 0, begin_fn=None, finish_fn=None,
                sets: int = 1, force_collectives: bool = False) -> torch.Tensor:
    """Evaluate all units, sharded cyclically over the ranks; with ``group_fn`` the local units go ``group`` at a
    time through one grouped call each (:func:`evaluate_units_grouped`; ``begin_fn`` / ``finish_fn`` / ``sets``: its
    pipelined form); with ``submit_fn`` / ``collect_fn`` they
    are pipelined ``depth`` deep (see :func:`evaluate_units_pipelined`)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    mine = partition(n_units, world, rank)
    if group_fn is not None or begin_fn is not None:
        local = evaluate_units_grouped(mine, group_fn, device, max(1, group), begin_fn, finish_fn, sets)
    elif submit_fn is not None and collect_fn is not None:
        local = evaluate_units_pipelined(mine, submit_fn, collect_fn, device, depth, lockstep)
    else:
        local = evaluate_units(mine, eval_fn, device)
    return gather_results(local, n_units, force_collectives)
