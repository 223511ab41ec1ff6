"""Sequence (frame) parallelism of one clip across the GPUs of a node: one process per GPU,
torch.distributed (backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests).

New design (the reference has no distributed code).  Every DiT op is token-local
except self-attention, so the token rows [S, D] are cut into `world` contiguous bands (token order is
(T H W), hence a band is a run of latent frames when world divides F) and exactly one exchange per
self-attention block is needed: the all-gather of the post-RMSNorm/RoPE K and V rows.  The bands are
gathered rank-major, which IS the global token order, so no permutation follows.

Two exchanges implement that one dependency:

* head <-> token all-to-all (default when `world` divides the head count): after the fused QKV projection of its token
  band a rank sends every peer the q|k|v columns of THAT peer's heads and receives all S tokens of its own heads;
  attention then runs over (heads/world) complete heads, and a second all-to-all returns the outputs to token bands.
  Per layer and rank 7/8 * (3 + 1) * S/world * D * 2 B = 66 MB leave over the 7 xGMI links (9.4 MB per link) - a quarter
  of the all-gather's traffic, in the pattern a point-to-point mesh serves best (every link busy, one hop).
* K|V all-gather (fallback, any world that divides S): every rank receives all other bands' K and V rows
  (S * 2D * 2 B * 7/8 = 264 MB per layer and rank).
"""
from typing import Tuple

import torch
import torch.distributed as dist


class ShardPlan:
    """Token-band decomposition of S rows over `world` ranks."""

    def __init__(self, S: int, rank: int = 0, world: int = 1):
        if S % world != 0:
            raise ValueError(f"token count {S} is not divisible by world size {world}")
        self.S, self.rank, self.world = S, rank, world
        self.rows = S // world
        self.start = rank * self.rows
        self.stop = self.start + self.rows

    def band(self, t: torch.Tensor) -> torch.Tensor:
        """Rows of a full [S, ...] tensor owned by this rank (a view)."""
        return t[self.start:self.stop]


def group_info(group=None) -> Tuple[int, int]:
    if group is None and not (dist.is_available() and dist.is_initialized()):
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


# tests set this to run the collectives of a 1-rank group too (RCCL call path on a single GPU)
SINGLE_RANK_COLLECTIVES = False


class ExchangeTimer:
    """How long the compute stream sits in front of an exchange (bench.py, N > 1): an event pair around every wait() of an
    asynchronous collective.  work.wait() makes the CURRENT stream wait for RCCL's stream, so the time between the two events is
    the part of the exchange that compute did not cover ("exposed"); an exchange that has already finished costs ~0."""

    def __init__(self, sample_every: int = 7):
        """Only every `sample_every`-th wait of a tag is bracketed (an event pair costs ~35 us of queue time on the timing rank;
        all waits of a forward would add ~3 ms to rank 0's step and skew the max-over-ranks time); use a stride coprime with the
        layer count."""
        self.sample_every = max(1, int(sample_every))
        self.counts = {}
        self.records = []          # (tag, start_event, end_event)

    def take(self, tag) -> bool:
        c = self.counts.get(tag, 0)
        self.counts[tag] = c + 1
        return c % self.sample_every == 0

    def summary(self):
        """tag -> waits seen, waits timed, mean exposed ms per wait (of the timed ones)."""
        out = {}
        for tag, a, b in self.records:
            d = out.setdefault(tag, {"timed": 0, "ms_total": 0.0})
            d["timed"] += 1
            d["ms_total"] += a.elapsed_time(b)
        for tag, d in out.items():
            d["waits"] = self.counts.get(tag, d["timed"])
            d["ms_avg"] = d["ms_total"] / max(d["timed"], 1)
        return out


_XTIMER = None


def set_exchange_timer(t):
    global _XTIMER
    _XTIMER = t


def wait_exchange(work, tag: str):
    """work.wait() (None = nothing was issued), timed when an ExchangeTimer is installed."""
    if work is None:
        return
    if _XTIMER is None or not torch.cuda.is_available() or not _XTIMER.take(tag):
        work.wait()
        return
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    work.wait()
    b.record()
    _XTIMER.records.append((tag, a, b))


def allgather_rows_(full: torch.Tensor, plan: ShardPlan, group=None, async_op: bool = False):
    """In-place all-gather: `full` is [S, C] contiguous, this rank's band already holds its rows."""
    if plan.world == 1 and not (SINGLE_RANK_COLLECTIVES and group is not None):
        return None
    assert full.is_contiguous() and full.shape[0] == plan.S
    if dist.get_backend(group) == "gloo" and full.is_cuda:
        # test-only path (two ranks sharing one GPU cannot use RCCL): gloo gathers device tensors through host staging
        parts = [torch.empty_like(plan.band(full)) for _ in range(plan.world)]
        dist.all_gather(parts, plan.band(full).clone(), group=group)
        for r, p in enumerate(parts):
            full[r * plan.rows:(r + 1) * plan.rows].copy_(p)
        return None
    return dist.all_gather_into_tensor(full, plan.band(full), group=group, async_op=async_op)


def allgather_rows(local: torch.Tensor, plan: ShardPlan, group=None) -> torch.Tensor:
    """Out-of-place variant: local [S/world, C] -> [S, C]."""
    if plan.world == 1:
        return local
    full = torch.empty((plan.S,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(full, local.contiguous(), group=group)
    return full


def alltoall_rows_(send: torch.Tensor, recv: torch.Tensor, group=None, async_op: bool = False):
    """Equal-split all-to-all: send [world, n, C] (slab r goes to rank r) -> recv [world, n, C] (slab r came from rank r)."""
    assert send.is_contiguous() and recv.is_contiguous() and send.shape == recv.shape
    if dist.get_backend(group) == "gloo" and send.is_cuda:
        # test-only path (two ranks sharing one GPU cannot use RCCL): host staging
        out = torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_to_all_single(out, send.cpu(), group=group)
        recv.copy_(out)
        return None
    return dist.all_to_all_single(recv, send, group=group, async_op=async_op)


def alltoall_bands_(send: torch.Tensor, recv: torch.Tensor, lo: int, hi: int, group=None, async_op: bool = False):
    """The part of alltoall_rows_(send, recv) whose DESTINATION ranks are lo <= r < hi: every rank sends slab r to rank r for those
    r only; a rank inside [lo, hi) receives its slab from every peer (all of `recv`), the others receive nothing.  The parts
    over a partition of the ranks add up to the whole exchange - used to return the attention output of the token bands whose
    queries are finished while the attention of the remaining bands still runs (dit_engine, head <-> token exchange)."""
    assert send.is_contiguous() and recv.is_contiguous() and send.shape == recv.shape
    world, n = send.shape[0], send.shape[1]
    rank = dist.get_rank(group)
    assert 0 <= lo <= hi <= world
    if lo == hi:
        return None
    tail = tuple(send.shape[2:])
    inp = send.view(world * n, *tail)[lo * n:hi * n]
    mine = lo <= rank < hi
    out = recv.view(world * n, *tail) if mine else recv.view(world * n, *tail)[:0]
    in_splits = [n if lo <= r < hi else 0 for r in range(world)]
    out_splits = [n if mine else 0] * world
    if dist.get_backend(group) == "gloo" and send.is_cuda:
        host = torch.empty(out.shape, dtype=recv.dtype)                # test-only path, see alltoall_rows_
        dist.all_to_all_single(host, inp.cpu(), out_splits, in_splits, group=group)
        if mine:
            out.copy_(host)
        return None
    return dist.all_to_all_single(out, inp, out_splits, in_splits, group=group, async_op=async_op)


def allgather_stack(local: torch.Tensor, group=None) -> torch.Tensor:
    """local [k, ...] on every rank (same k) -> [world, k, ...] in rank order."""
    world = dist.get_world_size(group)
    local = local.contiguous()
    out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
    if dist.get_backend(group) == "gloo" and local.is_cuda:
        parts = [torch.empty_like(local) for _ in range(world)]       # test-only path, see allgather_rows_
        dist.all_gather(parts, local, group=group)
        for r, p in enumerate(parts):
            out[r].copy_(p)
        return out
    dist.all_gather_into_tensor(out, local, group=group)
    return out
