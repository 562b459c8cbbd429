"""Data-parallel retrieval: clip sharding, RCCL all-gather of the product-descriptor bank,
street-sequence vs. bank match + top-k  (BASELINE.json north_star).

The reference has no counterpart for the collective (its "multi-GPU" is N independent
processes, ref train_movingfashion.py:46-50; the gallery is a Python list concatenated with
NumPy on rank 0, ref evaluate_movingfashion.py:45-47,82,92).  Here:

  * clips are independent            -> rank r owns clips  r, r+W, r+2W, ...   (no collective)
  * gallery products are independent -> rank r *computes* descriptors for products
    [lo_r, hi_r) (each shop image goes through the extractor once, somewhere)
  * the only exchange: every query sequence is matched against EVERY product, so the
    descriptor shards are all-gathered into the full bank [G,256] on every rank -- one
    ``all_gather_into_tensor`` (backend "nccl" == RCCL over xGMI), issued on a side stream so
    it overlaps the street-side trunk / NLB work, then the local pairwise + top-k kernels run.

Everything here works on CPU tensors with the gloo backend too (used by the world_size=2
tests for the sharding / gather logic; the match itself always needs the HIP kernels).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo,hi) of n items for ``rank``; the first n % world ranks get one extra."""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def clips_for_rank(n_clips: int, rank: int, world: int) -> List[int]:
    """Round-robin clip ownership: r, r+W, r+2W, ..."""
    return list(range(rank, n_clips, world))


def _world(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


class BankGather:
    """Handle of an in-flight all-gather of the product bank (``wait()`` -> [G,256]).
    ``events`` (when the gather was issued with ``timed=True`` on a HIP stream): (start, end) events recorded on the
    stream the collective runs on, for ``elapsed_us()`` after a synchronisation."""

    def __init__(self, bank: torch.Tensor, g_total: int, work=None, stream=None, pad_rows: int = 0,
                 sizes: Optional[List[int]] = None, events=None):
        self.bank, self.g_total, self.work, self.stream = bank, g_total, work, stream
        self.pad_rows, self.sizes, self.events = pad_rows, sizes, events

    def wait(self) -> torch.Tensor:
        if self.work is not None:
            self.work.wait()
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)
        if self.sizes is not None:      # ragged shards were padded to the largest: compact
            m = max(self.sizes)
            parts = [self.bank[i * m:i * m + s] for i, s in enumerate(self.sizes)]
            return torch.cat(parts, 0)
        return self.bank

    def elapsed_us(self) -> Optional[float]:
        """Device time of the collective (call after the stream was synchronised); None when not timed."""
        if self.events is None:
            return None
        return 1e3 * self.events[0].elapsed_time(self.events[1])


def _gather_into_tensor(group=None, device_type: str = "cuda") -> bool:
    """Which collective form to issue -- decided from the process group's backend CONFIGURATION and the shard's device type
    (both rank-invariant), so that every rank issues the same call (a per-rank try/except could pair
    ``all_gather_into_tensor`` on one rank with ``all_gather`` on another and deadlock): a CUDA shard in a group whose cuda
    backend is RCCL ("nccl", also as part of "cpu:gloo,cuda:nccl" or a group initialised without an explicit backend) gathers
    into one tensor; anything else (gloo, CPU tests) uses the list form."""
    if device_type != "cuda":
        return False
    try:
        cfg = str(dist.get_backend_config(group)).lower()            # e.g. "cuda:nccl" / "cpu:gloo,cuda:nccl" / "cpu:gloo"
    except (AttributeError, RuntimeError, ValueError):
        cfg = str(dist.get_backend(group)).lower()
    parts = dict(p.split(":", 1) for p in cfg.split(",") if ":" in p)
    return parts.get("cuda", cfg) == "nccl" or cfg == "nccl"


def gather_product_bank(local: torch.Tensor, g_total: int, group=None, side_stream=None, timed: bool = False,
                        force: bool = False) -> BankGather:
    """All-gather the per-rank descriptor shards ``local[g_r,256]`` (g_r from ``shard_range``)
    into the full bank, product order preserved.  Asynchronous: returns a ``BankGather``.

    side_stream: optional ``torch.cuda.Stream``; the collective is enqueued there (after the
    producer of ``local`` on the current stream) so compute on the current stream overlaps it.
    timed: bracket the collective with HIP events on the stream it runs on (``BankGather.elapsed_us``).
    force: issue the collective even in a one-rank process group (exercises the N > 1 code path on a single GPU)."""
    world = _world(group)
    if world == 1 and not (force and dist.is_available() and dist.is_initialized()):
        return BankGather(local, g_total)
    rank = dist.get_rank(group)
    sizes = [shard_range(g_total, r, world)[1] - shard_range(g_total, r, world)[0] for r in range(world)]
    assert local.shape[0] == sizes[rank], (local.shape, sizes, rank)
    m = max(sizes)
    ragged = min(sizes) != m
    src = local.contiguous()
    if ragged:
        pad = local.new_zeros((m, local.shape[1]))
        pad[:src.shape[0]] = src
        src = pad
    bank = local.new_empty((world * m, local.shape[1]))
    into = _gather_into_tensor(group, local.device.type)

    def issue():
        if into:
            return dist.all_gather_into_tensor(bank, src, group=group, async_op=True)
        return dist.all_gather(list(bank.view(world, m, -1).unbind(0)), src, group=group, async_op=True)

    if local.is_cuda and side_stream is not None:
        side_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side_stream):
            ev = None
            if timed:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            work = issue()
            if timed:
                try:
                    work.wait()         # stream-level wait (no host block): orders the end event behind the collective
                    ev[1].record()
                except RuntimeError:    # a backend without stream-level waits: the gather itself is unaffected, only untimed
                    ev = None
        return BankGather(bank, g_total, work, side_stream, sizes=sizes if ragged else None, events=ev)
    return BankGather(bank, g_total, issue(), None, sizes=sizes if ragged else None)


@torch.no_grad()
def match_sequences(aggregator, x3_1b: torch.Tensor, bank: torch.Tensor, k: int = 20):
    """Aggregated street descriptors [S,256] vs the full bank [G,256] with the aggregator's
    pairwise classifier (ref models/match_head.py:161-162), then score + rank
    (ref evaluate_movingfashion.py:263-269).  -> (x5 [S,G,2], idx [S,k] int64, score [S,k])."""
    from . import ops
    x5 = aggregator.pair(x3_1b, bank)
    idx, score = ops.rank_topk(x5, min(k, bank.shape[0]))
    return x5, idx, score


@torch.no_grad()
def match_sequences_topk(aggregator, x3_1b: torch.Tensor, bank: torch.Tensor, k: int = 20):
    """Same ranking as ``match_sequences`` without materialising the full [S,G,2] logits, for large galleries (configs 3/4:
    G = 20 000 / 50 000): ``ops.pair_topk`` -- banks of >= 8192 products go through the MFMA similarity + fused top-k
    (seam_pair_topk_mfma_f32: candidates from the logit-difference GEMM, exact re-scoring, rounding-error proof), smaller ones
    through query chunks of seam_pair_logits_f32 + seam_rank_topk_f32; bit-identical either way.
    -> (idx [S,k] int64, score [S,k])."""
    from . import ops
    return ops.pair_topk(x3_1b, bank, aggregator.last.weight, aggregator.last.bias, min(k, bank.shape[0]))


# ---------------------------------------------------------------------------------------------------
# Evaluator-side retrieval on the device in fp32: the closures of
# evaluate_movingfashion.py:94-121 (NumPy fp16, full argsort per query on the CPU) as kernel calls.
@torch.no_grad()
def compute_distances(street: torch.Tensor, shop: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """``compute_distances`` / ``compute_selfdist`` (pass shop=street): softmax((shop-street)^2 @ w.T + b)[...,1]
    -> [n_street, n_shop]  (ref evaluate_movingfashion.py:102-107,115-121)."""
    from . import ops
    return ops.match_scores(ops.pair_logits(street, shop, w, b))


@torch.no_grad()
def compute_rank_of(street: torch.Tensor, shop: torch.Tensor, w: torch.Tensor, b: torch.Tensor,
                    target: torch.Tensor) -> torch.Tensor:
    """Rank of the true product of each street descriptor: what the evaluator extracts from
    ``compute_ranking(inds)`` with ``(rankings == shop_prod_index).nonzero()`` (ref :94-100,228,268); the
    top-k accuracies only test ``rank < k`` for k in {1,5,10,20} (ref :15,229-231)."""
    from . import ops
    return ops.rank_of(ops.pair_logits(street, shop, w, b), target.to(torch.int64))
