"""CPU, world_size=2 over gloo: the data-parallel sharding + product-bank all-gather path
(seam-match-rcnn_amd/retrieval.py).  The match itself needs the GPU kernels; here the gathered
bank is checked for exact content/order and the per-rank match is checked with the oracle."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, g_total, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import seam_match_rcnn_amd.synth as synth
        from seam_match_rcnn_amd import retrieval
        from oracle import heads as OH
        full = torch.from_numpy(synth.gallery(7, g_total))
        lo, hi = retrieval.shard_range(g_total, rank, world)
        bank = retrieval.gather_product_bank(full[lo:hi].clone(), g_total).wait()
        ok = bank.shape == full.shape and torch.equal(bank, full)
        # each rank matches its own clips (round robin) against the gathered bank
        clips = retrieval.clips_for_rank(5, rank, world)
        w = torch.from_numpy(synth.normal(synth.stream_id(1, "w"), (2, 256))) / 16
        b = torch.zeros(2)
        tops = {}
        for c in clips:
            qd = torch.from_numpy(synth.normal(synth.stream_id(100 + c, "q"), (3, 256)))
            idx, _ = OH.rank_topk(OH.pair_logits(qd, bank, w, b), 5)
            ref, _ = OH.rank_topk(OH.pair_logits(qd, full, w, b), 5)
            ok = ok and torch.equal(idx, ref)
            tops[c] = idx
        q.put((rank, bool(ok), clips))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("g_total", [1000, 37])      # equal shards / ragged shards
def test_bank_all_gather_world2(g_total):
    world = 2
    port = 29500 + (os.getpid() + g_total) % 2000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, g_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert sorted(c for _, _, cs in res for c in cs) == [0, 1, 2, 3, 4]


def _c4_worker(rank, world, port, q):
    """The wiring of bench.py's c4 step (configs[3]) on CPU: each rank owns clips r, r+W, ... and a contiguous shard of the
    product bank, gathers the bank, and ranks ITS sequences against the whole of it -- with the match stubbed by the oracle
    (the HIP kernels need a GPU).  Checked against a single-process run over all clips and the full bank."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import seam_match_rcnn_amd.synth as synth
        from seam_match_rcnn_amd import retrieval
        from oracle import heads as OH
        g_total, n_clips, seqs = 5003, 6, 4                   # ragged shards (5003 = 2 * 2501 + 1)
        full = torch.from_numpy(synth.gallery(7, g_total))
        lo, hi = retrieval.shard_range(g_total, rank, world)
        assert retrieval._gather_into_tensor() is False       # gloo -> list form, decided from the backend on every rank alike
        pending = retrieval.gather_product_bank(full[lo:hi].clone(), g_total)
        w = torch.from_numpy(synth.normal(synth.stream_id(2, "w"), (2, 256))) / 16
        b = torch.tensor([0.1, -0.2])
        mine = retrieval.clips_for_rank(n_clips, rank, world)
        desc = {c: torch.from_numpy(synth.normal(synth.stream_id(200 + c, "x3_1b"), (seqs, 256))) for c in mine}
        bank = pending.wait()                                   # awaited right before the match, as in the step
        assert pending.elapsed_us() is None                     # not timed on CPU
        res = {}
        for c, d in desc.items():
            idx, sc = OH.rank_topk(OH.pair_logits(d, bank, w, b), 20)
            ridx, rsc = OH.rank_topk(OH.pair_logits(d, full, w, b), 20)
            res[c] = bool(torch.equal(idx, ridx) and torch.equal(sc, rsc))
        q.put((rank, bool(torch.equal(bank, full)), res))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_c4_step_wiring_world2():
    world = 2
    port = 31500 + os.getpid() % 2000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_c4_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    clips = {}
    for _, _, r in res:
        clips.update(r)
    assert sorted(clips) == list(range(6)) and all(clips.values())


def test_single_process_gather_is_identity():
    sys.path.insert(0, ROOT)
    from seam_match_rcnn_amd import retrieval
    x = torch.arange(12.0).view(3, 4)
    assert retrieval.gather_product_bank(x, 3).wait() is x
