"""configs[2]/[3]: the MFMA similarity + fused top-k (seam_pair_topk_mfma_f32, csrc/seam_pairmf.hip) must return, bit for bit,
what seam_pair_logits_f32 + seam_rank_topk_f32 return (the direct form the oracle restates, ref models/match_head.py:161-162 +
evaluate_movingfashion.py:94-100) -- at the full sizes of the configs, on ragged shapes, with the proof-failed path forced, and
on degenerate banks (duplicates, constant rows, NaN) where ties at the k-th place make the error-bound proof fail on purpose."""
import numpy as np
import pytest
import torch

import seam_match_rcnn_amd.synth as synth
from conftest import to_torch
from oracle import heads as OH

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def wb():
    p = to_torch(synth.temporal_aggregator_state(12))
    return p["last.weight"].to(DEV), p["last.bias"].to(DEV)


def queries(seed, q):
    return torch.from_numpy(synth.normal(synth.stream_id(seed, "pairmf_q"), (q, 256))).to(DEV)


def both(a, b, w, bias, k, **kw):
    from seam_match_rcnn_amd import ops
    stats = torch.full((4,), -1, dtype=torch.int32, device=DEV)
    idx, sc = ops.pair_topk(a, b, w, bias, k, mfma=True, stats=stats, **kw)
    ridx, rsc = ops.pair_topk(a, b, w, bias, k, mfma=False)           # seam_pair_logits_f32 + seam_rank_topk_f32 in query chunks
    torch.cuda.synchronize()
    return idx, sc, ridx, rsc, stats.tolist()


@pytest.mark.parametrize("q,g,k", [(256, 20000, 20), (256, 50000, 20), (37, 20011, 20), (300, 8192, 1), (64, 9001, 64), (1, 12345, 5)])
def test_bit_identical_to_the_direct_form(wb, q, g, k):
    w, bias = wb
    a, b = queries(400 + q, q), torch.from_numpy(synth.gallery(500 + q, g)).to(DEV)
    idx, sc, ridx, rsc, stats = both(a, b, w, bias, k)
    assert torch.equal(idx, ridx) and torch.equal(sc, rsc)
    # continuous data: the error-bound proof holds for every query, no list overflows, lists stay near G * kk / 4096
    assert stats[0] == 0 and stats[2] == 0 and 0 < stats[1] < 8 * g * (k + 19) // 4096 + 256, stats


def test_against_the_cpu_oracle(wb):
    w, bias = wb
    a, b = queries(411, 16), torch.from_numpy(synth.gallery(511, 8192)).to(DEV)
    idx, sc, _, _, stats = both(a, b, w, bias, 20)
    oidx, osc = OH.rank_topk(OH.pair_logits(a.cpu(), b.cpu(), w.cpu(), bias.cpu(), chunk=8), 20)
    assert torch.equal(idx.cpu(), oidx)
    np.testing.assert_allclose(sc.cpu().numpy(), osc.numpy(), rtol=1e-4)


def test_forced_direct_form_path(wb):
    """flags bit 0: every query skips the candidates and ranks the whole bank with the direct form inside pairmf_final."""
    w, bias = wb
    a, b = queries(421, 40), torch.from_numpy(synth.gallery(521, 8200)).to(DEV)
    idx, sc, ridx, rsc, stats = both(a, b, w, bias, 20, force_exact=True)
    assert torch.equal(idx, ridx) and torch.equal(sc, rsc) and stats[0] == 40


def test_gallery_permutation_equivariance(wb):
    """Size-independent property at the config-3 size: permuting the bank permutes the indices and nothing else."""
    from seam_match_rcnn_amd import ops
    w, bias = wb
    a, b = queries(431, 256), torch.from_numpy(synth.gallery(531, 20000)).to(DEV)
    perm = torch.from_numpy(np.argsort(synth.uniform(synth.stream_id(532, "perm"), (20000,)), kind="stable")).to(DEV)
    idx, sc = ops.pair_topk(a, b, w, bias, 20, mfma=True)
    pidx, psc = ops.pair_topk(a, b[perm].contiguous(), w, bias, 20, mfma=True)
    assert torch.equal(perm[pidx], idx) and torch.equal(psc, sc)


def test_duplicates_constant_rows_and_nan(wb):
    """Ties at the k-th place defeat the proof (d of the k-th winner == d of an outsider): those queries must come back through
    the direct form with the lowest-index-first tie rule; a constant bank overflows every candidate list; NaN rows rank last."""
    w, bias = wb
    a = queries(441, 48)
    base = torch.from_numpy(synth.gallery(541, 300))
    b = base.repeat(30, 1)[:8999].contiguous().to(DEV)                # every product 30 times: 30-way ties everywhere
    idx, sc, ridx, rsc, stats = both(a, b, w, bias, 20)
    assert torch.equal(idx, ridx) and torch.equal(sc, rsc) and stats[0] > 0
    const = torch.from_numpy(synth.gallery(542, 1)).repeat(8192, 1).contiguous().to(DEV)
    idx, sc, ridx, rsc, stats = both(a, const, w, bias, 20)
    assert torch.equal(idx, ridx) and torch.equal(sc, rsc)
    assert torch.equal(idx[0].cpu(), torch.arange(20)) and stats[0] == 48 and stats[2] == 48
    b = torch.from_numpy(synth.gallery(543, 8192)).to(DEV)
    b[5::7] = float("nan")
    b[11, 3] = float("inf")
    idx, sc, ridx, rsc, stats = both(a, b, w, bias, 20)
    assert torch.equal(idx, ridx) and torch.equal(sc.nan_to_num(-1.0), rsc.nan_to_num(-1.0))


def test_dispatch_and_argument_checks(wb):
    from seam_match_rcnn_amd import _native, ops
    w, bias = wb
    lib = _native.lib()
    assert lib.seam_pair_topk_mfma_min_gallery() == 8192 and lib.seam_pair_topk_mfma_max_k() == 64
    a = queries(451, 8)
    small = torch.from_numpy(synth.gallery(551, 1000)).to(DEV)
    with pytest.raises(ValueError):
        ops.pair_topk(a, small, w, bias, 20, mfma=True)                # below the sample size: the VALU paths own small banks
    idx, _ = ops.pair_topk(a, small, w, bias, 20)                      # default dispatch falls back to them
    ridx, _ = ops.rank_topk(ops.pair_logits(a, small, w, bias), 20)
    assert torch.equal(idx, ridx)
