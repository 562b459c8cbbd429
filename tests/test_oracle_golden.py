"""CPU: pin the oracle (oracle/heads.py) against vectors captured from the imported
reference heads (tests/golden/make_golden.py; rows a9-a14 of SURVEY.md section 8)."""
import numpy as np
import torch

import seam_match_rcnn_amd.synth as synth
from oracle import heads as O
from conftest import to_torch

SEED_MP, SEED_TA = 11, 12
RTOL, ATOL = 1e-4, 1e-5      # oracle vs reference: same ATen kernels, different op order


def close(a, b, rtol=RTOL, atol=ATOL):
    a = a.numpy() if isinstance(a, torch.Tensor) else a
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol * max(1.0, float(np.abs(b).max())))


def test_nlb_closed_form(golden):
    p = to_torch(synth.temporal_aggregator_state(SEED_TA))
    for t in (2, 3, 10):
        x = torch.from_numpy(synth.normal(synth.stream_id(21, f"nlb_x{t}"), (t, 256)))
        close(O.nlb_closed_form(x, p), golden[f"nlb_T{t}_z"])


def test_match_predictor(golden):
    p = to_torch(synth.match_predictor_state(SEED_MP))
    x = torch.from_numpy(synth.roi_features(31, 6))
    x3, x5 = O.match_predictor_forward(x, golden["mp_types"], p)
    close(x3, golden["mp_x3"])
    close(x5, golden["mp_x5"])


def test_temporal_aggregation_mode_a(golden):
    p = to_torch(synth.temporal_aggregator_state(SEED_TA))
    x = torch.from_numpy(synth.roi_features(32, 17))
    out = O.temporal_aggregation_forward(x, golden["ta_types"], golden["ta_ids"], p, getatt=True)
    for nm, v in zip(("x3_1b", "x3_2", "x5", "x3_1_seq"), out[:4]):
        close(v, golden["taA_" + nm])
    assert np.array_equal(out[4].numpy(), golden["taA_x3_1_mask"])
    assert np.array_equal(out[5].numpy(), golden["taA_x3_1_ids"])
    assert len(out[6]) == 3
    for i, a in enumerate(out[6]):
        close(a, golden[f"taA_att{i}"])


def _mode_b_inputs(golden):
    seq = torch.from_numpy(synth.normal(synth.stream_id(33, "seq"), (11, 4, 256)))
    lens = golden["taB_lens"].tolist()
    mask = torch.zeros((4, 11), dtype=torch.bool)
    seq[0] = 0
    for i, n in enumerate(lens):
        mask[i, n + 1:] = True
        seq[n + 1:, i] = 0
    return seq, mask, torch.from_numpy(synth.gallery(34, 16))


def test_temporal_aggregation_mode_b(golden):
    p = to_torch(synth.temporal_aggregator_state(SEED_TA))
    seq, mask, gal = _mode_b_inputs(golden)
    out = O.temporal_aggregation_forward(None, None, None, p, seq, mask, gal, getatt=True)
    close(out[0], golden["taB_x3_1b"])
    close(out[2], golden["taB_x5"])
    for i, a in enumerate(out[6]):
        close(a, golden[f"taB_att{i}"])
    assert out[5].shape == (1, 2)


def test_c2_mode_b_topk(golden):
    p = to_torch(synth.temporal_aggregator_state(SEED_TA))
    s, t, g = 32, 10, 1000
    seq = torch.zeros((1 + t, s, 256))
    seq[1:] = torch.from_numpy(synth.normal(synth.stream_id(35, "seq_c2"), (t, s, 256)))
    mask = torch.zeros((s, 1 + t), dtype=torch.bool)
    gal = torch.from_numpy(synth.gallery(36, g))
    out = O.temporal_aggregation_forward(None, None, None, p, seq, mask, gal)
    close(out[0], golden["c2_x3_1b"])
    x5 = out[2]
    close(x5.reshape(-1)[::16], golden["c2_x5_sample"])
    idx, sc = O.rank_topk(x5, 20)
    # ties are an unordered set: compare the score profile, then index sets
    close(sc, golden["c2_top20_score"], rtol=1e-3)
    agree = (idx.numpy() == golden["c2_top20"]).mean()
    assert agree > 0.98, agree
    for q in range(s):
        assert len(set(idx[q].tolist()) ^ set(golden["c2_top20"][q].tolist())) <= 2
