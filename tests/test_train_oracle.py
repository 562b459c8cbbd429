"""The oracle's training step (oracle/losses.py + oracle/heads.py under torch CPU autograd) against fixtures captured
from the reference's own modules and loss classes in train mode (tests/golden/make_train_golden.py)."""
import os

import numpy as np
import pytest
import torch

import seam_match_rcnn_amd.synth as synth
from conftest import ROOT, to_torch
from oracle import losses as OL


@pytest.fixture(scope="module")
def tg():
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "train_golden.npz")))


def check_grads(tg, prefix, grads, rtol=2e-4, atol_frac=2e-5, l2_keys=(), l2_tol=5e-3):
    """grads: name -> tensor; compares with the golden's full arrays / strided samples / float64 sums.
    l2_keys: parameters compared by relative L2 error instead of element-wise (see tests/test_gpu_train.py)."""
    n = 0
    for key, want in tg.items():
        if not key.startswith(prefix) or key.endswith("#sum"):
            continue
        name = key[len(prefix):]
        if name.startswith("bn_"):
            continue
        step = 1
        if "@" in name:
            name, step = name.split("@")[0], int(name.split("@")[1])
        got = grads[name].detach().cpu().numpy().reshape(-1)[::step]
        scale = float(np.abs(want).max()) + 1e-30
        n += 1
        if scale < 2e-5:           # analytically zero (a bias in front of a batch-statistics BatchNorm): rounding noise only
            assert float(np.abs(got).max()) < 1e-4, key
            continue
        if any(z in name for z in l2_keys):
            assert float(np.linalg.norm(got - want) / np.linalg.norm(want)) < l2_tol, key
            continue
        np.testing.assert_allclose(got, want, rtol=rtol, atol=atol_frac * scale, err_msg=key)
        s = tg[prefix + name + "#sum"]
        full = grads[name].detach().cpu().numpy().astype(np.float64)
        assert abs(np.abs(full).sum() - s[1]) <= 1e-4 * s[1] + 1e-12, key
    return n


def test_train_step_matches_reference_fixture(tg):
    mp, ta = to_torch(synth.match_predictor_state(11)), to_torch(synth.temporal_aggregator_state(12))
    types = torch.from_numpy(tg["types"])
    x = torch.from_numpy(synth.roi_features(41, len(types)))
    out = OL.train_step(x, types, tg["prod_ids"].tolist(), tg["img_ids"].tolist(), mp, ta, n_frames=3)
    np.testing.assert_allclose(out["logits"].numpy(), tg["logits"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(float(out["match_loss"]), float(tg["match_loss"]), rtol=1e-5)
    np.testing.assert_allclose(float(out["aggregation_loss"]), float(tg["aggregation_loss"]), rtol=1e-5)
    assert check_grads(tg, "mp.", out["grads_mp"]) == 14
    assert check_grads(tg, "ta.", out["grads_ta"]) >= 24
    for nm, p in (("mp", mp), ("ta", ta)):            # BatchNorm buffers were updated in place by the step
        np.testing.assert_allclose(p["linear.1.running_mean"].numpy(), tg[f"{nm}.bn_mean"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(p["linear.1.running_var"].numpy(), tg[f"{nm}.bn_var"], rtol=1e-5, atol=1e-6)


def test_df2_losses_match_reference_fixture(tg):
    from oracle import heads as OH
    mp, ta = to_torch(synth.match_predictor_state(11)), to_torch(synth.temporal_aggregator_state(12))
    types = torch.from_numpy(tg["types"])
    x = torch.from_numpy(synth.roi_features(41, len(types)))
    buf = ("running_mean", "running_var", "num_batches_tracked")
    mpg = {k: (v if k.endswith(buf) else v.clone().requires_grad_(True)) for k, v in mp.items()}
    tag = {k: (v if k.endswith(buf) else v.clone().requires_grad_(True)) for k, v in ta.items()}
    _, logits = OH.match_predictor_forward(x, types, mpg, bn_train=True)
    d1 = OL.match_loss_df2(logits, types, tg["df2_raw_gt"])
    d2 = OL.aggregation_loss_df2(types, x, tg["df2_raw_gt"], tag)
    (d1 + d2).backward()
    np.testing.assert_allclose(float(d1), float(tg["df2_match_loss"]), rtol=1e-5)
    np.testing.assert_allclose(float(d2), float(tg["df2_aggregation_loss"]), rtol=1e-5)
    assert check_grads(tg, "df2.mp.", {k: v.grad for k, v in mpg.items() if v.requires_grad}) == 6
    assert check_grads(tg, "df2.ta.", {k: v.grad for k, v in tag.items() if v.requires_grad}) == 7
