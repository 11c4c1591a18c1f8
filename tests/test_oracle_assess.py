"""Pins oracle/assess_oracle.py against goldens recorded from the imported reference AssessNet
(tests/golden/make_goldens.py): bbox, theta, ROI tiles, per-stage activations, pooled vector, scores."""
import json
import os

import numpy as np
import pytest

from ivos_w_amd import synth
from oracle import assess_oracle as ao


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "assess_forward.npz"))


@pytest.fixture(scope="module")
def sd():
    return ao.to_torch_sd(synth.assessnet_state_dict(0))


def test_state_dict_keys_match_reference(golden_dir):
    ref = json.load(open(os.path.join(golden_dir, "assessnet_keys.json")))
    mine = [[k, list(np.asarray(v).shape)] for k, v in synth.assessnet_state_dict(0).items()]
    assert mine == ref and len(mine) == 326


def _slice4(a):
    hs, ws = max(1, a.shape[2] // 4), max(1, a.shape[3] // 4)
    return a[:, :6, ::hs, ::ws]


def _stat(a):
    a = np.asarray(a, np.float64)
    return np.stack([[a[b].sum(), np.abs(a[b]).sum()] for b in range(a.shape[0])])


@pytest.mark.parametrize("tag,B,edge", [("B8", 8, True), ("B1", 1, False)])
def test_forward_with_taps(gold, sd, tag, B, edge):
    tf, tp = synth.assess_inputs(B, seed=1234 + B, edge_cases=edge, structured=True)
    taps = {}
    score = ao.assess_forward(sd, tf, tp, taps)
    np.testing.assert_array_equal(taps["yxhw"], gold[f"{tag}_yxhw"])          # bbox is exact arithmetic
    th = gold[f"{tag}_theta"]
    np.testing.assert_allclose(taps["theta"], np.stack([th[:, 0, 0], th[:, 0, 2], th[:, 1, 1], th[:, 1, 2]], 1),
                               rtol=1e-6, atol=1e-7)
    for nm, key in (("f_roi", "froi"), ("p_roi", "proi")):
        a = taps[nm] if nm == "f_roi" else taps[nm][:, None]
        np.testing.assert_allclose(_slice4(a), gold[f"slice_{tag}_{key}"], rtol=1e-4, atol=2e-4, err_msg=key)
        np.testing.assert_allclose(_stat(a), gold[f"stat_{tag}_{key}"], rtol=1e-5, atol=1e-2, err_msg=key)
    for nm in ("stem", "pool", "res2", "res3", "res4", "res5"):
        a = taps[nm].numpy()
        np.testing.assert_allclose(_slice4(a), gold[f"slice_{tag}_{nm}"], rtol=1e-3, atol=2e-4, err_msg=nm)
        np.testing.assert_allclose(_stat(a), gold[f"stat_{tag}_{nm}"], rtol=1e-5, err_msg=nm)
    np.testing.assert_allclose(score, gold[f"{tag}_score"].reshape(-1), rtol=1e-4)


def test_b3_scores(gold, sd):
    tf, tp = synth.assess_inputs(3, seed=1237, structured=True)
    np.testing.assert_allclose(ao.assess_forward(sd, tf, tp), gold["B3_score"].reshape(-1), rtol=1e-4)


def test_bbox_known_answers():
    m = np.zeros((1, 480, 854), np.float32)
    m[0, 60:300, 100:420] = 1
    np.testing.assert_array_equal(ao.mask_bbox_yxhw(m, 1.5), [[179.5, 259.5, 360.0, 480.0]])   # SURVEY App. B probe
    e = np.zeros((1, 480, 854), np.float32)
    np.testing.assert_array_equal(ao.mask_bbox_yxhw(e, 1.5), [[240.0, 427.0, 491.0, 865.0]])   # empty: H,W not H-1,W-1
