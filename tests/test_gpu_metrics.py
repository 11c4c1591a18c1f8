"""GPU parity: DAVIS J / F metrics (csrc/metrics.hip through the C ABI and ivos_w_amd.metrics) vs the numpy/scipy
oracle.  Integer counts -> float64 ratios formed by the same expressions: the bar is bit-exact equality."""
import numpy as np
import pytest
import torch

from ivos_w_amd import metrics, synth
from ivos_w_amd.utils import misc
from oracle import jf_oracle as jo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need the MI355X"
    return torch.device("cuda:0")


def test_hand_cases_on_device(dev):
    gt = np.zeros((1, 8, 8), int)
    pr = np.zeros((1, 8, 8), int)
    gt[0, 3, 3] = 1
    pr[0, 3, 5] = 1
    np.testing.assert_array_equal(metrics.batched_f_measure(gt, pr, nb_objects=1, bound_th=1), [0.5])
    np.testing.assert_array_equal(metrics.batched_jaccard(gt, pr, nb_objects=1), [0.0])
    ids, c = metrics.jf_counts(gt, pr, nb_objects=1, bound_th=1)
    np.testing.assert_array_equal(c[0, 0], [0, 2, 4, 4, 2, 2])
    H, W = 40, 60
    sq = np.zeros((H, W), int)
    sq[10:20, 10:30] = 1
    far = np.zeros((H, W), int)
    far[30:38, 45:58] = 1
    empty = np.zeros((H, W), int)
    g, p = np.stack([sq, empty, sq, sq, empty]), np.stack([sq, empty, empty, far, sq])
    np.testing.assert_array_equal(metrics.batched_jaccard(g, p, nb_objects=1), [1.0, 1.0, 0.0, 0.0, 0.0])
    np.testing.assert_array_equal(metrics.batched_f_measure(g, p, nb_objects=1), [1.0, 1.0, 0.0, 0.0, 0.0])
    with pytest.raises(ValueError):
        metrics.batched_jaccard(np.zeros((2, 4, 4), int), np.zeros((2, 4, 4), int))
    with pytest.raises(ValueError):
        metrics.batched_jaccard(np.zeros((4, 4), int), np.zeros((4, 4), int), nb_objects=1)


@pytest.mark.parametrize("N,H,W,O,bth", [
    (4, 480, 854, 3, 0.008),      # the reference's resolution: radius 8
    (3, 37, 45, 2, 0.008),        # radius 1, ragged last word
    (2, 64, 64, 1, 3),            # width a multiple of 32, explicit pixel tolerance
    (2, 50, 33, 2, 0.1),          # one pixel in the second word; radius 6
    (1, 9, 100, 1, 0.008),
    (2, 200, 300, 4, 20),         # wide disk: smears cross word borders by 20 px
    (2, 33, 32, 1, 32),           # the largest supported radius
    (1, 21, 1100, 2, 4),          # wider than one 1024-pixel wave segment: the east neighbour crosses segments
    (1, 9, 1024, 1, 2),           # exactly one segment
    (1, 9, 1025, 1, 2),           # one pixel in the second segment
    (2, 17, 16, 1, 1),            # a single lane per row
])
def test_counts_and_metrics_match_oracle(dev, N, H, W, O, bth):
    gt, pr = synth.label_maps(N, H, W, O, seed=N * 1000 + W, void=(O > 1))
    rs = np.random.RandomState(5)
    noise = rs.rand(N, H, W) < 0.002          # isolated wrong pixels: many tiny boundaries
    pr = np.where(noise, (pr + 1) % (O + 1), pr).astype(np.uint8)
    for avg in (True, False):
        np.testing.assert_array_equal(metrics.batched_jaccard(gt, pr, avg, O), jo.batched_jaccard(gt, pr, avg, O))
        np.testing.assert_array_equal(metrics.batched_f_measure(gt, pr, avg, O, bth), jo.batched_f_measure(gt, pr, avg, O, bth))
    # the raw integer counts against the oracle's own intermediate quantities
    ids, c = metrics.jf_counts(gt, pr, O, bth)
    r = jo.bound_pixels((H, W), bth)
    for n in range(N):
        for o, oid in enumerate(ids):
            g, p = gt[n] == oid, pr[n] == oid
            bg, bp = jo.seg2bmap(g), jo.seg2bmap(p)
            dg, dp = jo.binary_dilation(bg, jo.disk(r)), jo.binary_dilation(bp, jo.disk(r))
            want = [(g & p).sum(), (g | p).sum(), bp.sum(), bg.sum(), (bp & dg).sum(), (bg & dp).sum()]
            np.testing.assert_array_equal(c[n, o], want, err_msg=f"frame {n} object {oid}")


def test_object_ids_from_labels_and_device_inputs(dev):
    gt, pr = synth.label_maps(3, 120, 160, 3, seed=2, void=True)
    gt[gt == 2] = 0                              # ids {1, 3} only; 255 = void is never an object
    want_j, want_f = jo.batched_jaccard(gt, pr), jo.batched_f_measure(gt, pr)
    np.testing.assert_array_equal(metrics.batched_jaccard(gt, pr), want_j)
    np.testing.assert_array_equal(metrics.batched_f_measure(gt, pr), want_f)
    tg, tp = torch.from_numpy(gt.astype(np.int64)).to(dev), torch.from_numpy(pr).to(dev)     # int64 / uint8 device tensors
    j, f = metrics.batched_j_and_f(tg, tp)
    np.testing.assert_array_equal(j, want_j)
    np.testing.assert_array_equal(f, want_f)


def test_sequence_metric_drop_in(dev):
    gt, pr = synth.label_maps(5, 480, 854, 2, seed=9)
    for m in ("J", "F", "J_AND_F"):
        for avg in (True, False):
            got = misc.sequence_metric(m, gt, pr, 2, average_over_objects=avg)
            want = jo.sequence_metric(m, gt, pr, 2, average_over_objects=avg)
            assert got.dtype == np.float64 and got.shape == want.shape
            np.testing.assert_array_equal(got, want)
    g1, p1, g2, p2 = gt.copy(), pr.copy(), gt.copy(), pr.copy()
    got = misc.sequence_metric("J_AND_F", g1, p1, 2, convert_to_single_obj=True)
    want = jo.sequence_metric("J_AND_F", g2, p2, 2, convert_to_single_obj=True)
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(g1, g2)       # the in-place rewrite of the caller's arrays is part of the behaviour
    assert g1.max() == 1


def test_full_size_properties(dev):
    """100 frames x 3 objects at 480p (a DAVIS sequence): symmetric roles, identity and label permutation."""
    gt, pr = synth.label_maps(100, 480, 854, 3, seed=4)
    ids, c = metrics.jf_counts(gt, pr, 3)
    ids2, c2 = metrics.jf_counts(pr, gt, 3)
    # swapping gt and pred swaps (n_fg, n_gt) and (fg_match, gt_match), keeps intersection / union
    np.testing.assert_array_equal(c2[..., [0, 1, 3, 2, 5, 4]], c)
    j, f = metrics.batched_j_and_f(gt, gt, nb_objects=3)
    assert (j == 1).all() and (f == 1).all()
    perm = np.array([0, 2, 3, 1], np.uint8)     # relabel objects 1->2, 2->3, 3->1 in both maps
    _, cp = metrics.jf_counts(perm[gt], perm[pr], 3)
    np.testing.assert_array_equal(cp[:, [1, 2, 0]], c)
    sub = slice(10, 17)
    _, cs = metrics.jf_counts(gt[sub], pr[sub], 3)
    np.testing.assert_array_equal(cs, c[sub])   # frames are independent units
    np.testing.assert_array_equal(c[0, :, 0], c[0, :, 1])     # frame 0 of pred is a copy of gt
