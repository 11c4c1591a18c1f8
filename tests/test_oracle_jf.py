"""CPU: known-answer cases for the J/F oracle (oracle/jf_oracle.py).  The davisinteractive package the reference calls
is not vendored and the reference holds no vectors for it (parity unpinned), so these cases are derived by hand from
the published definition and are what pins the restatement."""
import numpy as np
import pytest

from oracle import jf_oracle as jo


def test_seg2bmap_known_answer():
    seg = np.array([[0, 0, 0, 0], [0, 1, 1, 0], [0, 1, 1, 0], [0, 0, 0, 0]])
    # (seg^e)|(seg^s)|(seg^se) puts the boundary on the pixels whose east / south / south-east neighbour differs:
    # the ring sits one pixel up-left of the square; last column = seg ^ south, last row = seg ^ east, corner 0
    want = np.array([[1, 1, 1, 0], [1, 0, 1, 0], [1, 1, 1, 0], [0, 0, 0, 0]], bool)
    np.testing.assert_array_equal(jo.seg2bmap(seg), want)
    # a mask touching the bottom-right corner: the special-cased last row / column
    seg = np.zeros((3, 3), int)
    seg[2, 2] = 1
    want = np.array([[0, 0, 0], [0, 1, 1], [0, 1, 0]], bool)     # [1,2]: seg^s = 1 ; [2,1]: seg^e = 1 ; [2,2] forced 0
    np.testing.assert_array_equal(jo.seg2bmap(seg), want)


def test_disk_matches_skimage_definition():
    d = jo.disk(8)
    assert d.shape == (17, 17) and d[8, 0] and d[0, 8] and not d[0, 0]
    # half widths per |dy| for r = 8: floor(sqrt(64 - dy^2))
    assert [int(d[8 + k].sum() - 1) // 2 for k in range(9)] == [8, 7, 7, 7, 6, 6, 5, 3, 0]
    assert jo.bound_pixels((480, 854)) == 8 and jo.bound_pixels((480, 854), 3) == 3


def test_single_pixel_pair_by_hand():
    gt = np.zeros((1, 8, 8), int)
    pr = np.zeros((1, 8, 8), int)
    gt[0, 3, 3] = 1
    pr[0, 3, 5] = 1
    # boundaries: the 2x2 blocks {2,3}x{2,3} and {2,3}x{4,5}; disk(1) is the plus shape: (2,4),(3,4) touch the gt
    # boundary, (2,3),(3,3) touch the pred boundary -> precision = recall = 2/4, F = 0.5; J = 0/2
    np.testing.assert_array_equal(jo.batched_f_measure(gt, pr, nb_objects=1, bound_th=1), [0.5])
    np.testing.assert_array_equal(jo.batched_jaccard(gt, pr, nb_objects=1), [0.0])


def test_special_cases():
    H, W = 40, 60
    sq = np.zeros((H, W), int)
    sq[10:20, 10:30] = 1
    far = np.zeros((H, W), int)
    far[30:38, 45:58] = 1
    empty = np.zeros((H, W), int)
    gt = np.stack([sq, empty, sq, sq, empty])
    pr = np.stack([sq, empty, empty, far, sq])
    j = jo.batched_jaccard(gt, pr, nb_objects=1)
    f = jo.batched_f_measure(gt, pr, nb_objects=1)
    np.testing.assert_array_equal(j, [1.0, 1.0, 0.0, 0.0, 0.0])          # identical; both empty -> 1; disjoint -> 0
    np.testing.assert_array_equal(f, [1.0, 1.0, 0.0, 0.0, 0.0])          # P=1,R=0 | P=R=0 | P=0,R=1 -> F = 0
    with pytest.raises(ValueError):
        jo.batched_jaccard(np.zeros((2, 4, 4), int), np.zeros((2, 4, 4), int))     # no object ids in y_true
    with pytest.raises(ValueError):
        jo.batched_jaccard(np.zeros((4, 4), int), np.zeros((4, 4), int), nb_objects=1)


def test_objects_and_sequence_metric():
    gt = np.zeros((2, 30, 30), np.int64)
    pr = np.zeros((2, 30, 30), np.int64)
    gt[:, 2:12, 2:12] = 1
    gt[:, 15:25, 15:28] = 2
    pr[:, 2:12, 2:12] = 1
    pr[:, 15:25, 15:21] = 2           # object 2: half of the box -> J = 60/130
    j = jo.batched_jaccard(gt, pr, average_over_objects=False, nb_objects=2)
    np.testing.assert_array_equal(j, [[1.0, 60 / 130]] * 2)
    np.testing.assert_array_equal(jo.batched_jaccard(gt, pr), j.mean(axis=1))     # ids from np.unique, mean over objects
    jf = jo.sequence_metric('J_AND_F', gt, pr, 2)
    np.testing.assert_array_equal(jf, .5 * jo.sequence_metric('J', gt, pr, 2) + .5 * jo.sequence_metric('F', gt, pr, 2))
    g2, p2 = gt.copy(), pr.copy()
    one = jo.sequence_metric('J', g2, p2, 2, convert_to_single_obj=True)
    assert g2.max() == 1 and p2.max() == 1                                           # the reference rewrites its inputs
    np.testing.assert_array_equal(one, [(100 + 60) / (100 + 130)] * 2)


def test_f_measure_against_brute_force_definition():
    """An independent restatement of the matching step: a boundary pixel of A is matched when some boundary pixel of B lies
    within Euclidean distance bound_pix (disk structuring element), checked pair by pair in pure Python on small random
    masks — pins the dilation / disk convention of the oracle (scipy.ndimage) against the definition itself."""
    rs = np.random.RandomState(0)
    for trial in range(12):
        H, W = rs.randint(6, 15), rs.randint(6, 15)
        gt = (rs.rand(H, W) < 0.35)
        pr = gt.copy()
        flip = rs.rand(H, W) < 0.15
        pr[flip] = ~pr[flip]
        r = int(rs.randint(1, 4))
        bg, bp = jo.seg2bmap(gt), jo.seg2bmap(pr)
        pg, pp = np.argwhere(bg), np.argwhere(bp)

        def matched(src, dst):
            n = 0
            for (y, x) in src:
                if any((y - v) ** 2 + (x - u) ** 2 <= r * r for (v, u) in dst):
                    n += 1
            return n
        want = jo.pr_to_f(len(pp), len(pg), matched(pp, pg), matched(pg, pp))
        got = jo.f_measure(gt, pr, bound_th=r)
        assert got == want, (trial, H, W, r)
