"""ORACLE (test infrastructure — never imported by the product path).  **Parity unpinned.**

CPU restatement (numpy + scipy.ndimage) of the DAVIS J / F metrics the reference calls from
``utils/misc.py:118-162`` (``sequence_metric``):

    from davisinteractive.metrics import batched_jaccard, batched_f_measure      utils/misc.py:8

``davisinteractive==1.0.4`` (requirements.txt:14) is a pip dependency that is NOT vendored under
/root/reference and is not installed in this image, and the reference holds no test or golden vector
for it.  The functions below restate the package's published algorithm (the DAVIS benchmark's
region-similarity and contour-accuracy measures, Perazzi et al. CVPR 2016, as packaged in
davisinteractive/metrics/jaccard.py):

  * ``batched_jaccard``    per frame and object id: |gt & pred| / |gt | pred|, 1.0 when the union is empty
  * ``seg2bmap``           boundary map: (seg ^ east) | (seg ^ south) | (seg ^ south-east); last row = seg ^ east,
                           last column = seg ^ south, bottom-right corner = 0
  * ``f_measure``          boundaries dilated by skimage ``disk(bound_pix)``, bound_pix = bound_th if >= 1 else
                           ceil(bound_th * ||(H, W)||_2); precision / recall special cases for empty boundaries;
                           F = 2PR / (P + R), 0 when P + R == 0
  * ``batched_f_measure``  loops objects x frames, optional mean over objects
  * ``sequence_metric``    the reference's own wrapper, utils/misc.py:118-162 (this one IS pinned by the reference
                           source: 'J' | 'F' | 'J_AND_F' = .5 J + .5 F, convert_to_single_obj mutates its inputs)

What pins it here: hand-derived known-answer cases in tests/test_oracle_jf.py (identical masks, empty masks,
disjoint masks, a shifted square whose matches can be counted by hand, a 4x4 boundary map).  Until the real
package's outputs can be recorded, DESIGN.md lists this row as "parity unpinned".
"""
import numpy as np
from scipy.ndimage import binary_dilation


def disk(radius):
    """skimage.morphology.disk: (2r+1)^2 footprint, x^2 + y^2 <= r^2."""
    r = int(radius)
    L = np.arange(-r, r + 1)
    X, Y = np.meshgrid(L, L)
    return (X ** 2 + Y ** 2) <= r ** 2


def _object_ids(y_true, nb_objects):
    if nb_objects is None:
        ids = np.unique(y_true[(y_true < 255) & (y_true > 0)])
        nb_objects = len(ids)
    else:
        ids = np.asarray([i + 1 for i in range(nb_objects)], dtype=np.int64)
    if nb_objects == 0:
        raise ValueError("Number of objects in y_true should be higher than 0.")
    return ids


def _check(y_true, y_pred):
    y_true = np.asarray(y_true, dtype=np.int64)
    y_pred = np.asarray(y_pred, dtype=np.int64)
    if y_true.ndim != 3:
        raise ValueError(f"y_true array must have 3 dimensions. Found {y_true.ndim} dimensions")
    if y_pred.ndim != 3:
        raise ValueError(f"y_pred array must have 3 dimensions. Found {y_pred.ndim} dimensions")
    if y_true.shape != y_pred.shape:
        raise ValueError(f"y_true and y_pred must have the same shape. {y_true.shape} != {y_pred.shape}")
    return y_true, y_pred


def batched_jaccard(y_true, y_pred, average_over_objects=True, nb_objects=None):
    y_true, y_pred = _check(y_true, y_pred)
    ids = _object_ids(y_true, nb_objects)
    nb_frames = len(y_true)
    jaccard = np.empty((nb_frames, len(ids)), dtype=np.float64)
    for i, obj_id in enumerate(ids):
        mask_true, mask_pred = y_true == obj_id, y_pred == obj_id
        union = (mask_true | mask_pred).sum(axis=(1, 2))
        intersection = (mask_true & mask_pred).sum(axis=(1, 2))
        for j in range(nb_frames):
            jaccard[j, i] = 1.0 if np.isclose(union[j], 0) else intersection[j] / union[j]
    if average_over_objects:
        jaccard = jaccard.mean(axis=1)
    return jaccard


def seg2bmap(seg):
    seg = np.asarray(seg).astype(bool)
    e = np.zeros_like(seg)
    s = np.zeros_like(seg)
    se = np.zeros_like(seg)
    e[:, :-1] = seg[:, 1:]
    s[:-1, :] = seg[1:, :]
    se[:-1, :-1] = seg[1:, 1:]
    b = seg ^ e | seg ^ s | seg ^ se
    b[-1, :] = seg[-1, :] ^ e[-1, :]
    b[:, -1] = seg[:, -1] ^ s[:, -1]
    b[-1, -1] = 0
    return b


def bound_pixels(shape, bound_th=0.008):
    return bound_th if bound_th >= 1 else np.ceil(bound_th * np.linalg.norm(shape))


def f_measure(true_mask, pred_mask, bound_th=0.008):
    true_mask = np.asarray(true_mask, dtype=bool)
    pred_mask = np.asarray(pred_mask, dtype=bool)
    assert true_mask.shape == pred_mask.shape
    bound_pix = bound_pixels(true_mask.shape, bound_th)
    fg_boundary = seg2bmap(pred_mask)
    gt_boundary = seg2bmap(true_mask)
    fg_dil = binary_dilation(fg_boundary, disk(bound_pix))
    gt_dil = binary_dilation(gt_boundary, disk(bound_pix))
    gt_match = gt_boundary * fg_dil
    fg_match = fg_boundary * gt_dil
    n_fg = np.sum(fg_boundary)
    n_gt = np.sum(gt_boundary)
    return pr_to_f(n_fg, n_gt, np.sum(fg_match), np.sum(gt_match))


def pr_to_f(n_fg, n_gt, n_fg_match, n_gt_match):
    if n_fg == 0 and n_gt > 0:
        precision, recall = 1, 0
    elif n_fg > 0 and n_gt == 0:
        precision, recall = 0, 1
    elif n_fg == 0 and n_gt == 0:
        precision, recall = 1, 1
    else:
        precision = n_fg_match / float(n_fg)
        recall = n_gt_match / float(n_gt)
    if precision + recall == 0:
        return 0
    return 2 * precision * recall / (precision + recall)


def batched_f_measure(y_true, y_pred, average_over_objects=True, nb_objects=None, bound_th=0.008):
    y_true, y_pred = _check(y_true, y_pred)
    ids = _object_ids(y_true, nb_objects)
    nb_frames = len(y_true)
    result = np.empty((nb_frames, len(ids)), dtype=np.float64)
    for i, obj_id in enumerate(ids):
        for frame_id in range(nb_frames):
            result[frame_id, i] = f_measure(y_true[frame_id] == obj_id, y_pred[frame_id] == obj_id, bound_th=bound_th)
    if average_over_objects:
        result = result.mean(axis=1)
    return result


def sequence_metric(metric_to_optimize, gt_masks, pred_masks, nb_objects, average_over_objects=True,
                    convert_to_single_obj=False):
    """utils/misc.py:118-162, line by line."""
    if convert_to_single_obj:
        gt_masks[gt_masks > 0] = 1
        pred_masks[pred_masks > 0] = 1
        nb_objects = 1
    if metric_to_optimize == 'J':
        metric = batched_jaccard(gt_masks, pred_masks, average_over_objects=average_over_objects, nb_objects=nb_objects)
    elif metric_to_optimize == 'F':
        metric = batched_f_measure(gt_masks, pred_masks, average_over_objects=average_over_objects, nb_objects=nb_objects)
    elif metric_to_optimize == 'J_AND_F':
        jaccard = batched_jaccard(gt_masks, pred_masks, average_over_objects=average_over_objects, nb_objects=nb_objects)
        contour = batched_f_measure(gt_masks, pred_masks, average_over_objects=average_over_objects, nb_objects=nb_objects)
        metric = .5 * jaccard + .5 * contour
    return metric
