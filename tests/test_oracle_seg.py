"""CPU: the oracle restatement of get_results' epilogue (oracle/seg_oracle.py) — frame order after the two propagation
sweeps, argmax/softmax consistency, label hand-over to the next propagation step."""
import numpy as np
import torch

from oracle import seg_oracle as so


class OrderModel:
    """Logits whose winning channel encodes (frame_num % C); records the call sequence and the labels it was handed."""
    dynamic_seghead = None

    def __init__(self, C):
        self.C, self.calls, self.prev = C, [], []

    def _lg(self, f):
        x = torch.zeros(1, self.C, 5, 6)
        x[0, f % self.C] = 2.0 + 0.01 * f
        return x

    def int_seghead(self, **kw):
        self.calls.append(kw["frame_num"][0])
        return {kw["seq_names"][0]: self._lg(kw["frame_num"][0])}, kw["local_map_dics"]

    def prop_seghead(self, ref, prev_emb, cur, scr, prev_label, **kw):
        self.calls.append(kw["frame_num"][0])
        self.prev.append(int(prev_label.reshape(-1)[0]))
        return {kw["seq_names"][0]: self._lg(kw["frame_num"][0])}, kw["global_map_tmp_dic"], kw["local_map_dics"]


def test_get_results_order_and_consistency():
    n, C, nf = 6, 4, 2
    m = OrderModel(C)
    store = {}
    emb = torch.zeros(n, 3, 2, 2)
    fm, ap = so.get_results(m, emb[nf:nf + 1], None, None, {}, ({}, {}), 1, "s", C - 1, nf, True, 10, 12, store, n, emb, knns=5)
    assert m.calls == [2, 3, 4, 5, 1, 0]                       # head on next_frame, sweep ->, sweep <-
    assert m.prev == [2, 3, 0, 2, 1]                           # each step receives the label of the frame before it in its sweep
    assert tuple(fm.shape) == (n, 10, 12) and tuple(ap.shape) == (n, C, 10, 12)
    np.testing.assert_array_equal(fm[:, 0, 0].numpy(), [f % C for f in range(n)])     # frames come back in temporal order
    np.testing.assert_array_equal(ap.argmax(1).float().numpy(), fm.numpy())
    np.testing.assert_allclose(ap.sum(1).numpy(), 1.0, atol=1e-6)
    assert sorted(store) == list(range(n)) and store[4].dtype == torch.int64


def test_epilogue_is_the_reference_ops():
    x = torch.randn(2, 3, 7, 9, generator=torch.Generator().manual_seed(0))
    up, lab = so.epilogue(x, 14, 27)
    # align_corners=True: corner samples are the source corners
    np.testing.assert_array_equal(up[:, :, 0, 0].numpy(), x[:, :, 0, 0].numpy())
    np.testing.assert_array_equal(up[:, :, -1, -1].numpy(), x[:, :, -1, -1].numpy())
    np.testing.assert_array_equal(lab.numpy(), up.argmax(1).numpy())


def test_oracle_get_results_vs_reference_golden(golden_dir):
    """oracle/seg_oracle.get_results against what the REFERENCE's get_results (utils/utils_manet.py:59-163, imported in the
    build container by tests/golden/make_goldens.py seg) returned for the same stand-in model: same torch CPU kernels, so the
    probabilities are equal bit for bit, and so are the label maps, the call order, the labels handed to the next propagation
    step (they steer the stand-in's logits) and prev_label_storage."""
    import os
    from tests.golden import scenarios as sc
    fx = np.load(os.path.join(golden_dir, "seg_get_results.npz"))
    for name in sc.SEG_CASES:
        model, kw = sc.seg_case(name, torch.device("cpu"))
        store = {}
        with torch.no_grad():
            fm, ap = so.get_results(model, prev_label_storage=store, knns=sc.SEG_KNNS, **kw)
        got = sc.seg_record(name, fm, ap, store, model.calls)
        for k, v in got.items():
            np.testing.assert_array_equal(v, fx[k], err_msg=k)
        assert bool(fx[f"{name}.final_is_float32"]) and bool(fx[f"{name}.storage_is_int64"]) and bool(fx[f"{name}.storage_equals_final"])
        assert fx[f"{name}.calls"][0, 0] == 0 and (fx[f"{name}.calls"][1:, 2] == sc.SEG_KNNS).all()      # cfg.KNNS reaches every propagation call
