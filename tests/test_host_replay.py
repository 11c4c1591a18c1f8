"""Host logic (no GPU): ReplayMemory / CSV persistence / dataset collation vs fixtures recorded from the reference
(tests/golden/replay_fixtures.json, made by make_goldens.py `replay`)."""
import io
import json
import os

import numpy as np
import pandas as pd
import pytest
import torch

from ivos_w_amd.datasets.agent_dataset import DAVIS2017AgentTrain
from ivos_w_amd.models.momory_pool import ReplayMemory, Transition, parse_rows


@pytest.fixture(scope="module")
def fx(golden_dir):
    return json.load(open(os.path.join(golden_dir, "replay_fixtures.json")))


def rows_of(fx):
    return pd.read_csv(io.StringIO(fx["pretrain_csv"]), index_col=0).to_dict("records")


def test_transition_field_order():
    assert Transition._fields == ("state", "action", "next_state", "reward_step", "reward_done", "done", "state_iou",
                                  "next_state_iou", "annotated_frames", "next_annotated_frames")


def test_push_and_push_to_csv_match_reference(fx, tmp_path):
    mem = ReplayMemory(5)
    assert mem.position == -1 and mem.basename_csv == "memory_pool.csv" and len(mem.COLUMNS) == 12
    for r in rows_of(fx)[:7]:
        st = dict(sequence=r["sequence"], scribble_iter=r["scribble_iter"], n_interaction=r["n_interaction"])
        nst = dict(sequence=r["sequence"], scribble_iter=r["scribble_iter"], n_interaction=r["n_interaction_next"])
        mem.push(st, r["action"], nst, r["reward_step"], r["reward_done"], r["done"], r["state_iou"],
                 r["next_state_iou"], r["annotated_frames"], r["next_annotated_frames"])
        mem.push_to_csv(str(tmp_path))
    assert mem.position == fx["push_position"] and len(mem) == fx["push_len"]
    assert [int(t.action) for t in mem.memory] == fx["push_actions_in_ring"]
    assert open(tmp_path / "memory_pool.csv").read() == fx["push_csv"]          # byte-identical file


def test_load_from_csv_filter_and_capacity_shrink(fx, tmp_path):
    src = tmp_path / "pretrain.csv"
    src.write_text(fx["pretrain_csv"])
    mem = ReplayMemory(8)
    mem.load_from_csv(str(src), str(tmp_path / "out"), sample_th=0.05)
    assert [str(s) for s in mem.seq_list] == fx["load_seq_list"]
    assert (mem.capacity, len(mem), mem.position) == (fx["load_capacity"], fx["load_len"], fx["load_position"])
    assert [int(t.action) for t in mem.memory] == fx["load_actions"]
    assert open(tmp_path / "out" / "memory_pool.csv").read() == fx["load_csv"]


def test_dataset_collation_matches_reference(fx, tmp_path):
    (tmp_path / "memory_pool.csv").write_text(fx["load_csv"])
    root = tmp_path / "DAVIS" / "ImageSets" / "2017"
    root.mkdir(parents=True)
    (root / "train.txt").write_text("bear\ncamel\ndrift\nelephant\nflamingo\n")
    np.random.seed(0)
    ds = DAVIS2017AgentTrain(split="train", db_root_dir=str(tmp_path / "DAVIS"), save_result_dir=str(tmp_path),
                             memory_size=100, seq_list=fx["load_seq_list"])
    assert len(ds) == fx["ds_len"]
    batch = next(iter(torch.utils.data.DataLoader(ds, batch_size=3, shuffle=False)))
    assert set(batch) == set(fx["batch"])
    for k, want in fx["batch"].items():
        assert str(batch[k].dtype) == want["dtype"] and list(batch[k].shape) == want["shape"], k
        np.testing.assert_array_equal(batch[k].double().numpy(), np.array(want["values"]), err_msg=k)


def test_parse_rows_soa(fx):
    frame = pd.read_csv(io.StringIO(fx["pretrain_csv"]), index_col=0)
    soa = parse_rows(frame, T=4)
    assert soa["old_state_iou"].shape == (9, 4) and soa["action"].dtype == np.int64
    np.testing.assert_allclose(soa["next_annotated_frames"].sum(1) - soa["annotated_frames"].sum(1), 1.0)


def test_random_sample_surface():
    mem = ReplayMemory(4)
    assert mem.random_sample(1) is None
    for i in range(3):
        mem.push({}, i, {}, 1, 0.0, False, "0.1", "0.2", "0.0", "1.0")
    s = mem.random_sample(2)
    assert isinstance(s, Transition) and len(s.action) == 2


# ---------------------------------------------------------------- §8(f) rank 2: persistence without pandas on the hot path
def _reference_push_to_csv(frame, row, capacity, columns):
    """The reference's push_to_csv (models/momory_pool.py:126-153) restated with pandas, as the checker."""
    frame = pd.concat([frame, pd.DataFrame(data={k: [v] for k, v in row.items()}, columns=columns)], ignore_index=True) \
        if len(frame) else pd.DataFrame(data={k: [v] for k, v in row.items()}, columns=columns)
    if len(frame) > capacity:
        frame = frame.drop(frame.index.min())
    return frame


def _random_rows(n, T, seed):
    rs = np.random.RandomState(seed)
    seqs = ["bear", "camel", "drift-chicane", "india", "kite_surf"]
    j = lambda a: "/".join(str(round(float(x), 4)) for x in a)
    rows = []
    for i in range(n):
        k = int(rs.randint(1, 5))
        rows.append(dict(sequence=seqs[int(rs.randint(len(seqs)))], scribble_iter=int(rs.randint(1, 4)), n_interaction=k,
                         n_interaction_next=k + 1, action=int(rs.randint(T)), reward_step=int(rs.choice([1, -1])),
                         reward_done=float(rs.randn()) if rs.rand() < 0.9 else float(int(rs.randint(-2, 3))),
                         done=bool(rs.rand() < 0.3), state_iou=j(rs.rand(T)), next_state_iou=j(rs.rand(T)),
                         annotated_frames=j(rs.randint(0, 2, T).astype(float)), next_annotated_frames=j(rs.randint(0, 3, T).astype(float))))
    return rows


def _push(mem, r):
    st = dict(sequence=r["sequence"], scribble_iter=r["scribble_iter"], n_interaction=r["n_interaction"])
    nst = dict(st, n_interaction=r["n_interaction_next"])
    mem.push(st, r["action"], nst, r["reward_step"], r["reward_done"], r["done"], r["state_iou"], r["next_state_iou"],
             r["annotated_frames"], r["next_annotated_frames"])


@pytest.mark.parametrize("capacity,n", [(6, 17), (50, 20)])
def test_text_row_writer_is_byte_identical_to_the_pandas_path(tmp_path, capacity, n):
    mem = ReplayMemory(capacity)
    frame = pd.DataFrame(columns=mem.COLUMNS)
    for r in _random_rows(n, T=5, seed=capacity):
        _push(mem, r)
        mem.push_to_csv(str(tmp_path))
        frame = _reference_push_to_csv(frame, r, capacity, mem.COLUMNS)
        assert open(tmp_path / "memory_pool.csv").read() == frame.to_csv()
    assert mem.memory_pd.to_csv() == frame.to_csv()                                  # the DataFrame view of the same rows


def test_deferred_sync_and_sidecar(tmp_path):
    rows = _random_rows(23, T=4, seed=3)
    mem = ReplayMemory(10)
    mem.csv_sync_every = 8
    frame = pd.DataFrame(columns=mem.COLUMNS)
    for i, r in enumerate(rows):
        _push(mem, r)
        mem.push_to_csv(str(tmp_path))
        frame = _reference_push_to_csv(frame, r, 10, mem.COLUMNS)
        if (i + 1) % 8 == 0:
            assert open(tmp_path / "memory_pool.csv").read() == frame.to_csv()       # written every 8th push ...
    assert open(tmp_path / "memory_pool.csv").read() != frame.to_csv()               # ... and stale in between
    mem.sync_csv(str(tmp_path))
    assert open(tmp_path / "memory_pool.csv").read() == frame.to_csv()
    # the binary sidecar holds the same rows; the dataset built from it equals the one parsed from the CSV text
    np.random.seed(11)
    a = DAVIS2017AgentTrain(split="train", db_root_dir=str(tmp_path / "nope"), save_result_dir=str(tmp_path), memory_size=100,
                            seq_list=["bear", "india", "kite_surf"])
    assert a.frame is None                                                           # came from memory_pool.npz
    os.remove(tmp_path / "memory_pool.npz")
    np.random.seed(11)
    b = DAVIS2017AgentTrain(split="train", db_root_dir=str(tmp_path / "nope"), save_result_dir=str(tmp_path), memory_size=100,
                            seq_list=["bear", "india", "kite_surf"])
    assert b.frame is not None and len(a) == len(b) > 0
    for x, y in zip(a.samples_list, b.samples_list):
        assert set(x) == set(y)
        for k in x:
            assert np.asarray(x[k]).dtype == np.asarray(y[k]).dtype and np.array_equal(x[k], y[k]), k


def test_cells_that_need_quoting_fall_back_to_pandas(tmp_path):
    mem = ReplayMemory(3)
    frame = pd.DataFrame(columns=mem.COLUMNS)
    for r in _random_rows(5, T=3, seed=9):
        r = dict(r, sequence=r["sequence"] + ",take 2")
        _push(mem, r)
        mem.push_to_csv(str(tmp_path))
        frame = _reference_push_to_csv(frame, r, 3, mem.COLUMNS)
        assert open(tmp_path / "memory_pool.csv").read() == frame.to_csv()


def _push_sequence_against_pandas(tmp_path, values, cap=4, preload=None):
    """The reference's push_to_csv (models/momory_pool.py:126-153) is a pandas concat + to_csv: replay it with pandas itself and
    compare the file after every push."""
    import pandas as pd
    from ivos_w_amd.models.momory_pool import ReplayMemory
    d = str(tmp_path)
    m = ReplayMemory(cap)
    ref = None
    cols = m.COLUMNS
    if preload is not None:
        p = os.path.join(d, "pre.csv")
        open(p, "w").write(preload)
        m.load_from_csv(p, d, 0)
        ref = pd.read_csv(os.path.join(d, "memory_pool.csv"), index_col=0)
        cap = m.capacity
    for i, rd in enumerate(values):
        st = dict(sequence="s", scribble_iter=1, n_interaction=i + 1)
        ns = dict(sequence="s", scribble_iter=1, n_interaction=i + 2)
        m.push(st, 3, ns, np.array(1), rd, False, "0.1/0.2", "0.2/0.3", "0.0/1.0", "1.0/1.0")
        m.push_to_csv(d)
        row = pd.DataFrame(data={k: [v] for k, v in m._row_at(m.position).items()}, columns=cols)
        ref = row if ref is None else pd.concat([ref, row], ignore_index=True)
        if len(ref) > cap:
            ref = ref.drop(ref.index.min())
        assert open(os.path.join(d, "memory_pool.csv")).read() == ref.to_csv(), i


@pytest.mark.parametrize("case", range(6))
def test_csv_image_follows_pandas_dtype_promotion(tmp_path, case):
    """when pandas promotes a column (an int64 column meets a typed float: every past row is rewritten "0" -> "0.0";
    a 0-d array cell turns the column into objects that keep their own text) the text-row image must do the same."""
    pre = (",sequence,scribble_iter,n_interaction,n_interaction_next,action,reward_step,reward_done,done,state_iou,next_state_iou,"
           "annotated_frames,next_annotated_frames\n" + "".join(f"{i},s,1,{i+1},{i+2},3,1,0,False,0.1/0.2,0.6/0.7,0.0/1.0,1.0/1.0\n" for i in range(3)))
    cases = [([np.array(0), np.array(0), 0.25, np.array(0), np.array(0), np.array(0), np.array(0)], None),
             ([0.5, np.array(0), np.array(1)], None),
             ([0, 0, 0.25, 1, 0, 0, 2], None),
             ([np.float64(0.5), 1, np.float64(0.25), np.array(0), 3, 0.5], None),
             ([1, 2, np.float64(0.5), 3, 4, 5, 6, 7], None),
             ([np.float64(0.5), 1, np.array(0)], pre)]          # int-valued column loaded from a file, then a float pushed
    values, preload = cases[case]
    _push_sequence_against_pandas(tmp_path, values, cap=10 if preload else 4, preload=preload)
