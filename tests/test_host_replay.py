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
