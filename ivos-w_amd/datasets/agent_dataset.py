"""Replay minibatch source — drop-in for the reference's ``datasets.agent_dataset``.

``DAVIS2017AgentTrain`` parses ``memory_pool.csv`` into per-sample dicts with the shapes/dtypes the reference
yields (datasets/agent_dataset.py:71-115: scalars + four ``(1, T)`` float64 arrays), so a stock DataLoader collates
them into the batch ``Agent.update_agent`` expects.  ``to_device_replay`` turns the same rows into the
device-resident SoA buffer that ``ivosw_replay_gather`` samples from (no per-step host work).
"""
import os
import time

import numpy as np
import pandas as pd
import torch

from ..models.momory_pool import DeviceReplay, parse_rows


class DAVIS2017AgentTrain(torch.utils.data.Dataset):
    def __init__(self, split=None, db_root_dir=None, save_result_dir=None, memory_size=None, transform=None,
                 seq_list=None):
        self.seq_list, self.split, self.db_root_dir = seq_list, split, db_root_dir
        self.save_result_dir, self.memory_size, self.transform = save_result_dir, memory_size, transform
        csv_path = os.path.join(save_result_dir, "memory_pool.csv")
        assert os.path.exists(csv_path), f"{csv_path} does not exist"
        self.seqs = []
        split_file = os.path.join(db_root_dir or "", "ImageSets", "2017", f"{split}.txt")
        if os.path.exists(split_file):
            with open(split_file) as f:
                self.seqs = [ln.strip() for ln in f.readlines()]
        npz_path = os.path.join(save_result_dir, "memory_pool.npz")
        soa = names = None
        if os.path.exists(npz_path) and os.path.getmtime(npz_path) >= os.path.getmtime(csv_path):
            # binary SoA sidecar written by ReplayMemory.sync_csv together with the CSV: same rows, no text parsing.
            # The shuffle consumes np.random exactly like DataFrame.sample does (choice without replacement).
            try:
                with np.load(npz_path) as z:
                    names = z["sequence"].astype(str)
                    soa = {k: z[k] for k in ("action", "reward_step", "reward_done", "done", "old_state_iou", "new_state_iou",
                                             "annotated_frames", "next_annotated_frames")}
                pick = np.random.choice(len(names), size=min(len(names), self.memory_size), replace=False)
                names, soa = names[pick], {k: v[pick] for k, v in soa.items()}
                if self.seq_list is not None:
                    assert len(self.seq_list) > 0
                    keep = np.isin(names, list(self.seq_list))
                    names, soa = names[keep], {k: v[keep] for k, v in soa.items()}
                self.frame = None
            except Exception:
                soa = names = None
        if soa is None:
            while True:                              # the writer may be mid-rewrite (reference :43-51)
                try:
                    pool = pd.read_csv(csv_path, index_col=0, low_memory=False)
                    break
                except Exception:
                    print(f"catch some EXCEPTION when try to load {csv_path}")
                    time.sleep(10)
            pool = pool.sample(min(pool.shape[0], self.memory_size))      # shuffle (np.random global state)
            if self.seq_list is not None:
                assert len(self.seq_list) > 0
                pool = pool[pool["sequence"].isin(set(self.seq_list))]
            names = pool["sequence"].to_numpy().astype(str)
            self.frame = pool
            soa = parse_rows(pool)
        if self.seqs:
            unknown = set(names) - set(self.seqs)
            assert not unknown, f"{sorted(unknown)[0]} not in {split} set."
        self.soa = soa
        npool = len(names)
        self.samples_list = [
            dict(action=soa["action"][i], old_state_iou=soa["old_state_iou"][i][None], new_state_iou=soa["new_state_iou"][i][None],
                 annotated_frames=soa["annotated_frames"][i][None], next_annotated_frames=soa["next_annotated_frames"][i][None],
                 reward_step=soa["reward_step"][i], reward_done=soa["reward_done"][i], done=soa["done"][i])
            for i in range(npool)]

    @classmethod
    def from_soa(cls, soa, transform=None):
        """The dataset over an in-memory SoA (the dict parse_rows / synth.replay_transitions produce) instead of memory_pool.csv:
        the same per-sample dicts, hence the same collated batches."""
        self = object.__new__(cls)
        self.transform, self.soa, self.frame, self.seq_list = transform, soa, None, None
        self.samples_list = [
            dict(action=soa["action"][i], old_state_iou=soa["old_state_iou"][i][None], new_state_iou=soa["new_state_iou"][i][None],
                 annotated_frames=soa["annotated_frames"][i][None], next_annotated_frames=soa["next_annotated_frames"][i][None],
                 reward_step=soa["reward_step"][i], reward_done=soa["reward_done"][i], done=soa["done"][i])
            for i in range(len(soa["action"]))]
        return self

    def __len__(self):
        return len(self.samples_list)

    def __getitem__(self, idx):
        sample = self.samples_list[idx]
        return self.transform(sample) if self.transform is not None else sample

    def to_device_replay(self, device):
        return DeviceReplay(self.soa, device)


def load_agent_dataset(cfg, seq_list):
    roots = {"davis": "root_dir_davis", "youtube_vos": "root_dir_scribble_youtube_vos", "combine": "root_dir_combine"}
    root = getattr(cfg.data, roots[cfg.dataset], None) if cfg.dataset in roots else None
    return DAVIS2017AgentTrain(transform=None, split=cfg.data.subset, memory_size=cfg.agent.memory_size,
                               db_root_dir=root, save_result_dir=cfg.agent.save_result_dir, seq_list=seq_list)
