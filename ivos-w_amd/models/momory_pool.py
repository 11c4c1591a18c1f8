"""Replay memory with the reference's class surface and CSV persistence format.

Mirrors /root/reference/models/momory_pool.py:8-162 (``Transition`` field order, ``ReplayMemory`` ring with
``position`` starting at -1, ``memory_pool.csv`` schema = index column + 12 columns, the per-sequence
``sample_th`` filter and the capacity shrink in ``load_from_csv``).  The module name keeps the reference's
spelling ("momory") so imports drop in.  ``DeviceReplay`` is the MI355X-side addition: the same rows as a
device-resident SoA that ``ivosw_replay_gather`` samples minibatches from.

Persistence.  The reference re-serialises the whole DataFrame through pandas on every push
(:126-153: O(N) per transition, ~100 MB of text at 50 k rows) and the dataset re-parses it every third sequence.
Here the CSV image is kept as pre-formatted text rows: ``push_to_csv`` formats ONE row, and the file is rewritten by a
single text join (no pandas) every ``csv_sync_every`` pushes (default 1 = the reference's behaviour: the file on disk is
byte-identical to the reference's after every push; at capacity every row's index label shifts, so the format itself
forbids a pure append).  Each sync also writes ``memory_pool.npz``, the same rows as a binary SoA that
``datasets.agent_dataset`` loads instead of re-parsing the text.
"""
import io
import os
import random
from collections import namedtuple

import numpy as np
import pandas as pd

Transition = namedtuple("Transition", ["state", "action", "next_state", "reward_step", "reward_done", "done",
                                       "state_iou", "next_state_iou", "annotated_frames", "next_annotated_frames"])

_CSV_COLUMNS = ("sequence", "scribble_iter", "n_interaction", "n_interaction_next", "action", "reward_step",
                "reward_done", "done", "state_iou", "next_state_iou", "annotated_frames", "next_annotated_frames")


def _mean_of_joined(strings):
    """'a/b/c' strings -> per-row mean (float64)."""
    return np.array([[float(tok) for tok in s.split("/")] for s in strings], dtype=np.float64).mean(axis=1)


class ReplayMemory:
    def __init__(self, capacity):
        self.capacity = capacity
        self.memory = []
        self.position = -1            # first push lands on slot 0
        self.basename_csv = "memory_pool.csv"
        self.COLUMNS = list(_CSV_COLUMNS)
        self._rows, self._label0, self._memory_pd = [], 0, None      # CSV image: text rows + first index label
        self._pending, self._needs_pandas, self._parsed = 0, False, {}
        self._col_kind = {}                        # pandas dtype of every numeric column of the image: int / float / object
        self.csv_sync_every = 1                    # 1 = rewrite memory_pool.csv on every push, like the reference
        self.seq_list = []

    def __len__(self):
        return len(self.memory)

    # ------------------------------------------------------------------ ring
    def push(self, *fields):
        if len(self.memory) < self.capacity:
            self.memory.append(None)
        self.position = (self.position + 1) % self.capacity
        self.memory[self.position] = Transition(*fields)

    def random_sample(self, batch_size):
        """Unused by the reference's training loop (minibatches come from the CSV), kept for the surface."""
        if batch_size > len(self.memory):
            return None
        picked = random.sample(self.memory, batch_size)
        return Transition(*zip(*picked))

    # ------------------------------------------------------------------ CSV
    def load_from_csv(self, path_to_random_memory_csv, report_save_dir=None, sample_th=0):
        frame = pd.read_csv(path_to_random_memory_csv, index_col=0)[:self.capacity]
        names = frame["sequence"].tolist()
        ordered_unique = list(dict.fromkeys(names))           # first-occurrence order
        if sample_th > 0:
            assert sample_th < 1
            self.seq_list = []
            for seq in ordered_unique:
                rows = frame[frame.sequence == seq]
                if len(rows) == 0:
                    continue
                lo = _mean_of_joined(rows.state_iou.values.tolist()).min()
                hi = _mean_of_joined(rows.next_state_iou.values.tolist()).max()
                if hi - lo > sample_th:
                    self.seq_list.append(seq)
            print(f"the number of available samples under threshold {sample_th}: {len(self.seq_list)}")
        else:
            self.seq_list.extend(ordered_unique)

        cols = {c: frame[c].tolist() for c in self.COLUMNS}
        kept = 0
        allowed = set(self.seq_list)
        for i in range(min(len(names), self.capacity)):
            if sample_th > 0:
                assert len(self.seq_list) > 0
                if names[i] not in allowed:
                    continue
            kept += 1
            where = dict(sequence=names[i], scribble_iter=cols["scribble_iter"][i])
            self.push(dict(where, n_interaction=cols["n_interaction"][i]), cols["action"][i],
                      dict(where, n_interaction=cols["n_interaction_next"][i]), cols["reward_step"][i],
                      cols["reward_done"][i], cols["done"][i], cols["state_iou"][i], cols["next_state_iou"][i],
                      cols["annotated_frames"][i], cols["next_annotated_frames"][i])
        self.capacity = kept           # the reference shrinks the ring to what it kept (:110)

        os.makedirs(report_save_dir, exist_ok=True)
        self.memory_pd = frame[:self.capacity]
        self.memory_pd.to_csv(os.path.join(report_save_dir, self.basename_csv))

    def _row_at(self, pos):
        t = self.memory[pos]
        return {"sequence": t.state["sequence"], "scribble_iter": t.state["scribble_iter"],
                "n_interaction": t.state["n_interaction"], "n_interaction_next": t.next_state["n_interaction"],
                "action": t.action, "reward_step": t.reward_step, "reward_done": t.reward_done, "done": t.done,
                "state_iou": t.state_iou, "next_state_iou": t.next_state_iou,
                "annotated_frames": t.annotated_frames, "next_annotated_frames": t.next_annotated_frames}

    # ---- CSV image as text rows -------------------------------------------------------------------------
    @staticmethod
    def _fmt(v):
        """One cell exactly as ``DataFrame.to_csv`` writes it (QUOTE_MINIMAL), or None if it would need quoting."""
        if isinstance(v, (bool, np.bool_)):
            return "True" if v else "False"
        if isinstance(v, (int, np.integer)):
            return str(int(v))
        if isinstance(v, (float, np.floating)):
            return "" if v != v else repr(float(v))
        t = str(v)
        return None if any(ch in t for ch in ',"\r\n') else t

    _NUMERIC = ("scribble_iter", "n_interaction", "n_interaction_next", "action", "reward_step", "reward_done")

    def _promote(self, col):
        """An int64 column that meets a typed float value becomes float64 for EVERY row, past and future: "0" is rewritten as "0.0"."""
        k = self.COLUMNS.index(col)
        out = []
        for r in self._rows:
            cells = r.split(",")
            c = cells[k]
            if c and "." not in c and "e" not in c and "n" not in c:       # an integer literal ('nan' / 'inf' stay)
                cells[k] = repr(float(int(c)))
            out.append(",".join(cells))
        self._rows = out
        self._parsed = {}

    def _format_row(self, row):
        """One CSV row as the reference's ``concat`` + ``to_csv`` (:139-152) would write it.  pandas types every numeric column
        of the pool: a one-row frame built from a python / numpy SCALAR is int64 or float64, one built from a 0-d ndarray
        (what ``agent_business`` passes for the rewards) is ``object``; concat gives int64 + float64 -> float64 (every row of
        the column is then written as a float, past rows included, and an int pushed later too), anything + object -> object
        (every cell keeps the text of its own type from then on)."""
        cells = []
        for c in self.COLUMNS:
            v = row[c]
            if c in self._NUMERIC and not isinstance(v, (bool, np.bool_)):
                if isinstance(v, np.ndarray) and v.ndim == 0:
                    kind, v = "object", v.item()
                elif isinstance(v, (float, np.floating)):
                    kind = "float"
                elif isinstance(v, (int, np.integer)):
                    kind = "int"
                else:
                    kind = "object"
                have = self._col_kind.get(c) if self._rows else None
                if have is None or have == kind:
                    now = kind
                elif "object" in (have, kind):
                    now = "object"
                else:
                    now = "float"                                  # int64 meets float64
                    if have == "int":
                        self._promote(c)
                    else:
                        v = float(v)
                if now == "float" and kind == "int":
                    v = float(v)
                self._col_kind[c] = now
            cells.append(self._fmt(v))
        return None if any(c is None for c in cells) else ",".join(cells)

    def _rows_from_frame(self, frame):
        """Text rows of a DataFrame exactly as pandas serialises them (labels stripped)."""
        lines = frame.to_csv().split("\n")[1:]
        return [ln.split(",", 1)[1] for ln in lines if ln]

    @property
    def memory_pd(self):
        """The reference's DataFrame image of the pool, materialised on demand from the text rows."""
        if self._memory_pd is None:
            text = "," + ",".join(self.COLUMNS) + "\n" + "".join(f"{self._label0 + i},{r}\n" for i, r in enumerate(self._rows))
            self._memory_pd = pd.read_csv(io.StringIO(text), index_col=0, float_precision="round_trip") if self._rows else pd.DataFrame(columns=self.COLUMNS)
        return self._memory_pd

    @memory_pd.setter
    def memory_pd(self, frame):
        self._memory_pd = frame
        self._rows = self._rows_from_frame(frame) if len(frame) else []
        self._label0 = int(frame.index.min()) if len(frame) else 0
        self._col_kind = {c: ("float" if str(frame[c].dtype).startswith("float") else "int" if str(frame[c].dtype).startswith("int") else "object")
                          for c in self._NUMERIC if c in frame.columns}

    def csv_text(self):
        return "," + ",".join(self.COLUMNS) + "\n" + "".join(f"{self._label0 + i},{r}\n" for i, r in enumerate(self._rows))

    def _write_csv(self, report_save_dir):
        os.makedirs(report_save_dir, exist_ok=True)
        path = os.path.join(report_save_dir, self.basename_csv)
        tmp = path + ".tmp"
        with open(tmp, "w") as f:
            f.write(self.csv_text())
        os.replace(tmp, path)                      # readers never see a half-written file
        self._pending = 0

    def sync_csv(self, report_save_dir, sidecar=True):
        """Write ``memory_pool.csv`` (byte-identical to the reference's file for the same push sequence) and, with
        ``sidecar``, the binary SoA image ``memory_pool.npz`` of the same rows (parsed rows are cached per text row)."""
        self._write_csv(report_save_dir)
        if not sidecar:
            return
        try:
            for r in self._rows:
                if r not in self._parsed:
                    self._parsed[r] = parse_text_rows([r])
            if len(self._parsed) > 2 * len(self._rows) + 64:
                live = set(self._rows)
                self._parsed = {k: v for k, v in self._parsed.items() if k in live}
            keys = ("action", "reward_step", "reward_done", "done", "old_state_iou", "new_state_iou", "annotated_frames",
                    "next_annotated_frames")
            soa = {k: np.concatenate([self._parsed[r][k] for r in self._rows]) for k in keys} if self._rows else parse_text_rows([])
            np.savez(os.path.join(report_save_dir, "memory_pool.npz"), nrows=np.int64(len(self._rows)),
                     sequence=np.array([r.split(",", 1)[0] for r in self._rows]), **soa)
        except Exception:                          # ragged / non-numeric rows: the CSV stays the source of truth
            pass

    def push_to_csv(self, report_save_dir):
        """Append the most recent transition to the CSV image, dropping the oldest row when over capacity
        (reference :126-153: concat with ignore_index, drop the smallest label, rewrite the file)."""
        line = self._format_row(self._row_at(self.position))
        if line is None or self._needs_pandas:
            # a cell needs CSV quoting: fall back to the reference's own pandas path for this pool
            self._needs_pandas = True
            row = pd.DataFrame(data={k: [v] for k, v in self._row_at(self.position).items()}, columns=self.COLUMNS)
            frame = pd.concat([self.memory_pd, row], ignore_index=True) if len(self.memory_pd) else row
            if len(frame) > self.capacity:
                frame = frame.drop(frame.index.min())
            self._memory_pd = frame
            os.makedirs(report_save_dir, exist_ok=True)
            frame.to_csv(os.path.join(report_save_dir, self.basename_csv))
            return
        self._rows.append(line)
        self._label0 = 0                           # concat(ignore_index=True) relabels from 0 ...
        if len(self._rows) > self.capacity:
            self._rows.pop(0)
            self._label0 = 1                       # ... and dropping label 0 leaves 1..capacity
        self._memory_pd = None
        self._pending += 1
        if self._pending >= self.csv_sync_every:
            # every push (the reference's cadence): text only; deferred cadence: also refresh the binary sidecar
            self.sync_csv(report_save_dir, sidecar=self.csv_sync_every > 1)


def parse_text_rows(rows):
    """Pre-formatted CSV rows (no label) -> the SoA dict of ``parse_rows`` without going through pandas."""
    cols = list(zip(*(r.split(",") for r in rows))) if rows else [[] for _ in _CSV_COLUMNS]
    by = dict(zip(_CSV_COLUMNS, cols))

    def mat(col):
        return np.array([[float(tok) for tok in s.split("/")] for s in by[col]], dtype=np.float64).reshape(len(rows), -1)
    return dict(action=np.array(by["action"], dtype=np.int64), reward_step=np.array(by["reward_step"], dtype=np.int64),
                reward_done=np.array(by["reward_done"], dtype=np.float64), done=np.array([d == "True" for d in by["done"]], dtype=bool),
                old_state_iou=mat("state_iou"), new_state_iou=mat("next_state_iou"),
                annotated_frames=mat("annotated_frames"), next_annotated_frames=mat("next_annotated_frames"))


def parse_rows(frame, T=None):
    """CSV frame -> SoA numpy dict (float64 [n,T] columns, int64/float64/bool scalars) — the arithmetic content
    of datasets/agent_dataset.py:71-115 without per-sample python objects."""
    def mat(col):
        return np.array([[float(tok) for tok in str(s).split("/")] for s in frame[col].tolist()], dtype=np.float64)
    out = dict(action=frame["action"].to_numpy(np.int64), reward_step=frame["reward_step"].to_numpy(np.int64),
               reward_done=frame["reward_done"].to_numpy(np.float64), done=frame["done"].to_numpy(bool),
               old_state_iou=mat("state_iou"), new_state_iou=mat("next_state_iou"),
               annotated_frames=mat("annotated_frames"), next_annotated_frames=mat("next_annotated_frames"))
    if T is not None:
        assert out["old_state_iou"].shape[1] == T
    return out


_M64 = (1 << 64) - 1


def draw_indices(seed, counter, B, n):
    """Host mirror of the device-side minibatch draw (``ivosw_replay_draw_gather``): the rows slot 0..B-1 of draw number
    ``counter`` read — a splitmix64 finaliser of (seed, counter, slot) scaled to [0, n) by the high half of a 64 x 64
    multiply.  Integer arithmetic only, so device, C host mirror (``ivosw_replay_draw_index``) and this agree bit for bit."""
    out = np.empty(B, dtype=np.int64)
    for b in range(B):
        z = (seed + 0x9E3779B97F4A7C15 * (counter + 1) + 0xD1B54A32D192ED03 * (b + 1)) & _M64
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
        z ^= z >> 31
        out[b] = (z * n) >> 64
    return out


class DeviceReplay:
    """Device-resident SoA replay buffer; ``sample(idx)`` gathers a minibatch with ``ivosw_replay_gather``,
    ``sample_drawn`` draws the indices on the device as well (``ivosw_replay_draw_gather``)."""

    def __init__(self, soa, device):
        import torch
        from .. import _lib
        self._lib = _lib
        self.device = torch.device(device)
        f32 = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).to(self.device).contiguous()
        self.old_iou, self.new_iou = f32(soa["old_state_iou"]), f32(soa["new_state_iou"])
        self.ann, self.next_ann = f32(soa["annotated_frames"]), f32(soa["next_annotated_frames"])
        self.action = torch.as_tensor(soa["action"], dtype=torch.int64).to(self.device)
        self.reward_step = f32(soa["reward_step"])
        self.reward_done = f32(soa["reward_done"])
        self.n, self.T = self.old_iou.shape

    def __len__(self):
        return self.n

    def sample(self, idx):
        """idx: int64 device tensor [B] -> dict of device tensors (state/new_state [B,T,2] fp32, ...)."""
        import torch
        L = self._lib
        B = idx.numel()
        state = torch.empty(B, self.T, 2, dtype=torch.float32, device=self.device)
        new_state = torch.empty_like(state)
        act = torch.empty(B, dtype=torch.int64, device=self.device)
        rs = torch.empty(B, dtype=torch.float32, device=self.device)
        rd = torch.empty_like(rs)
        L.check(L.lib().ivosw_replay_gather(
            L.dptr(self.old_iou), L.dptr(self.new_iou), L.dptr(self.ann), L.dptr(self.next_ann), L.dptr(self.action),
            L.dptr(self.reward_step), L.dptr(self.reward_done), L.dptr(idx, torch.int64), B, self.T,
            L.dptr(state), L.dptr(new_state), L.dptr(act), L.dptr(rs), L.dptr(rd), L.stream_ptr(self.device)),
            "replay_gather")
        return dict(state=state, new_state=new_state, action=act, reward_step=rs, reward_done=rd)

    def draw_state(self, seed, counter=0):
        """The 16-byte device state of the on-device draw: {uint64 seed, uint32 counter, uint32 owned by the library}."""
        import torch
        raw = np.zeros(2, dtype=np.uint64)
        raw[0] = np.uint64(seed & _M64)
        raw[1] = np.uint64(counter & 0xFFFFFFFF)
        assert self._lib.lib().ivosw_replay_draw_state_bytes() == raw.nbytes
        return torch.from_numpy(raw.view(np.uint8).copy()).to(self.device)

    def sample_drawn(self, B, draw_state, out=None):
        """Draw B rows on the device (advancing ``draw_state``'s counter) and gather them; ``out`` = preallocated dict with
        idx / state / new_state / action / reward_step / reward_done (what a captured graph records), else fresh tensors."""
        import torch
        L = self._lib
        if out is None:
            out = dict(idx=torch.empty(B, dtype=torch.int64, device=self.device),
                       state=torch.empty(B, self.T, 2, dtype=torch.float32, device=self.device),
                       new_state=torch.empty(B, self.T, 2, dtype=torch.float32, device=self.device),
                       action=torch.empty(B, dtype=torch.int64, device=self.device),
                       reward_step=torch.empty(B, dtype=torch.float32, device=self.device),
                       reward_done=torch.empty(B, dtype=torch.float32, device=self.device))
        L.check(L.lib().ivosw_replay_draw_gather(
            L.dptr(self.old_iou), L.dptr(self.new_iou), L.dptr(self.ann), L.dptr(self.next_ann), L.dptr(self.action),
            L.dptr(self.reward_step), L.dptr(self.reward_done), L.dptr(draw_state, torch.uint8), self.n, B, self.T,
            L.dptr(out["idx"], torch.int64), L.dptr(out["state"]), L.dptr(out["new_state"]), L.dptr(out["action"], torch.int64),
            L.dptr(out["reward_step"]), L.dptr(out["reward_done"]), L.stream_ptr(self.device)), "replay_draw_gather")
        return out
