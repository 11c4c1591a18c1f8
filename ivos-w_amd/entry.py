"""Entry-point runtime: what ``train_agent.py`` / ``eval_agent_{manet,atnet,ipn}.py`` at the repo root drive.

The reference's entry scripts (/root/reference/train_agent.py:119-379, eval_agent_manet.py:246-480) are sacred experiments
over the DAVIS-interactive session, a scribble robot and an external VOS clone.  None of those packages exists in this
image (sacred, easydict, davisinteractive, cv2, the MANet/ATNet/IPN clones, the DAVIS frames), so the scripts here are
written from scratch around the same loop

    session.next -> scribbles -> VOS segmentation -> sequence_metric -> recommend_frame -> submit_masks
                 -> [train: agent_business -> update_agent] -> summary.json / agent checkpoint

with two interchangeable back ends for the parts that are NOT on the hot path:

  * ``synthetic=1`` (default when the real stack is missing and ``synthetic`` was not set to 0): ``SyntheticDavis`` (seeded
    moving-blob videos with ground truth), ``SyntheticSession`` (the session protocol and the J&F curve bookkeeping of
    ``DavisInteractiveSession``; the "scribble" is the recommended frame's ground truth) and ``StandInVOS`` (a
    deterministic segmentation model whose error grows with the distance to the nearest annotated frame, producing
    stride-4 logits that go through the product's ``seg_epilogue`` kernel).
  * the real stack: refused with an explicit message listing what is missing (``require_real_stack``) — nothing is faked
    silently.

Everything ON the hot path is the product: AssessNet / Agent (HIP), ``recommend_frame``, ``agent_business``,
``sequence_metric`` (HIP J&F), ``ReplayMemory`` / ``load_agent_dataset``, ``save_agent_checkpoint``.
The CLI keeps sacred's ``with key=value`` form: ``python eval_agent_manet.py with setting=wild method=ours dataset=davis``.
"""
import copy
import importlib.util
import json
import os
import sys
import time

import numpy as np
import torch


class AttrDict(dict):
    """dict with attribute access (what the reference gets from easydict)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _attr(d):
    return AttrDict({k: _attr(v) if isinstance(v, dict) else v for k, v in d.items()})


# the keys the loop reads (the reference keeps them in configs/config.yaml); values are that file's defaults
DEFAULTS = dict(
    seed=0, gpu_id=0, phase="eval", setting="wild", method="ours", num_epochs=1, dataset="davis", ckpt_dir="weights",
    vos_adapter="",                    # module on PYTHONPATH that wraps the VOS backbone for the real stack (default ivosw_vos_<backbone>)
    synthetic=-1,                      # -1 auto (synthetic when the real stack is missing), 0 real stack only, 1 synthetic
    precision="bf16",                  # AssessNet mode: bf16 throughput (scores within 4e-3) / bf16x3 fast parity (1e-4) / fp32 exact parity
    report_save_dir="results",
    eval_max_nb_interactions=8,        # the eval scripts fix 8 interactions on the val subset (eval_agent_manet.py:64-65)
    data=dict(num_workers=0, root_dir_davis="data/DAVIS", root_dir_scribble_youtube_vos="data/Scribble_Youtube_VOS",
              subset="train", len_subseq=25),
    davis_interactive=dict(metric="J_AND_F", allow_repeat=1, max_nb_interactions=5, max_time_per_interaction=0, combine_th=0.4),
    agent=dict(save_result_dir="train", reward_csv="reward.csv", pretrain_csv="pretrain.csv", sample_th=0.05, optimizer="adam",
               lr=5e-6, lr_pow=0.9, momentum=0.9, weight_decay=5e-4, memory_size=100000, gamma=0.95, eps_start=0.7, eps_end=0.25,
               eps_k=5, eps_decay=500, update_rate=0.05, train_batch_size=32),
    synth=dict(n_sequences=3, n_frames=30, height=120, width=216, max_objects=3, baseline_runs=30),
)


def _coerce(text):
    for cast in (int, float):
        try:
            return cast(text)
        except ValueError:
            pass
    return {"true": True, "false": False, "none": None}.get(text.lower(), text)


def parse_cli(argv, **overrides):
    """sacred's surface: ``script.py [--config file.yaml] [with] a=1 b.c=x``.  Unknown top-level keys are accepted (sacred adds
    them too); a dotted key must address an existing section."""
    cfg = copy.deepcopy(DEFAULTS)
    cfg.update(overrides)
    args = list(argv)
    if "--config" in args:
        import yaml
        i = args.index("--config")
        with open(args[i + 1]) as f:
            for k, v in (yaml.safe_load(f) or {}).items():
                if isinstance(v, dict) and isinstance(cfg.get(k), dict):
                    cfg[k].update(v)
                else:
                    cfg[k] = v
        del args[i:i + 2]
    for tok in args:
        if tok == "with":
            continue
        if "=" not in tok:
            raise SystemExit(f"unrecognised argument {tok!r}: expected `with key=value ...`")
        key, val = tok.split("=", 1)
        node, parts = cfg, key.split(".")
        for p in parts[:-1]:
            if not isinstance(node.get(p), dict):
                raise SystemExit(f"unknown config section {p!r} in {key!r}")
            node = node[p]
        node[parts[-1]] = _coerce(val)
    return _attr(cfg)


# ------------------------------------------------------------------------------------------ the real stack, or a clear refusal
# What eval_agent_{manet,atnet,ipn}.py need besides this build: the davisinteractive package (session, scribble robot, Davis index),
# cv2 (frame decoding), the DAVIS frames, and the VOS backbone.  The backbone and its glue are third-party clones the reference
# keeps outside its own tree (README.md:35-41; SURVEY section 2 marks them OUT OF SCOPE): they enter through ONE small adapter module the
# caller puts on PYTHONPATH — ``cfg.vos_adapter`` (default ``ivosw_vos_<backbone>``) exposing
#
#     build(cfg, device) -> adapter
#     adapter.start_sequence(sequence, n_frame, n_objects, h, w)          once per (sequence, scribble) sample
#     adapter.segment(sequence, scribbles, annotated_frame, first_scribble, n_interaction)
#         -> (labels [n,H,W] integer label maps, all_P [n,O+1,H,W] float32 probabilities on the device)
#
# i.e. exactly what the reference's loop computes between "interaction initial" and "frame recommendation"
# (eval_agent_manet.py:321-396).  For MANet, ``ivos_w_amd.utils.utils_manet.get_results`` is the drop-in for the reference's
# get_results inside such an adapter.
REAL_STACK = ("davisinteractive", "cv2")


def adapter_module_name(backbone, cfg):
    return str(cfg.get("vos_adapter") or f"ivosw_vos_{backbone.lower()}")


def missing_real_stack(backbone, cfg):
    missing = []
    for mod in REAL_STACK + (adapter_module_name(backbone, cfg),):
        try:
            if mod not in sys.modules and importlib.util.find_spec(mod) is None:
                missing.append(f"python module {mod}")
        except (ImportError, ValueError):
            missing.append(f"python module {mod}")
    root = cfg.data.root_dir_davis
    if not os.path.isdir(os.path.join(root, "JPEGImages", "480p")):
        missing.append(f"DAVIS frames under {root}/JPEGImages/480p")
    return missing


def choose_backend(backbone, cfg):
    """-> True for the synthetic back end, False for the real stack (davisinteractive session + the caller's VOS adapter).
    ``synthetic=0`` with a missing stack stops with the list of what is missing."""
    missing = missing_real_stack(backbone, cfg)
    if cfg.synthetic == 1 or (cfg.synthetic == -1 and missing):
        if missing and cfg.synthetic == -1:
            print(f"[ivos-w] the {backbone} evaluation stack is not available here ({'; '.join(missing)}): running the SYNTHETIC "
                  "session with the stand-in VOS model (hot path = the product kernels). Pass synthetic=0 to insist on the real stack.")
        return True
    if missing:
        raise SystemExit(f"[ivos-w] synthetic=0 but the {backbone} stack is incomplete: " + "; ".join(missing))
    return False


# ------------------------------------------------------------------------------------------ synthetic data + session
class SyntheticDavis:
    """Seeded moving-blob videos with ground-truth label maps: the stand-in for davisinteractive.dataset.Davis + the frames."""

    def __init__(self, cfg, device):
        s, self.device = cfg.synth, device
        rs = np.random.RandomState(1000 + int(cfg.seed))
        self.dataset, self._params = {}, {}
        for i in range(int(s.n_sequences)):
            name = f"synth-{i:02d}"
            n_obj = 1 + rs.randint(int(s.max_objects))
            self.dataset[name] = dict(num_frames=int(s.n_frames), num_objects=n_obj, image_size=(int(s.width), int(s.height)))
            self._params[name] = dict(c0=rs.uniform(0.25, 0.75, (n_obj, 2)), v=rs.uniform(-0.012, 0.012, (n_obj, 2)),
                                      r=rs.uniform(0.10, 0.22, (n_obj, 2)), ph=rs.uniform(0, 6.28, n_obj),
                                      col=rs.uniform(0.2, 1.0, (n_obj + 1, 3)), seed=int(rs.randint(1 << 30)))
        self.sets = {cfg.data.subset: list(self.dataset)}
        self._cache = {}

    def _render(self, name):
        if name in self._cache:
            return self._cache[name]
        info, p = self.dataset[name], self._params[name]
        n, (w, h), O = info["num_frames"], info["image_size"], info["num_objects"]
        dev = self.device
        yy, xx = torch.meshgrid(torch.linspace(0, 1, h, device=dev), torch.linspace(0, 1, w, device=dev), indexing="ij")
        t = torch.arange(n, device=dev, dtype=torch.float32)[:, None, None]
        label = torch.zeros(n, h, w, dtype=torch.uint8, device=dev)
        g = torch.Generator(device="cpu").manual_seed(p["seed"])
        frames = (0.35 + 0.1 * torch.rand(n, 3, h, w, generator=g)).to(dev)
        frames += 0.15 * torch.sin(12.0 * xx + 7.0 * yy)[None, None]
        for o in range(O):
            cy = float(p["c0"][o, 0]) + float(p["v"][o, 0]) * t + 0.05 * torch.sin(0.3 * t + float(p["ph"][o]))
            cx = float(p["c0"][o, 1]) + float(p["v"][o, 1]) * t + 0.05 * torch.cos(0.2 * t + float(p["ph"][o]))
            inside = ((yy[None] - cy) / float(p["r"][o, 0])) ** 2 + ((xx[None] - cx) / float(p["r"][o, 1])) ** 2 < 1.0
            label[inside] = o + 1
            col = torch.tensor(p["col"][o + 1], dtype=torch.float32, device=dev)[None, :, None, None]
            frames = torch.where(inside[:, None], 0.6 * col + 0.4 * frames, frames)
        self._cache[name] = (frames.clamp_(0, 1).cpu(), label)      # frames on the host, like the reference's all_F
        return self._cache[name]

    def load_frames(self, name):
        return self._render(name)[0]

    def load_annotations(self, name):
        return self._render(name)[1]                                  # uint8 [n,H,W] on the device (the J/F kernels read it there)


class SyntheticSession:
    """The protocol of DavisInteractiveSession as the entry scripts use it (next / get_scribbles / submit_masks /
    get_global_summary, ``samples``), with the J&F curve bookkeeping; a "scribble" names the annotated frame only."""

    def __init__(self, davis, subset, metric_to_optimize, max_nb_interactions, report_save_dir, seed=0, rounds=1):
        self.davis, self.metric, self.max_nb = davis, metric_to_optimize, int(max_nb_interactions)
        self.report_save_dir = report_save_dir
        rs = np.random.RandomState(77 + seed)
        self.samples = [(seq, int(rs.randint(davis.dataset[seq]["num_frames"]))) for _ in range(rounds) for seq in davis.sets[subset]]
        self._i, self._k, self._frame = -1, 0, None
        self.records = []                                             # (sequence, sample index, interaction, mean metric, seconds)
        self._tic = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def next(self):
        if self._i >= 0 and 0 < self._k < self.max_nb:
            return True                                               # same sample, next interaction
        self._i += 1
        self._k = 0
        if self._i >= len(self.samples):
            return False
        self._frame = self.samples[self._i][1]
        return True

    def get_scribbles(self, only_last=False):
        seq = self.samples[self._i][0]
        n = self.davis.dataset[seq]["num_frames"]
        scribbles = dict(sequence=seq, annotated_frame=self._frame,
                         scribbles=[[dict(frame=self._frame)] if f == self._frame else [] for f in range(n)])
        self._tic = time.time()
        return seq, scribbles, self._k == 0

    def submit_masks(self, masks, next_scribble_frame_candidates=None):
        """masks: predicted labels [n,H,W] (device tensor or numpy).  The next scribble lands on the first candidate."""
        from .utils import misc
        seq = self.samples[self._i][0]
        gt = self.davis.load_annotations(seq)
        m = misc.sequence_metric(self.metric, gt, masks, self.davis.dataset[seq]["num_objects"])
        self._k += 1
        self.records.append((seq, self._i, self._k, float(np.mean(m)), time.time() - (self._tic or time.time())))
        if next_scribble_frame_candidates:
            self._frame = int(next_scribble_frame_candidates[0])
        return m

    def get_global_summary(self):
        curve, times = [], []
        for k in range(1, self.max_nb + 1):
            vals = [r[3] for r in self.records if r[2] == k]
            curve.append(float(np.mean(vals)) if vals else float("nan"))
            times.append(float(np.mean([r[4] for r in self.records if r[2] == k])) if vals else 0.0)
        # like the package's curve, one trailing point beyond the last interaction (the scripts drop it with [:-1])
        return dict(curve={self.metric: curve + curve[-1:], "time": list(np.cumsum(times)) + [float(np.sum(times))]},
                    metric_at_threshold={self.metric: curve[-1], "threshold": 60})


class StandInVOS:
    """Deterministic stand-in for the VOS backbone: stride-4 logits whose error grows with the distance to the nearest annotated
    frame (the ground truth shifted by 1.5 px per frame of distance, plus seeded noise).  Annotated frames come out (almost) exact,
    so annotating the worst frame helps most — the structure the agent learns from."""

    def __init__(self, device, seed=0):
        self.device, self.seed = device, seed

    def logits(self, gt_u8, n_objects, annotated_frames):
        n, h, w = gt_u8.shape
        C = n_objects + 1
        onehot = torch.nn.functional.one_hot(gt_u8.long().clamp_(0, C - 1), C).permute(0, 3, 1, 2).float()
        low = torch.nn.functional.avg_pool2d(onehot, 4)
        ann = torch.as_tensor(sorted(set(int(a) for a in annotated_frames)), device=gt_u8.device)
        dist = (torch.arange(n, device=gt_u8.device)[:, None] - ann[None]).abs().min(1)[0]
        g = torch.Generator(device="cpu").manual_seed(self.seed + 31 * len(annotated_frames))
        noise = torch.randn(low.shape, generator=g).to(gt_u8.device)
        out = torch.empty_like(low)
        for f in range(n):
            d = int(dist[f])
            sh = int(round(1.5 * d / 4 * 4)) // 4 if d else 0
            out[f] = torch.roll(low[f], shifts=(sh, -sh), dims=(1, 2))
        return 8.0 * (out - 0.5) + noise * (0.3 + 0.25 * dist.float())[:, None, None, None]


# ------------------------------------------------------------------------------------------ building the hot-path objects
def build_hot_path(cfg, device, need_assess):
    from .models.agent import Agent
    from .models.assessment import AssessNet
    from .utils import misc
    agent = Agent(device=device, cfg=cfg)
    misc.load_agent_checkpoint(agent, cfg.ckpt_dir, device="cpu")      # never raises; random init when there is no file
    assess_net = None
    if need_assess:
        assess_net = AssessNet(precision=cfg.precision)
        ck = os.path.join(cfg.ckpt_dir, "assess_net.pt")
        if not misc.load_network_checkpoint(ck, assess_net, device="cpu"):
            from . import synth
            print(f"[ivos-w] no {ck}: AssessNet runs on the seeded synthetic weights")
            assess_net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.assessnet_state_dict(0, spread=True).items()})
        assess_net = assess_net.to(device).eval()
    return agent, assess_net


def _segment(vos, davis, seq, n_objects, annotated, store):
    from .utils import utils_manet
    gt = davis.load_annotations(seq)
    lg = vos.logits(gt, n_objects, annotated)
    utils_manet.seg_epilogue(lg, gt.shape[1], gt.shape[2], store, 0)
    return store.labels_u8, store.all_P


# ------------------------------------------------------------------------------------------ eval_agent_*  (synthetic back end)
def run_eval(cfg, backbone="MANet"):
    """The loop of eval_agent_manet.py:246-480 on the synthetic back end.  Writes <report_save_dir>/summary.json =
    {"auc", "curve": {metric: [...]}} like the reference and returns it."""
    from .utils import misc, utils_agent, utils_manet
    cfg.phase = "eval"
    cfg.data.subset = "val"
    if not torch.cuda.is_available():
        raise SystemExit("[ivos-w] the hot path needs an MI355X (no CPU fallback)")
    device = torch.device(f"cuda:{cfg.gpu_id}")
    if not choose_backend(backbone, cfg):
        return run_eval_real(cfg, backbone, device)
    misc.set_random_seed(int(cfg.seed))
    davis = SyntheticDavis(cfg, device)
    needs_assess = cfg.setting == "wild" and cfg.method in ("ours", "worst")
    agent, assess_net = build_hot_path(cfg, device, needs_assess)
    agent.set_eval()
    vos = StandInVOS(device, seed=int(cfg.seed))
    metric = cfg.davis_interactive.metric
    max_nb = int(cfg.eval_max_nb_interactions)
    report_dir = os.path.join(cfg.report_save_dir, backbone, cfg.setting, cfg.dataset, cfg.method)
    os.makedirs(report_dir, exist_ok=True)
    corr_all, rec_time, seen_seq = misc.AverageMeter(), misc.AverageMeter(), {}
    with SyntheticSession(davis, cfg.data.subset, metric, max_nb, report_dir, seed=int(cfg.seed)) as sess:
        while sess.next():
            sequence, scribbles, first_scribble = sess.get_scribbles(only_last=True)
            if first_scribble:
                info = davis.dataset[sequence]
                n_frame, n_objects = info["num_frames"], info["num_objects"]
                w, h = info["image_size"]
                next_frame = first_frame = scribbles["annotated_frame"]
                seen_seq[sequence] = seen_seq.get(sequence, 0) + 1
                all_F = davis.load_frames(sequence)                   # host tensor, built once per sequence (the frame cache key)
                prev_frames = None if cfg.davis_interactive.allow_repeat > 0 else [next_frame]
                annotated = [next_frame]
                quality_pred = np.zeros(n_frame) if needs_assess else None
                store = utils_manet.ProbStore(n_frame, n_objects + 1, h, w, device)
                n_interaction = 1
            else:
                annotated.append(next_frame)
                n_interaction += 1
            labels, all_P = _segment(vos, davis, sequence, n_objects, annotated, store)
            quality = misc.sequence_metric(metric, davis.load_annotations(sequence), labels, n_objects)
            tic = time.time()
            next_frame = utils_agent.recommend_frame(
                cfg, assess_net, agent, device, n_frame=n_frame, n_objects=n_objects, all_F=all_F, all_P=all_P,
                new_masks_quality=quality, prev_frames=prev_frames, annotated_frames_list=copy.deepcopy(annotated),
                mask_quality=quality_pred, first_frame=first_frame, max_nb_interactions=max_nb)
            next_frame = int(next_frame)
            if prev_frames is not None:
                prev_frames.append(next_frame)
            rec_time.update(time.time() - tic)
            sess.submit_masks(labels, next_scribble_frame_candidates=[next_frame])
            corr = float(np.corrcoef([quality, quality_pred])[0, 1]) if quality_pred is not None else float("nan")
            corr_all.update(0.0 if np.isnan(corr) else corr)
            print(f"avg_{metric}: {quality.mean() * 100:.2f} rec_time:{rec_time.val * 1e3:.1f} ms next_frame: {next_frame:2d} "
                  f"[{int((quality < quality[next_frame]).sum()) + 1:2d}/{n_frame:2d}] corr: {corr:.2f} "
                  f"seq: {sequence}_{seen_seq[sequence]} [{n_interaction:2d}/{max_nb:2d}]")
        gs = sess.get_global_summary()
    curve = gs["curve"][metric][:-1]
    auc = float(np.trapz(curve) / (len(curve) - 1)) if len(curve) > 1 else float(curve[0])
    summary = {"auc": auc, "curve": {metric: curve}}
    with open(os.path.join(report_dir, "summary.json"), "w") as fp:
        json.dump(summary, fp)
    print(f"# global_summary: auc:{auc * 100:.4f}  recommend_frame avg {rec_time.avg * 1e3:.2f} ms  frame-cache uploads "
          f"{utils_agent.frame_cache.uploads}\n# {metric}: " + " ".join(f"{v * 100:.2f}" for v in curve))
    summary["backend"] = "synthetic"
    summary["report_dir"] = report_dir
    return summary


# ------------------------------------------------------------------------------------------ eval_agent_*  (real stack)
def run_eval_real(cfg, backbone, device):
    """The loop of eval_agent_manet.py:246-480 on the REAL stack: davisinteractive's DavisInteractiveSession and scribble robot, the
    DAVIS frames from disk, the caller's VOS adapter for the segmentation, and this build's hot path for everything the reference
    owns in that loop: ``sequence_metric`` (J / F), ``recommend_frame`` (AssessNet + agent), the checkpoint loaders.  Writes
    <report_save_dir>/<backbone>/<setting>/<dataset>/<method>/summary.json like the reference (eval_agent_manet.py:470-480)."""
    import cv2
    from davisinteractive import utils as interactive_utils
    from davisinteractive.dataset import Davis
    from davisinteractive.session import DavisInteractiveSession
    from .utils import misc, utils_agent
    adapter_mod = importlib.import_module(adapter_module_name(backbone, cfg))
    misc.set_random_seed(int(cfg.seed))
    root = cfg.data.root_dir_davis
    needs_assess = cfg.setting == "wild" and cfg.method in ("ours", "worst")
    agent, assess_net = build_hot_path(cfg, device, needs_assess)
    agent.set_eval()
    vos = adapter_mod.build(cfg, device)
    metric = cfg.davis_interactive.metric
    max_nb = int(cfg.eval_max_nb_interactions)
    report_dir = os.path.join(cfg.report_save_dir, backbone, cfg.setting, cfg.dataset, cfg.method)
    os.makedirs(report_dir, exist_ok=True)
    davis = Davis(root)
    seen_seq, frames_of = {}, {}
    rec_time, corr_all, final_q = misc.AverageMeter(), misc.AverageMeter(), misc.AverageMeter()
    with DavisInteractiveSession(host="localhost", davis_root=root, subset=cfg.data.subset, metric_to_optimize=metric,
                                 max_nb_interactions=max_nb, max_time=int(cfg.davis_interactive.max_time_per_interaction) or None,
                                 report_save_dir=report_dir) as sess:
        while sess.next():
            sequence, scribbles, first_scribble = sess.get_scribbles(only_last=True)
            annotated = interactive_utils.scribbles.annotated_frames(scribbles)
            if first_scribble:
                n_objects = Davis.dataset[sequence]["num_objects"]
                gt_masks = davis.load_annotations(sequence)
                assert len(annotated) > 0
                next_frame = first_frame = annotated[0]
                seen_seq[sequence] = seen_seq.get(sequence, 0) + 1
                if sequence not in frames_of:                  # decoded once per sequence: also the key of the device frame cache
                    frames_of.clear()                          # davisinteractive serves a sequence's samples consecutively: keep ONE decoded
                                                               # video (0.35 GB of fp32 at 480p), as the reference holds one all_F — not the whole set
                    jdir = os.path.join(root, "JPEGImages", "480p", sequence)
                    frames_of[sequence] = torch.from_numpy(np.ascontiguousarray(np.stack(
                        [np.asarray(cv2.imread(os.path.join(jdir, f)), dtype=np.float32)[:, :, [2, 1, 0]] / 255. for f in sorted(os.listdir(jdir))],
                        0).transpose(0, 3, 1, 2)))
                all_F = frames_of[sequence]
                n_frame = len(scribbles["scribbles"])
                h, w = int(all_F.shape[2]), int(all_F.shape[3])
                prev_frames = None if cfg.davis_interactive.allow_repeat > 0 else [next_frame]
                annotated_list = [next_frame]
                quality_pred = np.zeros(n_frame) if needs_assess else None
                vos.start_sequence(sequence, n_frame, n_objects, h, w)
                n_interaction = 1
            else:
                annotated_list.append(next_frame)
                n_interaction += 1
            scribbles["annotated_frame"] = next_frame
            with torch.no_grad():
                labels, all_P = vos.segment(sequence, scribbles, next_frame, first_scribble, n_interaction)
            new_masks = labels.cpu().numpy() if torch.is_tensor(labels) else np.asarray(labels)
            quality = misc.sequence_metric(metric, gt_masks, new_masks, n_objects)
            tic = time.time()
            next_frame = int(utils_agent.recommend_frame(
                cfg, assess_net, agent, device, n_frame=n_frame, n_objects=n_objects, all_F=all_F, all_P=all_P, new_masks_quality=quality,
                prev_frames=prev_frames, annotated_frames_list=copy.deepcopy(annotated_list), mask_quality=quality_pred,
                first_frame=first_frame, max_nb_interactions=max_nb))
            if prev_frames is not None:
                prev_frames.append(next_frame)
            rec_time.update(time.time() - tic)
            sess.submit_masks(new_masks, next_scribble_frame_candidates=[next_frame])
            corr = float(np.corrcoef([quality, quality_pred])[0, 1]) if quality_pred is not None else float("nan")
            corr_all.update(0.0 if np.isnan(corr) else corr)
            print(f"avg_{metric}: {quality.mean() * 100:.2f} rec_time:{rec_time.val:.2f} next_frame: {next_frame:2d} "
                  f"[{int((quality < quality[next_frame]).sum()) + 1:2d}/{n_frame:2d}] corr: {corr:.2f} ({corr_all.avg:.2f}) "
                  f"seq: {sequence}_{seen_seq[sequence]} [{n_interaction:2d}/{max_nb:2d}]")
            if n_interaction == max_nb:
                final_q.update(float(quality.mean()) * 100)
        gs = sess.get_global_summary()
    curve = list(gs["curve"][metric][:-1])
    auc = float(np.trapz(curve) / (len(curve) - 1)) if len(curve) > 1 else float(curve[0])
    summary = {"auc": auc, "curve": {metric: curve}}
    with open(os.path.join(report_dir, "summary.json"), "w") as fp:
        json.dump(summary, fp)
    print(f"# final avg {metric}: {final_q.avg:.4f}  final avg corr: {corr_all.avg:.4f}\n# global_summary: auc:{auc * 100:.4f}\n# {metric}: "
          + " ".join(f"{v * 100:.2f}" for v in curve))
    summary["backend"] = "real"
    summary["report_dir"] = report_dir
    return summary


# ------------------------------------------------------------------------------------------ train_agent (synthetic back end)
def _episode(cfg, davis, vos, sess, agent, device, df, train_loader_fn, policy, seen=None):
    """Interactions of ONE session sample on a len_subseq window, reference flow (train_agent.py:150-330): oracle state
    (true J&F), recommend -> submit -> agent_business.  ``policy`` = 'random' (baseline / pretrain collection) or 'ours'."""
    from .utils import misc, utils_agent, utils_manet
    metric, max_nb = cfg.davis_interactive.metric, int(cfg.davis_interactive.max_nb_interactions)
    out = dict(losses=[], rewards_done=[], finals=[])
    state = dict(seen=seen if seen is not None else {})
    while sess.next():
        sequence, scribbles, first_scribble = sess.get_scribbles(only_last=False)
        if first_scribble:
            info = davis.dataset[sequence]
            n_objects = info["num_objects"]
            w, h = info["image_size"]
            state["seen"][sequence] = state["seen"].get(sequence, 0) + 1
            first_frame = scribbles["annotated_frame"]
            len_subseq = min(int(cfg.data.len_subseq), info["num_frames"])
            subseq = utils_agent.gen_subseq(first_frame, info["num_frames"], len_subseq)
            gt = davis.load_annotations(sequence)[torch.as_tensor(subseq, device=device)].contiguous()
            next_frame = subseq.index(first_frame)
            prev_frames, annotated = [next_frame], [next_frame]
            store = utils_manet.ProbStore(len_subseq, n_objects + 1, h, w, device)
            n_interaction, old_frame, old_meta, old_metric, repeat = 1, None, None, None, None
            loader = train_loader_fn(state["seen"][sequence]) if train_loader_fn else None
        else:
            counts = np.zeros(len(new_metric))
            for i in annotated:
                counts[i] += 1
            repeat = next_frame not in list(np.where(counts == counts.min())[0])
            annotated.append(next_frame)
            old_frame, old_meta, old_metric = next_frame, new_meta, new_metric
            n_interaction += 1
        lg = vos.logits(gt, n_objects, annotated)
        utils_manet.seg_epilogue(lg, h, w, store, 0)
        new_metric = misc.sequence_metric(metric, gt, store.labels_u8, n_objects)
        rcfg = AttrDict(cfg, setting="oracle" if policy == "ours" else "wild", method=policy)
        next_frame = int(utils_agent.recommend_frame(
            rcfg, None, agent, device, n_frame=len_subseq, n_objects=n_objects, all_F=None, all_P=store.all_P,
            new_masks_quality=new_metric, prev_frames=prev_frames, annotated_frames_list=copy.deepcopy(annotated), mask_quality=None,
            first_frame=next_frame, max_nb_interactions=max_nb))
        prev_frames.append(next_frame)
        full = davis.load_annotations(sequence).clone()
        full[torch.as_tensor(subseq, device=device)] = store.labels_u8
        sess.submit_masks(full, next_scribble_frame_candidates=[subseq[next_frame]])
        new_meta = dict(sequence=sequence, scribble_iter=state["seen"][sequence], n_interaction=n_interaction)
        loss, r_step, r_done = utils_agent.agent_business(
            cfg, agent, max_nb, n_interaction, first_scribble=first_scribble, old_masks_metric=old_metric, new_masks_metric=new_metric,
            old_frame=old_frame, next_frame=next_frame, sequence=sequence, seen_seq=state["seen"], repeat_selection=repeat, df=df,
            annotated_frames_list=annotated, old_masks_meta=old_meta, new_masks_meta=new_meta,
            report_save_dir=cfg.agent.save_result_dir, agent_train_loader=loader)
        if n_interaction == max_nb:
            out["finals"].append(float(new_metric.mean()))
            out["rewards_done"].append(float(r_done))
            if float(loss) > 0:
                out["losses"].append(float(loss))
    return out


def bootstrap_synthetic_pool(cfg, davis, vos, device):
    """What the reference's baseline + pretrain phases leave on disk before train_agent.py starts (train_agent.py:90-97):
    ``reward.csv`` (30 random-policy episodes per (sequence, scribble_iter % 3): the reward baseline) and ``pretrain.csv``
    (their transitions: the initial replay pool) — produced here by running the random policy on the synthetic session."""
    import pandas as pd
    from .models.agent import Agent
    save_dir = cfg.agent.save_result_dir
    os.makedirs(save_dir, exist_ok=True)
    scratch = AttrDict(cfg, phase="pretrain")
    collector = Agent(device=device, cfg=scratch)
    collector.memory_pool.csv_sync_every = 256
    runs = int(cfg.synth.baseline_runs)
    sess = SyntheticSession(davis, cfg.data.subset, cfg.davis_interactive.metric, cfg.davis_interactive.max_nb_interactions, save_dir,
                            seed=1234, rounds=3 * runs)
    _episode(scratch, davis, vos, sess, collector, device, None, None, "random")
    collector.memory_pool.sync_csv(save_dir)
    pool = os.path.join(save_dir, collector.memory_pool.basename_csv)
    df = pd.read_csv(pool, index_col=0)
    df.to_csv(os.path.join(save_dir, cfg.agent.reward_csv))
    df.to_csv(os.path.join(save_dir, cfg.agent.pretrain_csv))
    return len(df)


def run_train(cfg):
    """train_agent.py:119-379 on the synthetic back end: replay pool from pretrain.csv, epochs over the session in the
    oracle/ours setting, agent_business -> update_agent at the end of every episode, agent.pt per epoch."""
    import pandas as pd
    from torch.utils.data import DataLoader
    from .datasets.agent_dataset import load_agent_dataset
    from .models.agent import Agent
    from .utils import misc
    cfg.phase = "train"
    if not torch.cuda.is_available():
        raise SystemExit("[ivos-w] the hot path needs an MI355X (no CPU fallback)")
    # Data parallel (BASELINE configs[3]; the reference has none, models/agent.py:90-92 is commented out): launched under
    # torch.distributed.run, every rank plays the SAME episodes (same seeds -> the same experience, the same replay pool, the same
    # epsilon / target-sync coins) and the exchange happens in the update loop: the shuffled minibatches of an episode are dealt
    # out to the ranks (parallel.RankBatchSampler), gradients are summed over the ranks and averaged inside clamp + Adam
    # (Agent.apply_gradients).  Rank 0 alone writes agent.pt / memory_pool.csv / the summaries into the configured directories;
    # the other ranks keep their (identical) working files under <save_result_dir>/rank<r>.
    from . import parallel
    rank, world = 0, 1
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        rank, world, device = parallel.init()
        if rank > 0:
            sys.stdout = open(os.devnull, "w")
    else:
        device = torch.device(f"cuda:{cfg.gpu_id}")
    if not choose_backend("ATNet", cfg):                               # the reference trains against ATNet
        raise SystemExit("[ivos-w] train_agent.py drives the synthetic session only: training against the real ATNet clone needs the "
                         "reference's run_VOS_singleiact glue (utils/utils_atnet.py, out of scope - SURVEY section 2); evaluate a trained "
                         "agent on the real stack with eval_agent_*.py synthetic=0, or run with synthetic=1.")
    cfg.data.subset = cfg.data.get("subset", "train")
    misc.set_random_seed(2019)
    davis = SyntheticDavis(cfg, device)
    vos = StandInVOS(device, seed=int(cfg.seed))
    shared_dir = cfg.agent.save_result_dir
    p_reward, p_pre = os.path.join(shared_dir, cfg.agent.reward_csv), os.path.join(shared_dir, cfg.agent.pretrain_csv)
    if rank == 0 and not (os.path.exists(p_reward) and os.path.exists(p_pre)):
        n = bootstrap_synthetic_pool(cfg, davis, vos, device)
        print(f"[ivos-w] bootstrapped {p_reward} / {p_pre} from random-policy episodes ({n} transitions)")
    if world > 1:
        torch.distributed.barrier()
        if rank > 0:
            import shutil
            cfg.agent.save_result_dir = os.path.join(shared_dir, f"rank{rank}")
            os.makedirs(cfg.agent.save_result_dir, exist_ok=True)
            for src in (p_reward, p_pre):
                shutil.copy(src, cfg.agent.save_result_dir)
            p_reward, p_pre = (os.path.join(cfg.agent.save_result_dir, os.path.basename(q)) for q in (p_reward, p_pre))
    save_dir = cfg.agent.save_result_dir
    misc.set_random_seed(2019)                                         # the bootstrap consumed rank 0's streams: re-align every rank
    agent = Agent(device=device, cfg=cfg)
    df = pd.read_csv(p_reward, index_col=0)
    agent.memory_pool.load_from_csv(p_pre, save_dir, cfg.agent.sample_th)
    print(f"init memory pool from {p_pre}, now the size of memory pool is: {len(agent.memory_pool)}")
    davis.sets[cfg.data.subset] = [s for s in agent.memory_pool.seq_list if s in davis.dataset] or list(davis.dataset)
    misc.set_random_seed(2019)
    cache = {}

    def loader_for(seen):
        if (seen - 1) % 3 == 0 or "ds" not in cache:
            cache["ds"] = load_agent_dataset(cfg, agent.memory_pool.seq_list)
        if world > 1:
            seed = int(torch.empty((), dtype=torch.int64).random_().item() & 0x7fffffff)      # the global stream is the same on every rank
            return DataLoader(cache["ds"], num_workers=0,
                              batch_sampler=parallel.RankBatchSampler(len(cache["ds"]), int(cfg.agent.train_batch_size), rank, world, seed))
        return DataLoader(cache["ds"], batch_size=int(cfg.agent.train_batch_size), shuffle=True, num_workers=0)
    history, seen_seq = [], {}
    for epoch in range(1, int(cfg.num_epochs) + 1):
        agent.set_train()
        sess = SyntheticSession(davis, cfg.data.subset, cfg.davis_interactive.metric, cfg.davis_interactive.max_nb_interactions,
                                save_dir, seed=epoch, rounds=3)
        out = _episode(cfg, davis, vos, sess, agent, device, df, loader_for, "ours", seen_seq)
        if rank == 0:
            misc.save_agent_checkpoint(agent.policy_net, ckpt_dir=cfg.ckpt_dir)
        gs = sess.get_global_summary()
        curve = gs["curve"][cfg.davis_interactive.metric][:-1]
        auc = float(np.trapz(curve) / (len(curve) - 1))
        history.append(dict(epoch=epoch, auc=auc, final=float(np.mean(out["finals"])), agent_loss=float(np.mean(out["losses"])) if out["losses"] else 0.0,
                            reward_done=float(np.mean(out["rewards_done"])), updates=agent.optimizer.state["step"]))
        print(f"# epoch {epoch}: auc:{auc:.4f} final {cfg.davis_interactive.metric}: {history[-1]['final'] * 100:.2f} agent loss: "
              f"{history[-1]['agent_loss']:.4f} reward_done: {history[-1]['reward_done']:.3f} updates: {history[-1]['updates']}")
    if world > 1:
        # replicas must be bit-identical: compare an exact integer checksum of the parameter bits on every rank
        bits = agent.policy_net.flat.detach().view(torch.int32).to(torch.int64)
        chk = torch.stack([bits.sum(), (bits * (torch.arange(bits.numel(), device=bits.device) % 1009 + 1)).sum()])
        if torch.distributed.get_backend() == "gloo":
            chk = chk.cpu()
        got = [torch.empty_like(chk) for _ in range(world)]
        torch.distributed.all_gather(got, chk)
        same = all(bool(torch.equal(g, got[0])) for g in got)
        for h in history:
            h.update(world=world, replicas_identical=same, collective=parallel.collective_path(agent.policy_net.flat_grad))
        for v in list(parallel._P2P.values()):
            if v is not None:
                v.close()
        parallel._P2P.clear()
        if not same:
            raise SystemExit(f"[ivos-w] rank {rank}: the data-parallel replicas diverged (parameter checksums differ)")
    if rank == 0:
        with open(os.path.join(save_dir, "train_summary.json"), "w") as fp:
            json.dump(history, fp)
    return history


def main_eval(backbone, argv=None):
    cfg = parse_cli(sys.argv[1:] if argv is None else argv)
    return run_eval(cfg, backbone)


def main_train(argv=None):
    cfg = parse_cli(sys.argv[1:] if argv is None else argv, phase="train")
    return run_train(cfg)
