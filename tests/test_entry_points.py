"""Entry scripts (train_agent.py / eval_agent_*.py at the repo root -> ivos_w_amd.entry): the CLI, the explicit refusal
when the real evaluation stack is missing, the synthetic data / stand-in VOS (CPU), and — on the GPU — the whole loops:
session -> segmentation epilogue -> J&F -> recommend_frame -> [agent_business -> update_agent] -> summary / checkpoint."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from ivos_w_amd import entry

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cli_keeps_sacreds_with_syntax():
    c = entry.parse_cli(["with", "setting=wild", "method=worst", "dataset=davis", "agent.lr=1e-5", "synth.n_frames=12", "gpu_id=0"])
    assert (c.setting, c.method, c.dataset, c.agent.lr, c.synth.n_frames) == ("wild", "worst", "davis", 1e-5, 12)
    assert c.agent.memory_size == 100000 and c.davis_interactive.max_nb_interactions == 5 and c.agent.update_rate == 0.05
    with pytest.raises(SystemExit):
        entry.parse_cli(["with", "nosuch.section=1"])
    with pytest.raises(SystemExit):
        entry.parse_cli(["bogus"])


def test_missing_real_stack_is_reported_not_faked(capsys):
    c = entry.parse_cli(["with", "synthetic=0"])
    with pytest.raises(SystemExit) as e:
        entry.choose_backend("MANet", c)
    assert "davisinteractive" in str(e.value) and "DAVIS frames" in str(e.value)
    assert entry.choose_backend("MANet", entry.parse_cli([])) is True           # auto: falls back, and says so
    assert "SYNTHETIC" in capsys.readouterr().out


def test_every_reference_entry_script_exists_and_is_thin():
    for name in ("train_agent.py", "eval_agent_manet.py", "eval_agent_atnet.py", "eval_agent_ipn.py"):
        text = open(os.path.join(ROOT, name)).read()
        assert "entry.main_" in text and len(text.splitlines()) < 30


def test_synthetic_video_and_stand_in_vos_on_cpu():
    c = entry.parse_cli(["with", "synth.n_frames=10", "synth.height=48", "synth.width=64"])
    dv = entry.SyntheticDavis(c, torch.device("cpu"))
    seq = list(dv.dataset)[0]
    F, gt = dv.load_frames(seq), dv.load_annotations(seq)
    O = dv.dataset[seq]["num_objects"]
    assert F.shape == (10, 3, 48, 64) and gt.shape == (10, 48, 64) and gt.dtype == torch.uint8
    assert 0 <= float(F.min()) and float(F.max()) <= 1 and int(gt.max()) == O
    vos = entry.StandInVOS(torch.device("cpu"))
    lg = vos.logits(gt, O, [2])
    assert lg.shape == (10, O + 1, 12, 16)
    up = torch.nn.functional.interpolate(lg, (48, 64), mode="bilinear", align_corners=True).argmax(1)
    acc = (up == gt.long()).float().mean((1, 2))
    assert acc[2] > 0.9 and acc[2] >= acc[9]                                      # exact at the annotated frame, worse far from it
    assert torch.equal(lg, vos.logits(gt, O, [2]))                                # deterministic


@pytest.mark.gpu
def test_eval_and_train_loops_on_the_synthetic_back_end(tmp_path):
    common = ["synthetic=1", "synth.n_sequences=2", "synth.n_frames=26", "synth.height=120", "synth.width=216",
              f"ckpt_dir={tmp_path}/weights", f"report_save_dir={tmp_path}/results", f"agent.save_result_dir={tmp_path}/train"]
    env = dict(os.environ, PYTHONPATH=ROOT)
    # --- training: bootstrap of reward.csv / pretrain.csv, one epoch, checkpoint
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train_agent.py"), "with", "num_epochs=1", "agent.train_batch_size=16"] + common,
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    hist = json.load(open(tmp_path / "train" / "train_summary.json"))
    assert len(hist) == 1 and hist[0]["updates"] > 0 and np.isfinite(hist[0]["agent_loss"]) and 0 < hist[0]["final"] <= 1
    assert (tmp_path / "weights" / "agent.pt").exists() and (tmp_path / "train" / "memory_pool.csv").exists()
    sd = torch.load(tmp_path / "weights" / "agent.pt")
    assert list(sd)[0] == "encoder_fc1.weight" and len(sd) == 10
    # --- evaluation with the trained agent: wild/ours (AssessNet + Brain on the device), wild/worst, wild/random
    aucs = {}
    for method in ("ours", "worst", "random"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "eval_agent_manet.py"), "with", "setting=wild", f"method={method}", "dataset=davis",
                            "eval_max_nb_interactions=4", "precision=fp32"] + common, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        s = json.load(open(tmp_path / "results" / "MANet" / "wild" / "davis" / method / "summary.json"))
        assert set(s) == {"auc", "curve"} and len(s["curve"]["J_AND_F"]) == 4 and 0 < s["auc"] <= 1
        aucs[method] = s["auc"]
        if method == "ours":
            assert "frame-cache uploads 2" in r.stdout                              # one upload per sequence, not per interaction
    print(aucs)
