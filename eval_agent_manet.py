#!/usr/bin/env python
"""eval_agent_manet.py — frame-recommendation evaluation loop with the MI355X hot path (AssessNet + Agent on HIP kernels).

    python eval_agent_manet.py with setting=wild method=ours dataset=davis [synthetic=1] [precision=fp32] [key=value ...]

Same `with key=value` CLI and the same loop as the reference's entry point of this name; the MANet backbone, the DAVIS-interactive
session and the scribble robot are external packages that are not part of this build: when they are missing the script says
so and runs the synthetic session with the stand-in VOS model (see ivos-w_amd/entry.py).  Writes
<report_save_dir>/MANet/<setting>/<dataset>/<method>/summary.json = {"auc", "curve"}.
"""
import ivos_w_amd  # noqa: F401  (registers the package under its importable name)
from ivos_w_amd import entry

if __name__ == "__main__":
    entry.main_eval("MANet")
