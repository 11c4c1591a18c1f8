#!/usr/bin/env python
"""train_agent.py — Double-DQN training of the frame-recommendation agent with the MI355X hot path.

    python train_agent.py with num_epochs=5 agent.save_result_dir=train [synthetic=1] [key=value ...]

Same `with key=value` CLI and the same loop as the reference's train_agent.py (oracle state, `ours` policy with the epsilon
schedule, agent_business -> update_agent at the end of every episode, agent.pt per epoch).  The replay pool comes from
<agent.save_result_dir>/pretrain.csv and the reward baseline from reward.csv; on the synthetic back end both are produced first
by random-policy episodes when they do not exist (see ivos-w_amd/entry.py).
"""
import ivos_w_amd  # noqa: F401
from ivos_w_amd import entry

if __name__ == "__main__":
    entry.main_train()
