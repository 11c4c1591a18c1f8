#!/bin/bash
# data-parallel train_agent.py with two ranks on one GPU (gloo), full log in gpurun_out/dp_train.log
W=/tmp/dpw; rm -rf $W; mkdir -p $W gpurun_out
IVOSW_LOCAL_DEVICE=0 IVOSW_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 train_agent.py with synthetic=1 ckpt_dir=$W/ckpt agent.save_result_dir=$W/results num_epochs=1 synth.n_sequences=2 synth.n_frames=26 synth.height=120 synth.width=216 agent.train_batch_size=16 agent.update_rate=0.3 > gpurun_out/dp_train.log 2>&1
echo "rc=$?"; grep -v "Warning\|warn" gpurun_out/dp_train.log | grep -B12 "Error\|Exit\|assert" | head -60
cat $W/results/train_summary.json 2>/dev/null
