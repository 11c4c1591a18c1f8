import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ivos_w_amd import synth
from oracle import assess_oracle as ao
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
sd = ao.to_torch_sd(synth.assessnet_state_dict(0))
tf, tp = synth.assess_inputs(8, seed=1234)
for th in (8, 16, 32, 64):
    torch.set_num_threads(th)
    ao.assess_forward(sd, tf, tp)
    t0 = time.perf_counter(); ao.assess_forward(sd, tf, tp); ao.assess_forward(sd, tf, tp); dt = (time.perf_counter() - t0) / 2
    print("threads", th, "s/8frames", round(dt, 3), "fps", round(8 / dt, 2))
