bash tools/profile_round.sh r06a > gpurun_out/profile_round_r06a.log 2>&1
tail -30 gpurun_out/profile_round_r06a.log
