cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_dqn
rocprofv3 --kernel-trace -d gpurun_out/prof_dqn -o t -- python bench.py --workload dqn --steps 60 --warmup 10 --dqn-mode ${DQN_MODE:-plain} --no-cpu-baseline > /dev/null 2>&1
ls gpurun_out/prof_dqn
