cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_dqn
rocprofv3 --kernel-trace -d gpurun_out/prof_dqn -o t -- python bench.py --workload dqn --steps 20 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
ls gpurun_out/prof_dqn
