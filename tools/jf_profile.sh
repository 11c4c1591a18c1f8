cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python bench.py --steps 2 --warmup 1 --dqn-steps 5 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps(d['jf']))"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_jf -o t -- python bench.py --steps 1 --warmup 1 --dqn-steps 2 --no-cpu-baseline > /dev/null 2>&1
python tools/rocprof_summary.py gpurun_out/prof_jf/*.db 2>/dev/null | grep -i "jf_\|total kernel" | cut -c1-200
rm -rf gpurun_out/prof_jf
