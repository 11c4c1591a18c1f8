# per-kernel GRBM_GUI_ACTIVE (GPU cycles while the kernel ran) + SQ busy / MFMA busy cycles of the DQN step's kernels, next to their
# durations from a separate --kernel-trace pass: cycles / duration = the shader clock the step's short kernels actually get
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_dqn; mkdir -p gpurun_out/pmc_dqn
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d gpurun_out/pmc_dqn -o c --output-format csv -- python bench.py --workload dqn --dqn-steps 40 --steps 20 --warmup 10 --no-cpu-baseline --dqn-eager > gpurun_out/pmc_dqn/log.txt 2>&1
python tools/pmc_sq_summary.py gpurun_out/pmc_dqn/c_counter_collection.csv > gpurun_out/pmc_dqn_summary.txt 2>&1
tail -40 gpurun_out/pmc_dqn_summary.txt
