python -m pytest tests/test_gpu_assess.py -q 2>&1 | tail -12
python bench.py --steps 20 --warmup 3 --no-clock-probe --no-live-traffic --no-cpu-baseline --no-fp32 --layer-report gpurun_out/r06_layers_g8.txt 2>&1 | tail -1 | cut -c1-200
IVOSW_TUNE_G8=0 python bench.py --steps 20 --warmup 3 --no-clock-probe --no-live-traffic --no-cpu-baseline --no-fp32 --layer-report gpurun_out/r06_layers_g8off.txt 2>&1 | tail -1 | cut -c1-200
