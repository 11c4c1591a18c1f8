for d in 0 1 2 3 4 7 8 16 24 31; do
IVOSW_TUNE_BDBG=$d timeout 200 python bench.py --layer-report gpurun_out/abl_$d.txt --steps 3 --warmup 1 >/dev/null 2>&1
echo "BDBG=$d $(grep -E '^ +256 +64 +256 +256 +0' gpurun_out/abl_$d.txt)"
done
