bash tools/pmc_tower.sh > gpurun_out/r06_pmc_tower_survey.txt 2>&1
tail -14 gpurun_out/r06_pmc_tower_survey.txt
grep -A 18 "res2_chain_kernel<true, true>" gpurun_out/r06_pmc_tower_survey.txt | head -22
