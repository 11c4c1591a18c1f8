python -m pytest tests/test_gpu_assess.py -x -q -k "res2_chain or bf16_scores_vs" 2>&1 | tail -2
python tools/res2_chain_ab.py 256 3 2>&1 | tail -13
