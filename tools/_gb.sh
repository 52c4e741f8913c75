for a in 0 3 4 5; do DVQ_GEMM_ABL=$a python tools/_gemm_bench.py 2>&1 | tail -1; done
