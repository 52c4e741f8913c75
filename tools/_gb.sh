python tools/_gemm_bench.py 2>&1 | tail -1
DVQ_GEMM_NODMA=1 python tools/_gemm_bench.py 2>&1 | tail -1
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
