for a in 0 6 4 5; do DVQ_GEMM_ABL=$a timeout 100 python tools/_gemm_clk.py 2>&1 | grep "blocks:" | tail -1; done
