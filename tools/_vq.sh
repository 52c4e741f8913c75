timeout 300 python -m pytest tests -m gpu -x -q -k "vq or quantizer" 2>&1 | tail -4
timeout 100 python bench.py --vq-only 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read())['roofline_vq_argmin']; print(d['us_per_call'], d['frac'], d['kernels_us'], d['bit_match_vs_exact_fp32_kernel'])"
