python -m pytest tests -m gpu -q -x -k "pointnet" 2>&1 | tail -3
for cfg in "0 0" "1 0" "1 2048" "1 4096" "1 1024"; do set -- $cfg
  echo "== STREAMS=$1 CHUNK=$2"; DVQ_PN_STREAMS=$1 DVQ_PN_CHUNK=$2 PN_B=16384 PN_REP=3 python3 tools/pn_quick.py 2>&1 | grep "^C=" | cut -c1-170
done
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-latency > gpurun_out/r05_d_bench.json 2>/dev/null; python3 -c "
import json; d=json.loads(open('gpurun_out/r05_d_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k:v['ms'] for k,v in list(d['kernels'].items())[:7]}, d['gathered_sha256'][:16])"
DVQ_PN_STREAMS=0 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-latency --no-prof 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('one stream:', d['value'], d['ms_per_step'], d['gathered_sha256'][:16])"
