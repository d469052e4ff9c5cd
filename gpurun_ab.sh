timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "hsv or chain or mog2_mask or grey" 2>&1 | tail -2
for r in 1 2; do for w in 1080p1 1080p16 4k1; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline --no-parity 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$w', round(j['value'],1), {k:round(x,4) for k,x in j['stage_ms'].items()})"
done; done
