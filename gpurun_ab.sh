timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
for r in 1 2; do for w in 1080p1 4k1 1080p16; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline --no-parity 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$w', round(j['value'],1), {k:round(x,4) for k,x in j['stage_ms'].items()}, j['positions_found'], round(j['roofline']['avg_launch_ms'],4))"
done; done
