# usage: tools/sweep.sh "<variant>:<tune list>" ... ; prints value / ms for each setting
for vt in "$@"; do
  v=${vt%%:*}; t=${vt#*:}
  timeout -k 10 120 python -u bench.py --steps 10 --warmup 2 --no-cpu-baseline --variant $v --tune "$t" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; k=d['kernels']; print('variant $v tune $t', round(d['value']), 'Mrays/s', round(d['ms_per_step'],3), 'ms | closest', round(k['closest_ms'],3), 'shadow', round(k['shadow_ms'],3), 'frame_dev', round(d['device_frame_ms_rank0'],3), 'frac', round(r['frac'],3))"
done
