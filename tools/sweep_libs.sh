# usage: tools/sweep_libs.sh "<library file in rust-raytracer_amd/>:<tune list>[:extra bench args]" ...
# Runs bench.py against alternative builds of the library (RTX_LIB) and prints one line per setting.
set -e
for spec in "$@"; do
  lib=${spec%%:*}; rest=${spec#*:}; t=${rest%%:*}; extra=""
  if [ "$rest" != "$t" ]; then extra=${rest#*:}; fi
  RTX_LIB=$PWD/rust-raytracer_amd/$lib timeout -k 10 120 python -u bench.py --steps 10 --warmup 2 --no-cpu-baseline --tune "$t" $extra | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; k=d['kernels']; print('$lib tune $t $extra', round(d['value']), 'Mrays/s', round(d['ms_per_step'],3), 'ms | closest', round(k['closest_ms'],3), 'shadow', round(k['shadow_ms'],3), 'frame_dev', round(d['device_frame_ms_rank0'],3))"
done
