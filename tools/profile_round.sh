# Full measurement pass for profiles/: bench line, kernel stats, PMC passes, FETCH_SIZE calibration.
# usage (on the GPU box): bash tools/profile_round.sh <tag>
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout -k 10 400 python -u $R/bench.py --steps 20 --warmup 3 --cpu-stride 40 > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/${TAG}_stats.log 2>&1 || exit 2
cd $R && bash tools/pmc_passes2.sh $TAG || exit 3
cd /tmp && timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_calib -- $R/tools/calib_fetch > $R/gpurun_out/${TAG}_calib.log 2>&1; echo calib rc=$?
