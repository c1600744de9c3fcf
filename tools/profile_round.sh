# Full measurement pass for profiles/ (run on the GPU box): bench line, kernel stats of the same command, PMC passes
# of the bench workload and of configs 4 and 5 (summaries: python3 tools/pmc_summary.py <tag>_c3|_c4|_c5 --merge).
# usage: bash tools/profile_round.sh <tag>
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $R/gpurun_out/${TAG}_stats.log 2>&1 || exit 2
cd $R && bash tools/pmc_passes2.sh ${TAG}_c3 || exit 3
cd $R && PMC_STEPS="--steps 5 --warmup 1 --frames-per-step 2" bash tools/pmc_passes2.sh ${TAG}_c4 --config 4 || exit 4
cd $R && PMC_STEPS="--steps 2 --warmup 1 --frames-per-step 1" PMC_TIMEOUT=240 bash tools/pmc_passes2.sh ${TAG}_c5 --config 5 || exit 5
cd $R && for c in c3 c4 c5; do python3 tools/pmc_summary.py ${TAG}_$c --merge > gpurun_out/${TAG}_${c}_summary.txt || exit 6; done
cp profiles/pmc_traffic.json gpurun_out/${TAG}_pmc_traffic.json
timeout -k 10 400 python -u $R/bench.py --steps 20 --warmup 5 > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err || exit 1
