# PMC passes over bench.py; each pass is its own rocprofv3 run (counters only, no API tracing).
# usage (on the GPU box): bash tools/pmc_passes2.sh <tag> [bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-x}; shift
ARGS="$@"
run() { name=$1; shift; timeout -s KILL ${PMC_TIMEOUT:-150} rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_$name -- python3 $R/bench.py ${PMC_STEPS:---steps 20 --warmup 2 --frames-per-step 8} --no-cpu-baseline --no-extras $ARGS > $R/gpurun_out/pmc_${TAG}_$name.log 2>&1; rc=$?; echo "$name rc=$rc"; return $rc; }
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD && \
run sq2 SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS && \
run ta TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE GRBM_TA_BUSY && \
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum && \
run fetch FETCH_SIZE && \
run write WRITE_SIZE
