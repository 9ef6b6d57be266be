#!/bin/bash
# Instruction-cache counters of the step kernel (own rocprofv3 passes; kernel-trace only).
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_ic
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
run() {
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-double-buffer --steps 40 --warmup 260 > $OUT/$name.log 2>&1 || echo "pass $name failed"
}
run ic1 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
run ic2 SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES
run ic3 SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_INST_REQ
echo done
