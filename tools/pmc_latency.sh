#!/bin/bash
# Average memory latencies of the step kernel from the SQ level counters (level / instructions), own passes each.
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_lat
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-double-buffer --steps 40 --warmup 260 > $OUT/$name.log 2>&1 || echo "pass $name failed"
}
run l1 SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES
run l2 SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
run l3 SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INSTS_VALU
run l4 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum
run l5 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_BUSY_CYCLES
echo done
