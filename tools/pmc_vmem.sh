#!/bin/bash
# Vector-memory path counters of the step kernel (TA / TCP / UTCL1), own rocprofv3 pass per group.  $1 = env copies.
N=${1:-4096}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_vmem_$N
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-double-buffer --steps 40 --warmup 260 --envs-per-gpu $N > $OUT/$name.log 2>&1 || echo "pass $name failed"
}
run v1 TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum
run v2 TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum
# (a pass with TA_* counters hung rocprofv3 on this pool: left out)
run v4 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum
run v5 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE
echo done
