#!/bin/bash
# PMC counter passes for the camera kernel (config 5: 512 copies, both cameras): where does its time go?
# Each --pmc set is its own rocprofv3 run, kernel-trace only.  Summary: tools/pmc_summary.py <dir> mjrl_render_kernel
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_render
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/tools/render_rate.py 512 > $OUT/$name.log 2>&1 || echo "pass $name failed"
}
run sq1 SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVES
run fetch FETCH_SIZE
run write WRITE_SIZE
echo done
