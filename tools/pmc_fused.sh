#!/bin/bash
# Dev tool: SQ counters of fused_stream_kernel, one rocprofv3 pass per group (gpurun: --pmc only with --kernel-trace).
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_fused
rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM" \
           "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_WAVES" \
           "TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_TA_TCP_STATE_READ TCP_GATE_EN1" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/tools/kbench.py fused > $OUT/p$i.log 2>&1 || echo "pass $i ($grp) failed: $(tail -2 $OUT/p$i.log)"
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT fused_stream
