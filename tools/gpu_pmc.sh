#!/bin/bash
# PMC passes for the per-step kernels (separate runs per counter group; no tracing domains mixed in)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
ARGS="--steps 1 --warmup 0 --no_cpu_baseline ${BENCH_ARGS}"
: > $REPO/gpurun_out/pmc.log
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_VMEM_RD SQ_INSTS_LDS"; do
  i=$((i+1))
  rm -rf /tmp/pmc$i
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc$i -o run -- python $REPO/bench.py $ARGS > /tmp/pmc$i.log 2>&1
  echo "== group $i: $grp (rc=$?)" >> $REPO/gpurun_out/pmc.log
  python $REPO/tools/pmc_summary.py /tmp/pmc$i >> $REPO/gpurun_out/pmc.log 2>&1
done
cat $REPO/gpurun_out/pmc.log
