#!/bin/bash
# Collects the judged profiles of one round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r03   ->  gpurun_out/r03_bench_c3_kernel_stats.csv, r03_hbm_traffic.json, r03_pmc_sq_counters.json
# Kernel trace and every counter group are separate rocprofv3 runs (never --pmc together with a trace domain).
set -e
R=${1:-rXX}
export TMPDIR=/tmp
O=$PWD/gpurun_out; W=/tmp/prof_$R; mkdir -p $O $W
B="python3 bench.py --no-cpu-baseline --no-sweep"
rocprofv3 --kernel-trace -d $W/kt -o kt -- $B --steps 5 --warmup 2 > $O/${R}_kt.log 2>&1
python3 tools/rocpd_stats.py $(find $W/kt -name '*results.db' | head -1) $O/${R}_bench_c3_kernel_stats.csv > $O/${R}_kt_stats.txt
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE -d $W/fetch -o f --output-format csv -- $B --steps 2 --warmup 1 > $O/${R}_fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE -d $W/write -o w --output-format csv -- $B --steps 2 --warmup 1 > $O/${R}_write.log 2>&1
python3 tools/hbm_traffic.py $O/${R}_hbm_traffic.json $W/fetch $W/write 128 > $O/${R}_hbm.txt
echo "hbm done"
i=0
for G in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
         "SQ_INSTS_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_DATA_FIFO_FULL SQ_ACTIVE_INST_VMEM"; do
  rocprofv3 --pmc $G -d $W/sq$i -o s --output-format csv -- $B --steps 1 --warmup 1 > $O/${R}_sq$i.log 2>&1
  echo "sq group $i done"; i=$((i+1))
done
python3 tools/pmc_summary.py $O/${R}_pmc_sq_counters.json $W/sq0 $W/sq1 $W/sq2 $W/sq3 > $O/${R}_sq.txt
echo "all done"
