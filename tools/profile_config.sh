#!/bin/bash
# Kernel-trace statistics of another BASELINE config (through gpurun): tools/profile_config.sh c5 -> gpurun_out/c5_kernel_stats.csv
C=${1:-c5}
export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/${C}kt -o kt -- python3 bench.py --config $C --no-cpu-baseline --steps 3 --warmup 2 > gpurun_out/${C}_kt.log 2>&1
python3 tools/rocpd_stats.py $(find /tmp/${C}kt -name '*results.db' | head -1) gpurun_out/${C}_kernel_stats.csv > gpurun_out/${C}_kt_stats.txt
