#!/bin/bash
# GPU box only.  Four rocprofv3 passes over the same bench command (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not
# fit one pass; counters are collected without any hip/hsa/memory trace domains):
#   $1       kernel trace + stats            ->  $1/kt_kernel_stats.csv
#   $1f/$1w  PMC FETCH_SIZE / WRITE_SIZE      (FETCH_SIZE is doubled by tools/summarize_profiles.py: gfx950 counts 1/2)
#   $1s      PMC SQ + GRBM counters           (MFMA-busy, wait breakdown, LDS bank conflicts, sustained clock)
# usage: bash tools/collect_profiles.sh gpurun_out/r02x [bench args...]; then (CPU side)
#        python tools/summarize_profiles.py gpurun_out/r02x r02x
# CONFIG5=1: the passes wrap `bench.py --config5-only` (the BASELINE config-5 set-transformer step, 4 x 4096 particles);
#        summarise with  python tools/summarize_profiles.py gpurun_out/r03x r03x_config5 @config5
B=$1; shift
ARGS="--steps 10 --warmup 2 --blocks 1 --no-cpu-baseline --no-extra --no-kernel-timing $*"
if [ -n "$CONFIG5" ]; then ARGS="--config5-only --steps 3 $*"; fi
R=$(pwd)
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$B -o kt -- python $R/bench.py $ARGS > $R/$B.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/${B}f -o f -- python $R/bench.py $ARGS >> $R/$B.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/${B}w -o w -- python $R/bench.py $ARGS >> $R/$B.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/${B}s -o s -- python $R/bench.py $ARGS >> $R/$B.log 2>&1
cd $R
# flatten rocprofv3's per-host subdirectory
for d in $B ${B}f ${B}w ${B}s; do find $d -mindepth 2 -type f -exec mv {} $d/ \; 2>/dev/null; done
ls $B ${B}f ${B}w ${B}s
