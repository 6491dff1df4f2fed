#!/bin/bash
# r06z: kernel timelines of the two B = 128 paths (where do the ~150 / ~118 us go: kernels or gaps?)
export TMPDIR=/tmp
R=$(pwd); O=$R/gpurun_out/r06z; mkdir -p $O
cd /tmp && rocprofv3 --kernel-trace -f csv -d /tmp/c2 -- python $R/tools/config2_loop_trace.py 128 > $O/c2.txt 2>&1
f=$(ls /tmp/c2/*/*kernel_trace.csv | head -n 1); python $R/tools/kernel_window.py $f -60 40 | tee $O/config2_loop_b128_timeline.txt
DIB_SMALL_EPOCHS=20 rocprofv3 --kernel-trace -f csv -d /tmp/kd -- python $R/tools/small_batch_bench.py > $O/kd.txt 2>&1
f=$(ls /tmp/kd/*/*kernel_trace.csv | head -n 1); python $R/tools/kernel_window.py $f -80 60 | tee $O/keras_default_b128_timeline.txt
tail -n 2 $O/c2.txt $O/kd.txt
