#!/bin/bash
# round 6: numpy-RNG mode as ONE launch per step (in-block compacted resets, recorded-j shuffle, deficit update)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06c; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu -k "numpy or round6 or strict or state or checkpoint or fuzz" > $O/pytest_numpy.log 2>&1; echo "rc $?" >> $O/pytest_numpy.log
tail -4 $O/pytest_numpy.log
timeout 300 python tools/numpy_mode_probe.py > $O/numpy_mode_probe.txt 2>&1; grep "numpy-RNG" $O/numpy_mode_probe.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o r -- python tools/numpy_mode_probe.py > $O/kt.log 2>&1
grep -h "numpy" $O/kt/*kernel_stats.csv | cut -c1-200
cp $O/kt/*kernel_stats.csv $O/numpy_mode_kernel_stats.csv; rm -rf $O/kt
