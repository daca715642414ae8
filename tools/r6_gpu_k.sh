#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06k; mkdir -p $O
timeout 1500 python tests/soak_parity.py 16 2500 0.01 numpy > $O/soak_numpy.txt 2>&1; tail -2 $O/soak_numpy.txt
timeout 1500 python tests/soak_parity.py 18 2000 0.01 > $O/soak.txt 2>&1; tail -2 $O/soak.txt
G2048_FUZZ_STREAMS=1 timeout 900 python tests/fuzz_parity.py 240 11 > $O/fuzz_streams.txt 2>&1; tail -1 $O/fuzz_streams.txt
timeout 900 python tests/fuzz_parity.py 240 12 > $O/fuzz.txt 2>&1; tail -1 $O/fuzz.txt
