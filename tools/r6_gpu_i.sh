#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r06i; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/strict_probe.py 2>/dev/null | tee $O/strict_probe.txt
# numpy-RNG mode: where the time goes (counters, each group in its own pass)
P="python tools/numpy_mode_probe.py"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d $O/pmc1 -o r -- $P > $O/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d $O/pmc2 -o r -- $P > $O/pmc2.log 2>&1
python tools/rocpd_summary.py --kernel "step_numpy_kernel" $O/pmc1/r_results.db $O/pmc2/r_results.db > $O/numpy_pmc_summary.txt 2>&1
rm -rf $O/pmc1 $O/pmc2; cat $O/numpy_pmc_summary.txt | tail -25
