#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root:  bash tools/gpu_profile.sh <tag> [bench args...]
# Collects, under gpurun_out/prof_<tag>/:  kernel-trace stats, then PMC passes (each in its own run,
# --kernel-trace only, as the node policy requires), and a text summary of all of them.
set -u
TAG=${1:-run}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
export TMPDIR=/tmp
cd $R
# The per-kernel evidence (kernel-trace statistics, PMC traffic) is taken from the ONE-CHAIN form of the benchmark: one
# whole-batch launch per step, kernels strictly one after the other, so a kernel's duration is a launch time.  The
# default two-chain form (half-batch kernels of two streams in flight together) gets its own kernel trace below.
BENCH="python bench.py --chains 1 --steps 200 --warmup 20 --no-extras $*"
rocprofv3 --kernel-trace --stats --output-format rocpd csv -d $O/stats -o r -- $BENCH > $O/stats.log 2>&1
cp $O/stats/r_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d $O/pmc_sq1 -o r -- $BENCH > $O/pmc_sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d $O/pmc_sq2 -o r -- $BENCH > $O/pmc_sq2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o r -- $BENCH > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o r -- $BENCH > $O/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $O/pmc_tcc -o r -- $BENCH > $O/pmc_tcc.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_DRAM_32B -d $O/pmc_rd32 -o r -- $BENCH > $O/pmc_rd32.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_WRITE_DRAM_32B -d $O/pmc_wr32 -o r -- $BENCH > $O/pmc_wr32.log 2>&1
python tools/rocpd_summary.py $O/stats/r_results.db $O/pmc_sq1/r_results.db $O/pmc_sq2/r_results.db $O/pmc_fetch/r_results.db $O/pmc_write/r_results.db $O/pmc_tcc/r_results.db $O/pmc_rd32/r_results.db $O/pmc_wr32/r_results.db > $O/summary.txt 2>&1
python tools/rocpd_summary.py --traffic-json $O/traffic.json --profile-tag $TAG --csrc-hash $(python -c "import bench; print(bench.csrc_hash())") $O/pmc_fetch/r_results.db $O/pmc_write/r_results.db $O/pmc_rd32/r_results.db $O/pmc_wr32/r_results.db > /dev/null 2>&1
# the default form: two chains.  Kernel trace + how many step kernels are in flight + the aggregate rate over the trains
BENCH2="python bench.py --chains 2 --steps 200 --warmup 20 --no-extras $*"
rocprofv3 --kernel-trace --stats --output-format rocpd csv -d $O/stats2 -o r -- $BENCH2 > $O/stats2.log 2>&1
cp $O/stats2/r_kernel_stats.csv $O/kernel_stats_two_chains.csv 2>/dev/null
python tools/rocpd_summary.py --concurrency $O/stats2/r_results.db > $O/summary_two_chains.txt 2>&1
grep -h '"metric"' $O/stats2.log > $O/bench_two_chains_under_rocprof.json
# the same with the extras (the observation-writing step kernel, fused rollouts, the 2^24 streaming run, ...):
# kernel-trace statistics of every kernel, and the HBM traffic counters of the step kernel that writes observations
# (one chain: counter collection serialises kernels ACROSS queues, so the ticket kernels of a two-chain rollout would wait
#  for each other until their bound runs out -- loudly, since round 5)
BENCHX="python bench.py --chains 1 --no-two-chain-extra --steps 200 --warmup 20 --cpu-seconds 1 $*"
rocprofv3 --kernel-trace --stats --output-format rocpd csv -d $O/statsx -o r -- $BENCHX > $O/statsx.log 2>&1
cp $O/statsx/r_kernel_stats.csv $O/kernel_stats_extras.csv 2>/dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmcx_fetch -o r -- $BENCHX > $O/pmcx_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmcx_write -o r -- $BENCHX > $O/pmcx_write.log 2>&1
python tools/rocpd_summary.py --kernel "step_kernel<1, true, true, true>" $O/statsx/r_results.db $O/pmcx_fetch/r_results.db $O/pmcx_write/r_results.db > $O/summary_obs_kernel.txt 2>&1
# ---- the step kernel PER BATCH SIZE (round 6): a kernel trace and the FETCH_SIZE / WRITE_SIZE passes of its own for the two
#      secondary sizes -- BASELINE configs[1] (65 536 boards: the launch train replayed from the cached hipGraph, the kernel is
#      step_graph_kernel) and 2^24 boards (256 MiB of records: every launch streams HBM) -- so that the 0.126 / 0.7375 figures
#      of bench.py's extras can be recomputed from profiles/ alone.  The 2^20 set is the main one above.
size_profile () {   # $1 = label, $2 = kernel-name filter, rest = bench args
  local L=$1 K=$2; shift 2
  local B="python bench.py --chains 1 --no-extras $*"
  rocprofv3 --kernel-trace --stats --output-format rocpd csv -d $O/stats_$L -o r -- $B > $O/stats_$L.log 2>&1
  cp $O/stats_$L/r_kernel_stats.csv $O/kernel_stats_$L.csv 2>/dev/null
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch_$L -o r -- $B > $O/pmc_fetch_$L.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write_$L -o r -- $B > $O/pmc_write_$L.log 2>&1
  python tools/rocpd_summary.py --kernel "$K" $O/stats_$L/r_results.db $O/pmc_fetch_$L/r_results.db $O/pmc_write_$L/r_results.db > $O/summary_$L.txt 2>&1
  grep -h '"metric"' $O/stats_$L.log > $O/bench_${L}_under_rocprof.json
}
size_profile 2p16 "step_graph_kernel<1, true>" --boards 65536 --steps 200 --warmup 20
size_profile 2p24 "step_kernel<1, true, true, false>" --boards 16777216 --steps 24 --warmup 8 --device-warmup 0
if [ "${G2048_PROFILE_POLICY:-0}" = "1" ]; then
  # BASELINE configs[4]: MIOpen's kernel selection happens in a FIRST process (its user find-db keeps the result); the SECOND
  # process runs under the kernel trace -- a naive_conv* kernel in it would be MIOpen's chosen solution, not a find
  python bench_policy.py --steps 4 --warmup 1 > $O/policy_first_process.json 2> $O/policy_first_process.log
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_policy -o r -- python bench_policy.py --steps 10 --warmup 2 > $O/policy_under_rocprof.json 2> $O/stats_policy.log
  cp $O/stats_policy/r_kernel_stats.csv $O/kernel_stats_policy.csv 2>/dev/null
  echo "naive_conv kernels in the profiled process: $(grep -c naive_conv $O/kernel_stats_policy.csv)" > $O/policy_naive_conv_check.txt
fi
grep -h '"metric"' $O/statsx.log > $O/bench_extras_under_rocprof.json
grep -h '"metric"' $O/stats.log > $O/bench_under_rocprof.json
rm -rf $O/*/r_results.db   # keep the merged gpurun_out small; the summary is what gets committed
cat $O/summary.txt
