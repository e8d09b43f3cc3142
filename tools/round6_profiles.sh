#!/bin/bash
# The round-6 evidence set in one call on the GPU box (outputs under gpurun_out/r6final/, copied into profiles/ by hand):
#   bench_line.json                       python bench.py (the driver's command)
#   (the profiled runs leave out the scaling model and the one-device config 4, whose launches of the same kernels at other sizes
#   would mix into the per-kernel means)
#   bench_kernel_stats.csv + bench_line_same_run_as_kernel_stats.json + iteration_timeline.txt
#                                         rocprofv3 --kernel-trace --stats of a shorter bench run, the line printed inside it, the
#                                         kernels of one iteration in start order (tools/trace_timeline.py)
#   pmc_fetch_write.json                  FETCH_SIZE / WRITE_SIZE, one --pmc pass each (tools/pmc_fetch_write.py)
#   pmc_fused_clock.{txt,json}            SQ counters and effective shader clock of k_fused<8> / k_build_gram<8>
#   select_newton_phases.txt              MBAR_DEBUG_STAMPS=1: shader-clock phases of k_select_newton, LDL^T and Gauss-Jordan
#   tail_probe.txt                        what an iteration costs outside its sweep (config 3 and its 8-rank shard)
#   class_bench.txt, expectation_family.txt   per-call times of the class
# Usage: gpurun --timeout 2400 -- 'bash tools/round6_profiles.sh'
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r6final
mkdir -p $OUT
REPO=$(pwd)
export TMPDIR=/tmp
python bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err
BENCH="python $REPO/bench.py --steps 10 --warmup 3 --cpu-sample 0 --api-e2e 0 --scaling-model 0 --config4-one-device 0"
( cd /tmp && rm -rf /tmp/prof_s && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -o ks -- $BENCH > $REPO/$OUT/bench_line_same_run_as_kernel_stats.json 2> $REPO/$OUT/rocprof_stats.err )
find /tmp/prof_s -name "*kernel_stats.csv" -exec cp {} $OUT/bench_kernel_stats.csv \;
TRACE=$(find /tmp/prof_s -name "*kernel_trace.csv" | head -1)
[ -n "$TRACE" ] && python tools/trace_timeline.py "$TRACE" k_select > $OUT/iteration_timeline.txt 2>&1
SHORT="python $REPO/bench.py --steps 3 --warmup 1 --cpu-sample 0 --api-e2e 0 --scaling-model 0 --config4-one-device 0 --small-configs 0"
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rm -rf /tmp/prof_$C && rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/prof_$C -o pm -- $SHORT > /dev/null 2> $REPO/$OUT/rocprof_$C.err )
done
F=$(find /tmp/prof_FETCH_SIZE -name "*counter_collection.csv" | head -1)
W=$(find /tmp/prof_WRITE_SIZE -name "*counter_collection.csv" | head -1)
[ -n "$F" ] && [ -n "$W" ] && python tools/pmc_fetch_write.py "$F" "$W" 128 10000000 $OUT/pmc_fetch_write.json > $OUT/pmc_fetch_write.log 2>&1
( cd /tmp && rm -rf /tmp/prof_clk && rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d /tmp/prof_clk -o pc -- $SHORT > /dev/null 2> $REPO/$OUT/rocprof_clk.err )
CC=$(find /tmp/prof_clk -name "*counter_collection.csv" | head -1)
CT=$(find /tmp/prof_clk -name "*kernel_trace.csv" | head -1)
if [ -n "$CC" ] && [ -n "$CT" ]; then
  python tools/pmc_kernel_clock.py "$CC" "$CT" k_fused 100 --json $OUT/pmc_fused_clock.json > $OUT/pmc_fused_clock.txt 2>&1
  python tools/pmc_kernel_clock.py "$CC" "$CT" k_build_gram 100 >> $OUT/pmc_fused_clock.txt 2>&1
fi
{ MBAR_DEBUG_STAMPS=1 python tools/tail_probe.py "LDL^T (default)" 2>&1 | grep -v "launch \(1[0-9]\|[4-9]\|2[0-9]\) "; MBAR_NEWTON_LDLT=0 MBAR_DEBUG_STAMPS=1 python tools/tail_probe.py "register Gauss-Jordan (newton_ldlt = 0)" 2>&1 | grep -v "launch \(1[0-9]\|[4-9]\|2[0-9]\) "; } > $OUT/select_newton_phases.txt
{ python tools/tail_probe.py "LDL^T (default)"; MBAR_NEWTON_LDLT=0 python tools/tail_probe.py "register Gauss-Jordan"; } > $OUT/tail_probe.txt 2>&1
python tools/bench_class.py > $OUT/class_bench.txt 2>&1
python tools/profile_expectations.py all 7 > $OUT/expectation_family.txt 2>&1
ls -la $OUT
