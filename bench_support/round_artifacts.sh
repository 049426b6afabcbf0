#!/bin/bash
# Everything a round's profiles/ directory is built from, in one GPU-box call (run through gpurun from the repo root):
#   bash bench_support/round_artifacts.sh r2        -> gpurun_out/<tag>_final/{bench_lines.jsonl, measures.txt, micro_*.txt, ...}
# plus the counter profile of the headline config (bench_support/profile.sh), whose traffic record carries the hash of the
# library it measured.  Copy what should be judged from gpurun_out/ into profiles/ afterwards.
set -u
TAG=${1:-r2}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/${TAG}_final
mkdir -p "$OUT"
M=bench_support/micro
LIBDIR=polars-strsim_amd/polars_strsim

# -- micro-benchmarks (built here: the executables are not tracked)
hipcc --offload-arch=gfx950 -O3 $M/op_cost.hip -o $M/op_cost 2>/dev/null && $M/op_cost > "$OUT/micro_op_cost.txt" 2>&1
hipcc --offload-arch=gfx950 -O3 $M/lds_window.hip -o $M/lds_window 2>/dev/null && $M/lds_window > "$OUT/micro_lds_window.txt" 2>&1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ipolars-strsim_amd/csrc $M/core_issue.hip -o $M/core_issue 2>/dev/null && $M/core_issue > "$OUT/micro_core_issue.txt" 2>&1
hipcc -O3 -std=c++17 $M/register_vs_copy.hip -o $M/register_vs_copy -lpthread 2>/dev/null && { $M/register_vs_copy 160; $M/register_vs_copy 1600; } > "$OUT/f1_register_vs_copy.txt" 2>&1
g++ -O2 -Iinclude $M/small_call_latency.cpp -o $M/small_call_latency $LIBDIR/libpolars_strsim_amd.so -Wl,-rpath,"$ROOT/$LIBDIR" 2>/dev/null && $M/small_call_latency > "$OUT/f3_small_call_latency.txt" 2>&1

# -- bench lines: the default line (headline, with cpu_baseline and the PCIe-inclusive extra field), then the other configs
: > "$OUT/bench_lines.jsonl"
python bench.py 2> "$OUT/bench_default.err" | tail -1 >> "$OUT/bench_lines.jsonl"
for cfg in cfg1 cfg3 cfg5; do
  python bench.py --config $cfg --no-cpu-baseline --no-e2e 2>> "$OUT/bench_default.err" | tail -1 >> "$OUT/bench_lines.jsonl"
done
python bench.py --config cfg4 --rows 100000000 --steps 20 --warmup 10 --no-cpu-baseline --no-e2e 2>> "$OUT/bench_default.err" | tail -1 >> "$OUT/bench_lines.jsonl"
bash bench_support/bench_measures.sh > "$OUT/measures.txt" 2>&1
python bench_support/bench_literal.py > "$OUT/literal.txt" 2>&1
python bench_support/bench_plugin_e2e.py > "$OUT/plugin_e2e.txt" 2>&1
python bench_support/bench_small_calls.py > "$OUT/small_calls_plugin.txt" 2>&1

# -- kernel trace + counters of the headline config (five rocprofv3 passes)
TRAFFIC_KEY=cfg2:levenshtein:100000000 TRAFFIC_KERNEL=k_lane_stage bash bench_support/profile.sh ${TAG}_cfg2_final > /dev/null 2>&1
for cfg in cfg3 cfg5; do
  ( export TMPDIR=/tmp; cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_${TAG}_${cfg}_final/trace" -- python3 "$ROOT/bench.py" --config $cfg --no-cpu-baseline --no-e2e > "$ROOT/gpurun_out/prof_${TAG}_${cfg}_final.log" 2>&1 )
  python3 bench_support/summarize_profile.py "gpurun_out/prof_${TAG}_${cfg}_final" > "gpurun_out/prof_${TAG}_${cfg}_final/summary.txt" 2>&1
done
( export TMPDIR=/tmp; cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_${TAG}_cfg4_final/trace" -- python3 "$ROOT/bench.py" --config cfg4 --rows 100000000 --steps 20 --warmup 10 --no-cpu-baseline --no-e2e > "$ROOT/gpurun_out/prof_${TAG}_cfg4_final.log" 2>&1 )
python3 bench_support/summarize_profile.py "gpurun_out/prof_${TAG}_cfg4_final" > "gpurun_out/prof_${TAG}_cfg4_final/summary.txt" 2>&1
cat "$OUT/bench_lines.jsonl" "$OUT/measures.txt"
