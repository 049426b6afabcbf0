#!/bin/bash
# copy what bench_support/jobs/r6_final.sh left under gpurun_out/ into profiles/ (tracked): bash bench_support/jobs/r6_collect.sh
P=profiles
for c in cfg2 cfg3 cfg5; do
  d=gpurun_out/prof_r6_${c}
  [ -f $d/summary.txt ] && grep -v "amdgpu.ids" $d/summary.txt > $P/r6_${c}_summary.txt
  f=$(find $d/trace -name "*kernel_stats.csv" 2>/dev/null | head -1)
  [ -n "$f" ] && cp $f $P/r6_${c}_kernel_stats.csv
done
F=gpurun_out/r6_final
cp $F/bench_lines.jsonl $P/r6_bench_lines.jsonl
cp $F/bench_2rank_one_gpu.jsonl $P/r6_bench_2rank_one_gpu.jsonl
cp $F/traffic.json $P/traffic.json
for f in small_frames literal mid_ascii plugin_e2e n8_first_contact_on_one_gpu; do grep -av "amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl path" $F/$f.txt > $P/r6_$f.txt; done
grep -av "RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl path" $F/pytest_gpu.txt > $P/r6_pytest_gpu.txt
python3 - <<'PY'
import json
out = open("profiles/r6_plugin_staging.txt", "w")
out.write("# tests/helpers/staging_child.py 32 4000000: 32 threads x one 4 M-row call each through the plugin ABI (engine-parallel mode), then idle, then a lone call\n")
for name, label in (("staging_32_threads_default_budget", "POLARS_STRSIM_STAGING_BUDGET_MB unset (4096)"), ("staging_32_threads_1gib", "POLARS_STRSIM_STAGING_BUDGET_MB=1024")):
    d = json.load(open("gpurun_out/r6_final/%s.json" % name))
    out.write("\n== %s\n" % label)
    out.write("   rows differing from the oracle: %s (threads), %d (lone call)\n" % (d["bad"], d["lone_bad"]))
    out.write("   wall %.3f s for %d x %d rows; a call took %.3f .. %.3f s\n" % (d["wall_s"], d["threads"], d["rows"], d["fastest_call_s"], d["slowest_call_s"]))
    for k in ("idle", "after_lone"):
        s = d[k]
        out.write("   %-10s live pinned %.1f MB, live device %.1f MB (budget %.0f MB); sets %d, in use %d, released %d, calls that waited %d, peak live at a return %.1f MB\n" % (
            k, s["live_pinned"] / 1e6, s["live_device"] / 1e6, s["budget"] / 1e6, s["sets"], s["sets_in_use"], s["sets_released"], s["calls_waited"], s["peak_live_at_a_return"] / 1e6))
    out.write("   peak live staging sampled every 2 ms while the calls ran: %.1f MB\n" % (d["peak_live_sampled"] / 1e6))
PY
ls profiles/n8_first_contact_*.jsonl 2>/dev/null
