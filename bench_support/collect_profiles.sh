#!/bin/bash
# copy what bench_support/round_artifacts.sh <tag> left under gpurun_out/ into profiles/ (tracked): bash bench_support/collect_profiles.sh r2
TAG=${1:-r2}
P=profiles
for c in cfg2 cfg3 cfg4 cfg5; do
  d=gpurun_out/prof_${TAG}_${c}_final
  [ -f $d/summary.txt ] && cp $d/summary.txt $P/${TAG}_${c}_summary.txt
  f=$(find $d/trace -name "*kernel_stats.csv" 2>/dev/null | head -1)
  [ -n "$f" ] && cp $f $P/${TAG}_${c}_kernel_stats.csv
done
cp gpurun_out/${TAG}_final/bench_lines.jsonl $P/${TAG}_bench_lines.jsonl
for f in gpurun_out/${TAG}_final/*.txt; do grep -v "amdgpu.ids" $f > $P/${TAG}_$(basename $f); done
python3 - "$TAG" <<'PY'
import json, sys
tag = sys.argv[1]
db = json.load(open('profiles/traffic.json'))
db.update(json.load(open('gpurun_out/prof_%s_cfg2_final/traffic_record.json' % tag)))
json.dump(db, open('profiles/traffic.json', 'w'), indent=1, sort_keys=True)
print({k: v.get('lib_sha256') for k, v in db.items()})
PY
