#!/bin/bash
# FIRST CONTACT with more than one GPU: everything north_star asks about N > 1 -- "Utf8 chunks scattered across
# the 8 GPUs of one node with a final RCCL gather of the f64 result column over xGMI ... reported at 1, 2, 4 and 8 GPUs" -- in ONE
# run on an 8-GPU node, appended to profiles/ so that the first box that has the GPUs leaves a complete record:
#
#   bash bench_support/jobs/n8_first_contact.sh [max gpus, default: all]      (from the repository root; ~10 min on 8 GPUs)
#
#   1. the C-ABI gather's own N-rank test over REAL RCCL (skipped on a one-GPU box; tests/test_gather_abi.py)
#   2. cfg2 (the headline) at N = 1, 2, 4, 8 x {abi: raw f64 through strsim_gather_f64_ranges | torch: coded through
#      torch.distributed.gather} x --root-share {1.0 (the reference's partition) | 0.4 (the named deviation)}
#   3. cfg4 (five measures of one 200 M-row frame) at the largest N, both gathers
# bench.py starts its own ranks (one process per GPU, torch.distributed.run as a child).  Every line carries config.distributed
# (world, backend, every rank's device, which gather ran, RCCL's own rank count) and gather_verified.
set -u
cd "$(dirname "$0")/../.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
HAVE=$(python - <<'PY'
import torch
print(torch.cuda.device_count())
PY
)
MAX=${1:-$HAVE}
[ "$MAX" -gt "$HAVE" ] && MAX=$HAVE
STAMP=$(date -u +%Y%m%dT%H%M%SZ)
OUT=profiles/n8_first_contact_${STAMP}.jsonl
LOG=profiles/n8_first_contact_${STAMP}.txt
echo "# n8_first_contact.sh at $STAMP: $HAVE GPU(s) visible, running up to N = $MAX; $(git rev-parse --short HEAD 2>/dev/null)" | tee "$LOG"
if [ "$HAVE" -lt 2 ]; then
  echo "# ONE GPU: RCCL refuses two ranks on one device -- only the N = 1 lines below are measurements; the N > 1 control flow is" | tee -a "$LOG"
  echo "# covered by tests/test_gather_abi.py and tests/test_gpu_multirank_smoke.py (gloo / the stand-in transport on one GPU)" | tee -a "$LOG"
fi
python -m pytest tests/test_gather_abi.py -m gpu -q -k "real_rccl" 2>&1 | tail -3 | tee -a "$LOG"
STEPS=${STEPS:-30}; WARMUP=${WARMUP:-10}
run() { # label, bench.py arguments...
  local label="$1"; shift
  echo "== $label: bench.py $*" | tee -a "$LOG"
  python bench.py "$@" --steps "$STEPS" --warmup "$WARMUP" --no-cpu-baseline --no-e2e 2>>"$LOG" | tail -1 | tee -a "$OUT" | python -c '
import json, sys
for l in sys.stdin:
    d = json.loads(l); c = d["config"]; dd = c["distributed"]
    print("   N=%d  %.1f M pairs/s  %.3f ms/step  gather=%s  transport=%s  rccl_ranks=%s  verified=%s  partition=%s  abi_leg=%s" % (
        d["n_gpus"], d["value"], d["ms_per_step"], (dd.get("gather_impl") or "-")[:5], c["gather_transport"], dd.get("rccl_comm_ranks"),
        c["gather_verified"], c["partition"][:9], (c.get("abi_gather_leg") or {}).get("value")))' | tee -a "$LOG"
}
for N in 1 2 4 8; do
  [ "$N" -gt "$MAX" ] && break
  if [ "$N" -eq 1 ]; then run "cfg2 N=1" --gpus 1; continue; fi
  for SHARE in 1.0 0.4; do
    run "cfg2 N=$N abi f64, root share $SHARE" --gpus "$N" --gather abi --root-share "$SHARE"
    run "cfg2 N=$N torch coded, root share $SHARE" --gpus "$N" --gather torch --root-share "$SHARE"
  done
  run "cfg2 N=$N as the driver runs it (no flags: torch coded + the C-ABI-gather leg)" --gpus "$N"
done
if [ "$MAX" -ge 2 ]; then
  run "cfg4 N=$MAX abi f64" --gpus "$MAX" --config cfg4 --gather abi
  run "cfg4 N=$MAX torch coded" --gpus "$MAX" --config cfg4 --gather torch
fi
echo "# lines: $OUT" | tee -a "$LOG"
