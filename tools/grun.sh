#!/bin/bash
# usage: tools/grun.sh <timeout-seconds> '<command>'   -- gpurun, re-queued only while the pod has no free GPU slot (nothing charged)
t=$1; shift
for i in $(seq 1 40); do
    /usr/local/graft/bin/gpurun --timeout $t -- "$@"
    if python3 - <<'PY'
import json,sys
try:
    d=json.load(open('/root/repo/gpurun_out/.last_call.json'))
except Exception:
    sys.exit(1)
sys.exit(0 if d.get('status')=='transient' else 1)
PY
    then echo "[grun] no slot, retry $i"; sleep 45; else exit 0; fi
done
