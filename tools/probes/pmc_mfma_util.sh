# MFMA utilisation of the forward conv kernel (one launch per fast layer): SQ_VALU_MFMA_BUSY_CYCLES against GRBM_GUI_ACTIVE
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d /tmp/pmc_u -o u -- python3 $GRAFT_REPO_ROOT/tools/probes/conv_fwd_all.py ${ORN_PREC:-fp16} ${ORN_DBG:-0} > /tmp/pmc_u.log 2>&1
python3 - <<'PY'
import sqlite3
db = sqlite3.connect('/tmp/pmc_u/u_results.db')
c = db.cursor()
rows = list(c.execute("select dispatch_id, name, counter_name, sum(counter_value), max(end - start) from pmc_events where name like '%k_conv_nhwc%' group by dispatch_id, counter_name order by dispatch_id"))
import collections
d = collections.OrderedDict()
for did, name, cn, v, dur in rows:
    d.setdefault(did, {'dur_ns': dur})[cn] = v
for did, x in d.items():
    print(did, {k: (round(v) if isinstance(v, float) else v) for k, v in x.items()})
PY
