# MFMA utilisation of the three big 16-bit conv kernels at the L4 shape of BASELINE config 2 (360x640, 96 <-> 384 channels, fp16
# builds, random data): SQ_VALU_MFMA_BUSY_CYCLES (cycles of matrix-pipe work, summed over the 1024 SIMDs) against GRBM_GUI_ACTIVE
# (summed over the 8 XCDs) -> busy fraction = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024).  One --pmc pass with --kernel-trace only.
# Writes gpurun_out/mfma_util.json.
cd /tmp && export TMPDIR=/tmp && export ORN_HALF=fp16
rm -rf /tmp/pmc_u
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY -d /tmp/pmc_u -o u --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/probes/conv_kernels.py fwd,dgrad,wgrad 6 4 > /tmp/pmc_u.log 2>&1
python3 - > $GRAFT_REPO_ROOT/gpurun_out/mfma_util.json <<'PY'
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/pmc_u/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if any(t in k for t in ('k_conv2_nhwc', 'k_conv_fwd_nhwc', 'k_conv_nhwc_bf16', 'k_wgrad_nhwc')):
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
dur = collections.defaultdict(list)
for f in glob.glob('/tmp/pmc_u/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r['Kernel_Name']].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
out = {'shape': 'L4 of BASELINE config 2: 360x640 pixels, 96 <-> 384 channels, fp16 builds, random operands',
       'method': 'tools/probes/pmc_mfma_util.sh: rocprofv3 --kernel-trace --pmc (one pass) over tools/probes/conv_kernels.py; '
                 'mfma_busy_fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024); durations are under the profiler',
       'kernels': []}
for k, cs in sorted(acc.items()):
    m = lambda c: sum(cs[c]) / max(len(cs[c]), 1)
    gui, busy = m('GRBM_GUI_ACTIVE'), m('SQ_VALU_MFMA_BUSY_CYCLES')
    d = sorted(dur.get(k, [0.0]))
    out['kernels'].append({'kernel': k, 'launches': len(cs['GRBM_GUI_ACTIVE']), 'dur_us_median': d[len(d) // 2], 'GRBM_GUI_ACTIVE': gui,
                           'SQ_VALU_MFMA_BUSY_CYCLES': busy, 'SQ_WAVE_CYCLES': m('SQ_WAVE_CYCLES'), 'SQ_WAIT_ANY': m('SQ_WAIT_ANY'),
                           'gfx_clock_GHz': (gui / 8) / (d[len(d) // 2] * 1e3) if d[len(d) // 2] else None,
                           'mfma_busy_fraction': busy / (gui / 8 * 1024) if gui else None})
print(json.dumps(out, indent=1))
PY
python3 -c "
import json,os
for k in json.load(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/mfma_util.json'))['kernels']: print('%-70s %6.1f us busy %.3f clock %.2f' % (k['kernel'][:70], k['dur_us_median'], k['mfma_busy_fraction'], k['gfx_clock_GHz'] or 0))"
