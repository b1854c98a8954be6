# HBM traffic of every conv kernel of the 720p fp16 training step, per SYMBOL as bench.py prices them (the engine's own launches,
# not per-op probes), collected as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (with
# --kernel-trace only); counters are KiB; FETCH_SIZE tallies the 128-B requests of wide coalesced / LDS-DMA loads at 64 B, so
# reads are doubled.  Writes gpurun_out/conv_traffic.json (copy to profiles/conv_traffic.json: bench.py reads it, matching symbols).
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_s_$c
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_s_$c -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-fp32 --quick --mode graph --steps 4 --warmup 2 > /tmp/pmc_s_$c.log 2>&1
  tail -1 /tmp/pmc_s_$c.log | cut -c1-200
done
python3 - > $GRAFT_REPO_ROOT/gpurun_out/conv_traffic.json <<'PY'
import csv, glob, json, collections, os, sys
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import bench
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    for f in glob.glob(f'/tmp/pmc_s_{c}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == c:
                acc[r['Kernel_Name']][c].append(float(r['Counter_Value']))
cfg = bench.CONFIGS['720p']
geo = bench.layer_geo(cfg)
nl, ff = len(geo), bench.first_fast_layer(geo, 'fp16')
def sizes(L):
    H, W, O, s = L['H'], L['W'], L['O'], L['s']
    return dict(xpad=(H + 2) * (W + 2) * 96 * 2, dypad=(H + 2) * (W + 2) * O * 2, z=H * s * W * s * (O // (s * s)) * 2,
                apad=(H * s + 2) * (W * s + 2) * (O // (s * s)) * 2, wts=9 * O * 96 * 2, zprev=H * W * 96 * 2, dw=9 * O * 96 * 4,
                tiles=((W + 31) // 32) * ((H + 7) // 8))
alg = collections.defaultdict(lambda: [0, 0])          # symbol fragment -> [bytes per step, launches per step]
def add(k, b):
    alg[k][0] += b; alg[k][1] += 1
for i in range(ff, nl):
    L, S = geo[i], sizes(geo[i])
    last = i == nl - 1
    fwd = S['xpad'] + S['wts'] + S['z'] + (0 if last else S['apad'])
    if last and S['tiles'] >= 128: add('k_conv2_nhwc<2>', fwd)
    elif last: add('k_conv_fwd_nhwc_bf16<4, 2, 2, 2, 3,', fwd)
    elif L['C'] <= 32: add('k_conv_fwd_nhwc_bf16<4, 2, 2, 2, 0, 32', fwd)
    else: add('k_conv_fwd_nhwc_bf16<4, 2, 2, 2, 0, 96', fwd)
    if i > ff and S['tiles'] >= 128: add('k_conv2_nhwc<0>', S['dypad'] + S['wts'] + S['zprev'] + S['zprev'])   # + z_prev read, dy_prev written
    add('k_wgrad_nhwc_bf16_all', 0); alg['k_wgrad_nhwc_bf16_all'][1] = 1
    alg['k_wgrad_nhwc_bf16_all'][0] += S['xpad'] + S['dypad'] + S['dw']
out = {'csrc_hash': bench.csrc_hash(),      # the build these numbers belong to: bench.py refuses the file for any other
       'unit': 'bytes per launch (mean over the launches of the symbol in the step)',
       'shape': 'BASELINE config 2 training step (720p, ERB), fp16 engine: the launches bench.py times',
       'method': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over bench.py --mode graph --steps 4 --warmup 2 '
                 '(graph replays + the eager profile steps); counters are KiB; reads corrected x2 per MI355X_MICROARCH.md (HBM section)',
       'kernels': []}
for k, cs in sorted(acc.items()):
    if not any(t in k for t in ('k_conv', 'k_wgrad', 'k_dgrad', 'k_head', 'k_adam', 'k_fusion6', 'k_loss', 'k_gemm', 'k_mgemm', 'k_stage0', 'k_w2', 'k_merge')):
        continue
    fe = sum(cs['FETCH_SIZE']) / max(len(cs['FETCH_SIZE']), 1) * 1024
    wr = sum(cs['WRITE_SIZE']) / max(len(cs['WRITE_SIZE']), 1) * 1024
    a = next((v for t, v in alg.items() if t in k), None)
    out['kernels'].append({'symbol': k, 'launches_measured': len(cs['FETCH_SIZE']), 'FETCH_SIZE_raw_bytes': fe, 'WRITE_SIZE_bytes': wr,
                           'traffic_bytes_per_launch': 2 * fe + wr,
                           'algorithmic_bytes_per_launch': (a[0] / a[1]) if a else None, 'launches_per_step': a[1] if a else None})
print(json.dumps(out, indent=1))
PY
python3 -c "
import json,os
d=json.load(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/conv_traffic.json'))
for k in d['kernels']: print('%-90s n=%3d traffic %7.1f MB  alg %s' % (k['symbol'][:90], k['launches_measured'], k['traffic_bytes_per_launch']/1e6, ('%.1f MB' % (k['algorithmic_bytes_per_launch']/1e6)) if k['algorithmic_bytes_per_launch'] else '-'))"
