# HBM traffic of the three 16-bit conv kernels at the L4 shape of BASELINE config 2 (360x640, 96 -> 384 channels), fp16 builds,
# collected as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (with --kernel-trace only);
# counters are KiB; FETCH_SIZE tallies the 128-B requests of wide coalesced / LDS-DMA loads at 64 B, so reads are doubled.
# Writes profiles-ready JSON to gpurun_out/conv_traffic.json (copy to profiles/conv_traffic.json: bench.py reads it and checks
# the kernel symbol).
cd /tmp && export TMPDIR=/tmp && export ORN_HALF=fp16
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_t_$c
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_t_$c -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/probes/conv_kernels.py fwd,dgrad,wgrad 2 4 > /tmp/pmc_t_$c.log 2>&1
done
python3 - > $GRAFT_REPO_ROOT/gpurun_out/conv_traffic.json <<'PY'
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    for f in glob.glob(f'/tmp/pmc_t_{c}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == c:
                acc[r['Kernel_Name']][c].append(float(r['Counter_Value']))
H, W, C, O = 360, 640, 96, 384
xpad, dypad, z, wts = (H + 2) * (W + 2) * C * 2, (H + 2) * (W + 2) * O * 2, 4 * H * W * 96 * 2, 9 * O * C * 2
alg = {'k_conv_fwd_nhwc_bf16': xpad + wts + z, 'k_conv_nhwc_bf16': dypad + wts + H * W * C * 2 + (H // 2 + 2) * (W // 2 + 2) * 4 * C * 2,
       'k_wgrad_nhwc_bf16': xpad + dypad, 'k_wgrad_bf16_reduce': 0}
out = {'unit': 'bytes per launch', 'shape': 'L4 of BASELINE config 2: 360x640 pixels, 96 -> 384 channels, fp16',
       'method': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over tools/probes/conv_kernels.py '
                 '(ORN_HALF=fp16); counters are KiB; reads corrected x2 per MI355X_MICROARCH.md (HBM section)', 'kernels': []}
for k, cs in acc.items():
    if not any(t in k for t in alg):
        continue
    fe = sum(cs['FETCH_SIZE']) / max(len(cs['FETCH_SIZE']), 1) * 1024
    wr = sum(cs['WRITE_SIZE']) / max(len(cs['WRITE_SIZE']), 1) * 1024
    a = next(v for t, v in alg.items() if t in k)
    out['kernels'].append({'symbol': k, 'launches_measured': len(cs['FETCH_SIZE']), 'FETCH_SIZE_raw_bytes': fe, 'WRITE_SIZE_bytes': wr,
                           'traffic_bytes_per_launch': 2 * fe + wr, 'algorithmic_bytes_per_launch': a or None})
print(json.dumps(out, indent=1))
PY
cat $GRAFT_REPO_ROOT/gpurun_out/conv_traffic.json
