# SQ counters of the conv kernels at the L4 shape: where the wave cycles go (busy / parked on s_waitcnt or barrier / issue stalls),
# MFMA pipe busy, LDS bank conflicts.  usage: pmc_conv.sh <fwd|dgrad|wgrad> <tag>      (writes gpurun_out/pmc_<tag>.txt)
which=${1:-fwd}; tag=${2:-$which}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_c1 /tmp/pmc_c2
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/pmc_c1 -o u --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/probes/conv_kernels.py $which 3 > /tmp/pmc_c1.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU -d /tmp/pmc_c2 -o u --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/probes/conv_kernels.py $which 3 > /tmp/pmc_c2.log 2>&1
python3 - "$which" > $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}.txt <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ('/tmp/pmc_c1', '/tmp/pmc_c2'):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if 'k_conv_nhwc' in k or 'k_conv_fwd' in k or 'k_wgrad' in k or 'k_dgrad2' in k:
                acc[k[:110]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f'   {c:28s} n={len(v):3d} mean={sum(v)/len(v):.5g}')
PY
tail -3 /tmp/pmc_c1.log /tmp/pmc_c2.log >> $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}.txt
cat $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}.txt
