# per-op L4..L1 wgrad (+ reduce) time for each library given: tools/probes/wgrad_variants.sh lib1.so lib2.so ..
for l in "$@"; do echo "== $l"; ORN_LIB_PATH=$PWD/$l python3 tools/probes/wgrad_ablate.py quick 2>&1 | grep flags; done
