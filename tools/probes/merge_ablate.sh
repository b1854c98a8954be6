# per-op stamps of the forward-merge GEMM for diagnostic builds liborn_mabl<flags>.so (ORN_BUILD_TAG=mabl<flags> ORN_EXTRA_DEFS="-DORN_MERGE_STAMP -DG2_ABL=<flags>")
for a in "$@"; do echo "ABL $a"; ORN_LIB_PATH=$(ls $PWD/boosting*/liborn_mabl$a.so) python tools/probes/merge_stamps.py 2>&1 | grep -E "per-op|in-step"; done
