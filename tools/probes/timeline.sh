# kernel timeline of one training step for a library build.  usage: timeline.sh <lib.so> <tag> [extra bench.py flags, e.g. --no-graph] -> gpurun_out/timeline_<tag>.txt
lib=$1; tag=$2; shift 2
export ORN_LIB_PATH=$(realpath $lib)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tl_$tag
rocprofv3 --kernel-trace -d /tmp/tl_$tag -o tl -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-fp32 --quick --steps 8 --warmup 4 "$@" > /tmp/tl_$tag.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/probes/step_timeline.py $(find /tmp/tl_$tag -name "*.db" | head -1) ${TL_BACK:-40} > $GRAFT_REPO_ROOT/gpurun_out/timeline_$tag.txt 2>&1
