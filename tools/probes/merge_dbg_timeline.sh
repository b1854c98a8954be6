# timeline of a step with the forward-merge launches altered by ORN_MERGE_DBG (1: no pack riders, 2: no stem riders; 4: one launch per problem, needs --no-graph)
# needs a diagnostic build: ORN_BUILD_TAG=probe ORN_EXTRA_DEFS=-DORN_PROBE_BUILD python -m orn_amd._build (the product library compiles the switch out)
# usage: merge_dbg_timeline.sh [--no-graph] <flags> ...
extra=""; if [ "$1" = "--no-graph" ]; then extra="--no-graph"; shift; fi
for d in "$@"; do ORN_MERGE_DBG=$d bash tools/probes/timeline.sh boosting*/liborn_probe.so dbg$d $extra; done
