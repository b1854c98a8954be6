# timeline of a step with the forward-merge launches altered by ORN_MERGE_DBG (1: no pack riders, 2: no stem riders; 4: one launch per problem, needs --no-graph)
# usage: merge_dbg_timeline.sh [--no-graph] <flags> ...
extra=""; if [ "$1" = "--no-graph" ]; then extra="--no-graph"; shift; fi
for d in "$@"; do ORN_MERGE_DBG=$d bash tools/probes/timeline.sh boosting*/liborn.so dbg$d $extra; done
