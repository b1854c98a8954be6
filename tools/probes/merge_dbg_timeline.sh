for d in 0 1 3; do ORN_MERGE_DBG=$d bash tools/probes/timeline.sh boosting*/liborn.so r03_f_dbg$d; done
