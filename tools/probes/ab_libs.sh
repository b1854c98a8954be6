# usage: ab_libs.sh "<probe command>" <tag> <tag> ...   -- runs the probe against liborn.so and each liborn_<tag>.so on one box
cmd=$1; shift
for t in "" "$@"; do s=${t:+_$t}; L=$(ls $GRAFT_REPO_ROOT/boosting*/liborn$s.so); echo "lib liborn$s.so"; ORN_LIB_PATH=$L $cmd; done
