# compile-time timing ablations of k_dgrad2_nhwc (results are wrong by construction): build the variants with
#   ORN_BUILD_TAG=nw ORN_EXTRA_DEFS=-DD2_ABL_NO_WDMA   (np: -DD2_ABL_NO_PDMA, nwp: both, nepi: -DD2_ABL_NO_EPI), then run this on the box
for t in "" _nw _np _nwp _nepi; do L=$(ls $GRAFT_REPO_ROOT/boosting*/liborn$t.so); echo "lib liborn$t.so"; ORN_LIB_PATH=$L ORN_HALF=fp16 python tools/probes/conv_kernels.py dgrad 30 4; done
