# compile-time timing ablations of k_conv2_nhwc<0> (the dgrad of the large images; results are wrong by construction): build the
# variants with
#   ORN_BUILD_TAG=nw ORN_EXTRA_DEFS=-DC2_ABL_NO_WDMA   (np: -DC2_ABL_NO_PDMA, nwp: both)
# through boosting*/_build.py (build(force=True) under those environment variables), then run this on the box.
for t in "" _nw _np _nwp; do L=$(ls $GRAFT_REPO_ROOT/boosting*/liborn$t.so); echo "lib liborn$t.so"; ORN_LIB_PATH=$L ORN_HALF=fp16 python tools/probes/conv_kernels.py dgrad 30 4; done
