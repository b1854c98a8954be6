# round-end evidence on one box: GPU tests, smoke, the driver's bench command, the default bench, rocprofv3 stats, PMC traffic, MFMA busy
tag=$1
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_gputest.log 2>&1; tail -3 gpurun_out/${tag}_gputest.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout -k 10 600 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_driver_cmd.json; cut -c1-300 gpurun_out/${tag}_bench_driver_cmd.json
timeout -k 10 600 python bench.py 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_default.json; cut -c1-300 gpurun_out/${tag}_bench_default.json
bash tools/probes/kstats.sh boosting-neural-video-representation-via-online-structural-reparameteration_amd/liborn.so $tag
TL_BACK=60 bash tools/probes/timeline.sh boosting-neural-video-representation-via-online-structural-reparameteration_amd/liborn.so ${tag}_stream --mode stream
bash tools/probes/mode_ab.sh > gpurun_out/${tag}_modes.log 2>&1; cat gpurun_out/${tag}_modes.log
bash tools/probes/pmc_traffic_step.sh > gpurun_out/${tag}_traffic.log 2>&1; tail -30 gpurun_out/${tag}_traffic.log | cut -c1-160
bash tools/probes/pmc_mfma_util.sh 2>&1 | tail -8
