for v in w6ns w6ns_nont; do
cd /tmp && export TMPDIR=/tmp && CHUNKY_HIP_LIB=$GRAFT_REPO_ROOT/.variants/libchunky_hip_$v.so rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r02h_$v -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-roofline --steps 4 > $GRAFT_REPO_ROOT/gpurun_out/r02h_$v.log 2>&1
cd $GRAFT_REPO_ROOT; tail -1 gpurun_out/r02h_$v.log | cut -c1-200; find gpurun_out/r02h_$v -name "*kernel_stats.csv" | head -1 | xargs head -5
done
