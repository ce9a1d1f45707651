export CHUNKY_ORACLE_NO_BUILD=1
bash tools/ab.sh "default bvh6k8 bvh6k4 bvh5k8" entities entities4k > gpurun_out/r06_bvh_occupancy_ab.txt 2>&1
cat gpurun_out/r06_bvh_occupancy_ab.txt
