#!/bin/bash
# stick-first voxel binning: its tests, the neighbouring voxel tests, and the A/B timing of the 256^3 query (one box)
mkdir -p gpurun_out/sticks
timeout 900 python -m pytest tests/test_voxel_sticks_gpu.py -q --tb=short -p no:cacheprovider > gpurun_out/sticks/pytest_sticks.log 2>&1
echo "sticks tests rc=$?"; tail -25 gpurun_out/sticks/pytest_sticks.log
timeout 600 python scripts/voxel_ab.py 20 > gpurun_out/sticks/ab.txt 2>&1
echo "ab rc=$?"; cat gpurun_out/sticks/ab.txt | tail -12
if [ "$1" = "more" ]; then
  timeout 900 python -m pytest tests/test_voxel_gpu.py tests/test_threads_gpu.py tests/test_multistream_gpu.py -q --tb=short -p no:cacheprovider > gpurun_out/sticks/pytest_voxel.log 2>&1
  echo "voxel tests rc=$?"; tail -8 gpurun_out/sticks/pytest_voxel.log
fi
