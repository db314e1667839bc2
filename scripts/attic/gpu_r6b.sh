#!/bin/bash
# round 6: x-slab tests after the cross-section chunk rule + the rule's effect on the query time of every grid (base = the R rule)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/ab
timeout 1500 python -m pytest tests/test_voxel_gpu.py tests/test_voxel_sticks_gpu.py tests/test_boundaries_gpu.py tests/test_reference_python_gpu.py tests/test_reported_configs_gpu.py -q -m gpu --durations=8 2>&1 | tail -25 | tee gpurun_out/pytest_r6b.log
for rep in 1 2; do
  for L in libr2hip_base.so libr2hip.so; do
    echo "== $L rep $rep: $(R2HIP_LIB=$PWD/r2_gaussian_amd/$L timeout 300 python scripts/voxel_grid_sweep.py 20 2>&1 | tail -1)"
  done
done | tee gpurun_out/ab/r6b_voxel_chunk_rule.txt
