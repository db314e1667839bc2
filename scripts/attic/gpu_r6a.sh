#!/bin/bash
# round 6, first call: the x-slab query (bit-identical), the voxelizer suites, a baseline C-host reading of the tree
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/ab
timeout 1500 python -m pytest tests/test_voxel_gpu.py tests/test_voxel_sticks_gpu.py tests/test_boundaries_gpu.py tests/test_reference_python_gpu.py -q -m gpu -x --durations=8 2>&1 | tail -25 | tee gpurun_out/pytest_r6a.log
timeout 300 scripts/cbench 300 r2_gaussian_amd/libr2hip.so single,stages,voxel > gpurun_out/ab/r6a_base.txt 2>&1
grep -E 'BEST|raster\.|^voxel|GVoxel' gpurun_out/ab/r6a_base.txt | cut -c1-200
