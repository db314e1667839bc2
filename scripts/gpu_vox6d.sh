#!/bin/bash
# voxel tests, then the (Gaussians, grid) sweep and the cloud A/B (scripts/gpu_vox6.sh) of two libraries, alternating
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${TAG:-vox6d}
LIBS=${LIBS:-"libr2hip_prev.so libr2hip.so"}
mkdir -p gpurun_out/ab
timeout 1500 python -m pytest tests/test_voxel_gpu.py tests/test_voxel_sticks_gpu.py tests/test_variants_gpu.py tests/test_dispatch_gpu.py tests/test_reported_configs_gpu.py -q -m gpu -x 2>&1 | tail -5 | tee gpurun_out/pytest_$TAG.log
for rep in 1 2; do for L in $LIBS; do
  echo "== $L rep $rep: $(R2HIP_LIB=$PWD/r2_gaussian_amd/$L python scripts/voxel_grid_sweep.py 20 2>&1 | tail -1)"
done; done | tee gpurun_out/ab/${TAG}_sweep.txt
TAG=$TAG TESTS=0 LIBS="$LIBS" bash scripts/gpu_vox6.sh
