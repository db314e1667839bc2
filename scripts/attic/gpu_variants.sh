#!/bin/bash
# time several builds of the library through the C-ABI harness: scripts/gpu_variants.sh libA.so libB.so ...
cd $GRAFT_REPO_ROOT
for L in "$@"; do
  echo "=== $L"
  for rep in 1 2; do timeout 200 scripts/cbench ${STEPS:-300} r2_gaussian_amd/$L > gpurun_out/_v.txt 2>&1
    grep -E "BEST|raster\.render|raster.geom|^voxel 256|^voxel 32|BATCH V=4" gpurun_out/_v.txt | grep -v "V=" | tr '\n' ';' | sed 's/  */ /g'; grep "BATCH V=4" gpurun_out/_v.txt | cut -c1-40; done
done
