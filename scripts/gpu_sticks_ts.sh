#!/bin/bash
# in-kernel stamps of the stick-first voxel chain (experiment build) + the C host's timing of the 256^3 query on both chains
mkdir -p gpurun_out/sticks
[ -f scripts/_scene/scene.bin ] || python scripts/dump_scene.py > /dev/null 2>&1
R2_TS_DUMP=gpurun_out/sticks/ts_sticks.bin timeout 200 scripts/cbench 20 r2_gaussian_amd/libr2hip_ts.so voxel 2>&1 | grep -E "^voxel|stamps|voxel\." | head -20
python scripts/sticks_timeline.py gpurun_out/sticks/ts_sticks.bin | tee gpurun_out/sticks/timeline.txt
for M in 0 1; do R2_VOXEL_STICKS=$M timeout 200 scripts/cbench 20 r2_gaussian_amd/libr2hip.so voxel 2>&1 | grep -E "^voxel 256|  voxel\." | tr '\n' ';'; echo; done
