#!/bin/bash
# depth slabs of the tile-first chain: parity with forced slab counts, then timing on the headline and the trained clouds
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/ab
for S in 2 4; do R2_TF_SLABS=$S timeout 600 python -m pytest tests/test_tilefirst_gpu.py -x -q 2>&1 | tail -2; done
timeout 600 python -m pytest tests/test_tilefirst_gpu.py tests/test_raster_gpu.py tests/test_autograd_gpu.py tests/test_threads_gpu.py -x -q 2>&1 | tail -2
for rep in 1 2; do
  for L in ${LIBS:-libr2hip_r05a.so libr2hip.so}; do
    E=""; F=$L; case $L in *:*) E=${L%%:*}; F=${L#*:};; esac
    env $E timeout 200 scripts/cbench 300 r2_gaussian_amd/$F single,stages > gpurun_out/ab/${L}_$rep.txt 2>&1
    echo "== $L (rep $rep): $(grep -E 'BEST|raster\.' gpurun_out/ab/${L}_$rep.txt | tr '\n' ';' | sed 's/  */ /g')"
  done
done
for V in r05a auto; do
  case $V in r05a) E="R2HIP_LIB=$PWD/r2_gaussian_amd/libr2hip_r05a.so";; auto) E="R2_X=1";; esac
  for C in large small; do
  env $E timeout 300 python bench.py --cloud $C --no-voxel --no-streams --no-batched --no-forward-only --no-cpu-baseline --no-train-iteration --no-densify-pattern > gpurun_out/ab/trained_${C}_$V.json 2> gpurun_out/ab/trained_${C}_$V.err
  echo "== trained $C $V: $(python -c "import json,sys; d=json.load(open('gpurun_out/ab/trained_${C}_$V.json')); print(d['value'], d['ms_per_step'], {k: round(v['us'],1) for k,v in d.get('kernels',{}).items() if isinstance(v, dict) and 'us' in v})" 2>&1 | tail -1)"
  done
done
