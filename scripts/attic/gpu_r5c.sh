#!/bin/bash
# round 5: parity of the two-kernel tile-first chain, A/B against the round-4 library, in-kernel stamps
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/ab gpurun_out/ts
timeout 900 python -m pytest tests/test_tilefirst_gpu.py tests/test_raster_gpu.py tests/test_variants_gpu.py tests/test_knn_gpu.py -x -q 2>&1 | tail -15
for rep in 1 2; do
  for L in libr2hip_base.so libr2hip.so R2_TF_SLABS=1:libr2hip.so R2_TF_SLABS=2:libr2hip.so; do
    E=""; F=$L; case $L in *:*) E=${L%%:*}; F=${L#*:};; esac
    env $E timeout 200 scripts/cbench ${STEPS:-300} r2_gaussian_amd/$F single,stages > gpurun_out/ab/${L}_$rep.txt 2>&1
    echo "== $L (rep $rep): $(grep -E 'BEST|raster\.' gpurun_out/ab/${L}_$rep.txt | tr '\n' ';' | sed 's/  */ /g')"
  done
done
for L in R2_TF_SLABS=1:libr2hip_ts.so; do
  E=""; F=$L; case $L in *:*) E=${L%%:*}; F=${L#*:};; esac
  env $E timeout 200 scripts/cbench 100 r2_gaussian_amd/$F single > gpurun_out/ts/$F.txt 2>&1
  echo "== $L"; grep -E "BEST|TS (geom|tilefirst)" gpurun_out/ts/$F.txt
done
for V in base new s1 s2 s4; do
  case $V in base) E="R2HIP_LIB=$PWD/r2_gaussian_amd/libr2hip_base.so";; new) E="R2_X=1";; s1) E="R2_TF_SLABS=1";; s2) E="R2_TF_SLABS=2";; s4) E="R2_TF_SLABS=4";; esac
  env $E timeout 300 python bench.py --cloud large --no-voxel --no-streams --no-batched --no-forward-only --no-cpu-baseline > gpurun_out/ab/trained_large_$V.json 2> gpurun_out/ab/trained_large_$V.err
  echo "== trained large $V: $(python -c "import json,sys; d=json.load(open('gpurun_out/ab/trained_large_$V.json')); print(d['value'], d['ms_per_step'], {k: round(v['us'],1) for k,v in d.get('kernels',{}).items() if isinstance(v, dict) and 'us' in v})" 2>&1 | tail -1)"
done
