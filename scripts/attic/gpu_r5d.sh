#!/bin/bash
# round 5: parity of the balanced / epilogue-free tile-first chain, A/B against the round-4 library, in-kernel stamps
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/ab gpurun_out/ts
timeout 900 python -m pytest tests/test_tilefirst_gpu.py tests/test_raster_gpu.py tests/test_variants_gpu.py tests/test_knn_gpu.py tests/test_threads_gpu.py -x -q 2>&1 | tail -15
for rep in 1 2; do
  for L in libr2hip_base.so libr2hip.so; do
    timeout 200 scripts/cbench ${STEPS:-300} r2_gaussian_amd/$L single,stages > gpurun_out/ab/${L}_$rep.txt 2>&1
    echo "== $L (rep $rep): $(grep -E 'BEST|raster\.' gpurun_out/ab/${L}_$rep.txt | tr '\n' ';' | sed 's/  */ /g')"
  done
done
timeout 200 scripts/cbench 100 r2_gaussian_amd/libr2hip_ts.so single > gpurun_out/ts/libr2hip_ts.so.txt 2>&1
echo "== stamps"; grep -E "BEST|TS (geom|tilefirst)" gpurun_out/ts/libr2hip_ts.so.txt
for V in base new; do
  case $V in base) E="R2HIP_LIB=$PWD/r2_gaussian_amd/libr2hip_base.so";; new) E="R2_X=1";; esac
  for C in large small; do
  env $E timeout 300 python bench.py --cloud $C --no-voxel --no-streams --no-batched --no-forward-only --no-cpu-baseline > gpurun_out/ab/trained_${C}_$V.json 2> gpurun_out/ab/trained_${C}_$V.err
  echo "== trained $C $V: $(python -c "import json,sys; d=json.load(open('gpurun_out/ab/trained_${C}_$V.json')); print(d['value'], d['ms_per_step'], {k: round(v['us'],1) for k,v in d.get('kernels',{}).items() if isinstance(v, dict) and 'us' in v})" 2>&1 | tail -1)"
  done
  env $E timeout 300 python bench.py --gaussians 50000 --no-voxel --no-streams --no-batched --no-forward-only --no-cpu-baseline > gpurun_out/ab/B_$V.json 2> gpurun_out/ab/B_$V.err
  echo "== config B $V: $(python -c "import json,sys; d=json.load(open('gpurun_out/ab/B_$V.json')); print(d['value'], d['ms_per_step'], {k: round(v['us'],1) for k,v in d.get('kernels',{}).items() if isinstance(v, dict) and 'us' in v})" 2>&1 | tail -1)"
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab/bench_new.json 2> gpurun_out/ab/bench_new.err; cut -c1-300 gpurun_out/ab/bench_new.json; tail -3 gpurun_out/ab/bench_new.err
