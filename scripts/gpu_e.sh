#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/ab
timeout 600 python -m pytest tests/test_reported_configs_gpu.py -x -q -k "config_E" 2>&1 | tail -2
for V in r05a new; do
  case $V in r05a) E="R2HIP_LIB=$PWD/r2_gaussian_amd/libr2hip_r05a.so";; new) E="R2_X=1";; esac
  env $E timeout 600 python bench.py --gaussians 1000000 --detector 1024 --views 360 --steps 300 --warmup 30 --no-voxel --no-streams --no-batched --no-forward-only --no-cpu-baseline --no-train-iteration --no-densify-pattern > gpurun_out/ab/E_$V.json 2> gpurun_out/ab/E_$V.err
  echo "== E $V: $(python -c "import json,sys; d=json.load(open('gpurun_out/ab/E_$V.json')); print(d['value'], d['ms_per_step'], {k: round(v['us'],1) for k,v in d.get('kernels',{}).items() if isinstance(v, dict) and 'us' in v})" 2>&1 | tail -1)"
done
