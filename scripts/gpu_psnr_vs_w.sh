#!/bin/bash
# PSNR-vs-views-per-step study (scripts/psnr_vs_w.py) at the headline sizes and at the reduced size the GPU test pins
#   gpurun -- bash scripts/gpu_psnr_vs_w.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python scripts/psnr_vs_w.py --out gpurun_out/psnr_vs_W.json 2>&1 | grep -v Warning | tail -30
timeout 400 python scripts/psnr_vs_w.py --detector 128 --nvol 64 --init 5000 --iterations 1200 --densify-from 200 --densify-until 700 \
   --interval 50 --grad-threshold 5e-5 --ws 8,16 --seeds 0,1,2,3 --out gpurun_out/psnr_vs_W_small.json 2>&1 | grep -v Warning | tail -30
