export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_tilefirst_gpu.py -q -m gpu -x 2>&1 | tail -3
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], "tv", d["voxelizer"]["tv_patch_32cube_fwd_bwd_us"], "sort", d["kernels"].get("raster.sort"))'
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --cloud large 2>/dev/null | python -c "$P" trained_large
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "$P" full
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-batched 2>/dev/null | python -c "$P" no-batched
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-forward-only 2>/dev/null | python -c "$P" no-fwdonly
timeout 300 python bench.py --steps 200 --warmup 20 --no-batched --no-forward-only 2>/dev/null | python -c "$P" with-cpu-baseline
