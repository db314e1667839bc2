set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -q -m gpu -x 2>&1 | tail -5
python __graft_entry__.py smoke 2>&1 | tail -3
python bench.py --steps 50 --warmup 10 > gpurun_out/bench1.json 2> gpurun_out/bench1.err; tail -3 gpurun_out/bench1.err; cat gpurun_out/bench1.json
