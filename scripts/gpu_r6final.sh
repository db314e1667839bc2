#!/bin/bash
# round 6: the whole GPU suite in both file orders + smoke, then the profile set (bench lines, rocprofv3 kernel stats, PMC passes)
cd $GRAFT_REPO_ROOT
TAG=${1:-r06c}
bash scripts/gpu_suite2.sh $TAG
bash scripts/gpu_profile6.sh $TAG
