#!/bin/bash
# counter passes first (so that the bench line of the evidence set carries roofline.traffic from passes on these very sources), then the evidence set
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
bash tools/collect_pmc.sh r05 > gpurun_out/collect_pmc_r05.log 2>&1
python tools/pmc_to_json.py r05 conv_fwd_planes_w4_kernel > gpurun_out/pmc_to_json_r05.log 2>&1
mkdir -p gpurun_out/prof_r05/pmc_json; cp profiles/r05_pmc* gpurun_out/prof_r05/pmc_json/ 2>/dev/null
bash tools/collect_profiles.sh r05 > gpurun_out/collect_profiles_r05.log 2>&1
tail -3 gpurun_out/collect_pmc_r05.log gpurun_out/pmc_to_json_r05.log
