#!/bin/bash
# Re-tune check of the fp32 planner's cost constants on whole steps (each variant its own process; the library reads them once):
#   bash tools/planner_knobs_ab.sh > gpurun_out/planner_knobs.txt
run() { echo "== $*"; env "$@" python tools/train_probe.py c2 f32 60 2>&1 | grep -v amdgpu.ids; env "$@" python tools/sample_probe.py 1 300 f32 graph 2>&1 | grep -v amdgpu.ids | tail -1; }
run SDM_NOP=1
run SDM_IGEMM_SPLIT_COST=3000
run SDM_IGEMM_SPLIT_COST=9000
run SDM_IGEMM_SPLIT_COST=12000
run SDM_IGEMM_BIGTILE_PENALTY=-600
run SDM_IGEMM_BIGTILE_PENALTY=600
run SDM_NOP=1
