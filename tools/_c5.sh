#!/bin/bash
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_resident_sgd.py tests/test_gpu_glm.py -q -x 2>&1 | tail -3
python $R/tools/c5_resident.py host resident 8 136
RR_GLM_PREFETCH_STAGES=1 python $R/tools/c5_resident.py host resident 8 136
python $R/tools/c5_resident.py host hostloop 8 72
RR_GLM_PREFETCH_STAGES=1 python $R/tools/c5_resident.py host hostloop 8 72
