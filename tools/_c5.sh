#!/bin/bash
R=$GRAFT_REPO_ROOT
python - <<PY
import numpy as np, time
from revrand_amd import _hip
o=np.empty(2_000_000,dtype=np.int64)
for i in range(3):
    b=np.random.RandomState(7)
    t1=time.perf_counter(); pb=_hip.legacy_permutation(b,2_000_000,out=o); t2=time.perf_counter()
    print("%.2f ms native perm"%(1e3*(t2-t1)))
PY
timeout 600 python -m pytest tests/test_gpu_resident_sgd.py tests/test_gpu_glm.py tests/test_gpu_multigpu.py -q -x 2>&1 | tail -2
REPS=5 python $R/tools/c5_resident.py host resident 8 200 2>&1 | grep -E "intervals|per step"
REPS=2 python $R/tools/c5_resident.py host hostloop 8 72 2>&1 | grep -E "intervals|per step"
