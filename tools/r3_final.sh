#!/bin/bash
# end of round 3: the whole GPU suite, smoke(), the driver's bench command
out=gpurun_out/${1:-final}
mkdir -p $out
( time timeout 3000 python -m pytest tests -m gpu -q ) > $out/pytest_gpu.log 2>&1; echo "gpu suite rc=$?"; tail -4 $out/pytest_gpu.log | cut -c1-300
if [ -n "$DEBUGLIB" ]; then ( time REVRAND_HIP_LIB=$PWD/revrand_amd/lib/librevrand_hip_debug.so timeout 3000 python -m pytest tests -m gpu -q ) > $out/pytest_gpu_debuglib.log 2>&1; echo "gpu suite on the bounds-checking build rc=$?"; tail -4 $out/pytest_gpu_debuglib.log | cut -c1-300; fi
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $out/smoke.log | cut -c1-300
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; tail -3 $out/bench.err | cut -c1-200
python - <<PY
import json
l=[x for x in open("$out/bench.json") if x.startswith("{")]
d=json.loads(l[-1])
print(d["value"], d["ms_per_step"], {k:(round(v,4) if isinstance(v,float) else v) for k,v in d["roofline"].items() if not isinstance(v,(dict,list,str))})
for k,v in d.get("configs",{}).items():
    print(k, {kk:(round(vv,4) if isinstance(vv,float) else vv) for kk,vv in v.items() if isinstance(vv,(int,float))}, {kk:(round(vv,4) if isinstance(vv,float) else vv) for kk,vv in v.get("roofline",{}).items() if isinstance(vv,(int,float))})
PY
