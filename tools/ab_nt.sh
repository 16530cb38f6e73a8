for lib in librevrand_hip.so librevrand_hip_nt.so; do
  echo "== $lib"
  REVRAND_HIP_LIB=$PWD/revrand_amd/lib/$lib python bench.py --rows 4000000 --steps 2 --warmup 1 --no-cpu-baseline --no-alt-engine --configs c4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['configs']['C4_fastfood_f16384']
print('value', d['value'], 'syrk', r['kernel_ms_per_step'], 'feat', r['other_kernels_ms_per_step'], 'wp', r['whole_path_frac'])
print('c4', c['ms_per_pass'], c['roofline']['frac'])"
done
