# usage: tools/ab_env.sh VAR v1 v2 [v1 v2 ...]: bench class times (nano + 1 degree) for each value of an A/B switch, same box
var=$1; shift
for v in "$@"; do
  env $var=$v python bench.py --steps 10 --warmup 2 --no-cpu-baseline --rollout-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['roofline']['class_ms_per_call']
k=('gc_mlp','gc_rowop','gc_gemm_qkv','gc_attention','gc_gemm_out','gc_gemm_ffw1','gc_gemm_ffw2')
print('$var=$v nano', d['value'], {x:c[x] for x in k})
o=d['one_degree']; c=o['roofline']['class_ms_per_call']; print('$var=$v 1deg', o['value'], {x:c[x] for x in k})"
done
