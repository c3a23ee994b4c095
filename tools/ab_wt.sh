for w in 0 1 2 3 0 3; do
  GC_TUNE_WT_STORES=$w python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --rollout-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['roofline']['class_ms_per_call']
print('wt=$w', d['value'], 'ffw', c['gc_gemm_ffw1'], 'rowop', c['gc_rowop'], 'mlp', c['gc_mlp'], 'seg', c['gc_segsum'], 'qkv', c['gc_gemm_qkv'], 'ffw_us', d['roofline']['avg_launch_us'])"
done
