mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -q -k "f16" 2>&1 | tail -4
for v in "" "HOISDF_ATTN16=f16"; do
  echo "== bench --config 4 $v"; env $v timeout 600 python bench.py --config 4 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], [ (f['kernel'][:40], f['ms_per_step'], f['achieved_tflops']) for f in d.get('families',[])[:6]])"
done
echo "== bench --config 4 --f16-attention 0"; timeout 600 python bench.py --config 4 --f16-attention 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
} > gpurun_out/cfg4.log 2>&1
cat gpurun_out/cfg4.log
