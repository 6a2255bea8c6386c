# the driver-style bench three times in a row on one box (allocator / first-process effects): gpurun -- bash tools/bench_repeat.sh
for i in 1 2 3; do
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-fp32-products 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('run', d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['roofline_kernels'][0]['launch_ms'], d['roofline_kernels'][1]['launch_ms'])"
done
python - <<'PY'
import torch
print(torch.cuda.get_device_properties(0).total_memory/2**30, [x/2**30 for x in torch.cuda.mem_get_info(0)])
PY
