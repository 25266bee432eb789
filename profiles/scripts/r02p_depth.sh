mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_columns.py tests/test_gpu_wide.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -3
timeout 600 python bench.py --maf-columns 0 --text-path 0 --sustained-seconds 0 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1])
c = d['columns']; print('depth', c['value'], c['kernel_ms'], c['roofline']['frac'])"
