timeout 300 python -m pytest tests/test_gpu_pipelined.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -3
timeout 600 python bench.py --workload cfg4 --queries 1250000 --columns 0 --maf-columns 0 --text-path 0 --cpu-sample 0 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print(d['value'], d['ms_per_step'], d['one_plan']['value'])"
