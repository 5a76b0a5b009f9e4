python -m pytest tests/test_gpu_determinism.py -x -q -m gpu 2>&1 | tail -5
python bench.py --no-extra-legs --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('default', d['value'], d['ms_per_step'])"
AMDSEG_DETERMINISTIC=1 python bench.py --no-extra-legs --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('det', d['value'], d['ms_per_step'])"
