python -m pytest tests -q -m gpu 2>&1 | tail -12
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
python bench.py --no-cpu-baseline --no-via-trainer --no-extra-legs --steps 60 --warmup 10 2>/dev/null | tail -1 | cut -c1-1500
