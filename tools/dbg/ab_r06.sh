python -m pytest tests -q -m gpu -x 2>&1 | tail -6
tools/dbg/ab_step_r06.sh r06_pf3
