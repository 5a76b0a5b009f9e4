import json,sys
d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["achieved"]); print(" ".join(f'{s["N"]}x{s["K"]}e{s["epi"]}:{s["us"]}' for s in d["roofline"]["per_shape"]))
