#!/usr/bin/env python
"""profiles/pmc_traffic.json from the output of tools/run_pmc_instep.sh (two separate rocprofv3 --pmc passes: FETCH_SIZE, WRITE_SIZE over the
real bench step): HBM bytes per launch of every kernel class the launch timer knows = 2 x FETCH_SIZE (the gfx950 correction of
MI355X_MICROARCH.md: the counter reports half the bytes of the 16-B/lane loads these kernels issue) + WRITE_SIZE, both in KB per dispatch.
usage: python tools/pmc_to_json.py gpurun_out/<tag>_pmc_instep.txt <git hash> [source path recorded in the json]"""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# algorithmic bytes per launch.  Round 4, second session: the FFN pre-activation tensor became a one-byte derivative (100.7 MB less
# per layer over the eight NT launches: 1.356e8 -> 1.230e8 on average); AdamW runs as two launches (eagerly zeroed front part + encoder layers:
# 3.38e9 B per step together)
CLASSES = {"gemm_nt_dp_kernel": 1.230e8, "gemm_tn_dp_kernel": 4.3e8, "attn_fwd_kernel": 1.14e8, "attn_bwd_dq_kernel": 1.64e8,
           "attn_bwd_dkv_kernel": 1.65e8, "ln_bwd": 1.007e8, "add_ln_fwd": 1.007e8, "adamw_kernel": 1.69e9, "attn_keepmask_kernel": 2.52e7}
# round 6 (ABI 14): eleven of the 24 dropout + residual + LayerNorm launches of a step also write the next layer's keep masks (add_ln_fwd_km_kernel:
# + 2.52e7 B); the launch timer files both under AMDSEG_PROF_ADD_LN_FWD, so the class's algorithmic bytes are the dispatch-weighted mean of the two
ALGO_BY_NAME = {"add_ln_fwd_km_kernel": 1.007e8 + 2.52e7}


def main():
    src, git = sys.argv[1], sys.argv[2]
    label = sys.argv[3] if len(sys.argv) > 3 else src
    acc, algo = {}, {}
    ctr = None
    measured_sha = None
    for ln in open(src):
        m = re.match(r"## csrc_sha: (\S+)", ln)
        if m:
            measured_sha = m.group(1)          # stamped by tools/run_pmc_instep.sh WHEN the passes ran (not when this converter runs)
            continue
        m = re.match(r"## pass: --pmc (\S+)", ln)
        if m:
            ctr = m.group(1)
            continue
        m = re.match(r"\| `(.*?)` \| (\d+) \| (\S+) \| (\S+) \| (\S+) \|", ln)
        if not m or ctr is None:
            continue
        name, n, per = m.group(1), int(m.group(2)), float(m.group(5))
        for cls in CLASSES:
            if cls in name:
                d = acc.setdefault(cls, {}).setdefault(ctr, [0.0, 0])
                d[0] += per * n; d[1] += n
                if ctr == "FETCH_SIZE":
                    a = algo.setdefault(cls, [0.0, 0])
                    a[0] += next((v for k, v in ALGO_BY_NAME.items() if k in name), CLASSES[cls]) * n; a[1] += n
    now = __import__("spokennlp_amd.build", fromlist=["sources_sha"]).sources_sha()
    if measured_sha is None:
        sys.exit(f"{src} carries no '## csrc_sha:' line (written by tools/run_pmc_instep.sh at measurement time): refusing to stamp it with today's sources")
    if measured_sha != now:
        sys.exit(f"{src} was measured on kernel sources {measured_sha}; the shipped sources hash to {now}: re-run tools/run_pmc_instep.sh")
    kernels = {}
    for cls, d in acc.items():
        if "FETCH_SIZE" not in d or "WRITE_SIZE" not in d:
            continue
        f = d["FETCH_SIZE"][0] / d["FETCH_SIZE"][1] * 1024.0
        w = d["WRITE_SIZE"][0] / d["WRITE_SIZE"][1] * 1024.0
        key = {"ln_bwd": "ln_bwd_kernel", "add_ln_fwd": "add_ln_fwd_kernel"}.get(cls, cls)
        kernels[key] = dict(hbm_bytes_per_launch=float(f"{2 * f + w:.4g}"), fetch_bytes_x2=float(f"{2 * f:.4g}"), write_bytes=float(f"{w:.4g}"),
                            algorithmic_bytes_per_launch=float(f"{algo[cls][0] / algo[cls][1]:.4g}"), launches=d["FETCH_SIZE"][1])
    out = {"_comment": "HBM bytes per launch of the profiled kernel classes from the committed rocprofv3 --pmc passes (FETCH_SIZE doubled per the gfx950 "
                       "correction of MI355X_MICROARCH.md, + WRITE_SIZE; two separate passes, --kernel-trace only).  bench.py reads this file for "
                       "roofline.traffic; written by tools/pmc_to_json.py from tools/run_pmc_instep.sh output.  PMC counters cannot be read from inside "
                       "the process, so the figure is NOT re-measured by a bench run: `git` names the commit the passes ran on and `csrc_sha` (spokennlp_amd.build.sources_sha) the kernel sources -- bench.py reports "
                       "traffic: null + traffic_stale when the shipped sources hash differently.",
           "source": label, "git": git,
           "csrc_sha": measured_sha,
           "command": "python bench.py --no-cpu-baseline --no-via-trainer --no-roofline --no-extra-legs --steps 4 --warmup 2",
           "workload": {"model": "bert", "mode": "train", "seq_len": 512, "seqs_per_gpu": 32, "workload": "full_da", "precision": "bf16"},
           "kernels": kernels}
    json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
    print(json.dumps(kernels, indent=1))


if __name__ == "__main__":
    main()
