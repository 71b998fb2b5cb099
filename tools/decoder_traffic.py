#!/usr/bin/env python3
"""HBM bytes per prompt of the SAM mask decoder from two rocprofv3 counter passes (--pmc FETCH_SIZE / --pmc WRITE_SIZE) over
tools/decoder_bench.py: every kernel launched between the model's construction and the end is the decoder's (23 x 23 = 529
prompts per call), summed and divided by calls x prompts.  usage: decoder_traffic.py <fetch_dir> <write_dir> <calls> <prompts> <tag>
-> profiles/<tag>_decoder_traffic.json and profiles/decoder_traffic.json (what bench.py replays)."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from profile_summary import short

fd, wd, calls, prompts, tag = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
DEC = ("dec_", "attn_fewq", "gemm_x3p_kernel", "gemm_x3_skinny", "gemm_f32_kernel", "gemm_f16x3_kernel", "layernorm", "add_rows_bcast",
       "pe_kernel", "build_tokens", "hyper_logits", "ln256", "ln_gelu64", "attn_x3_kernel", "attn_f32", "gather_rows", "attn_smallk")


def agg(d, ctr):
    out = collections.defaultdict(lambda: [0, 0.0])
    for path in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == ctr:
                k = short(r["Kernel_Name"])
                if k.startswith(DEC):
                    out[k][0] += 1
                    out[k][1] += float(r["Counter_Value"])
    return out


f, w = agg(fd, "FETCH_SIZE"), agg(wd, "WRITE_SIZE")
# the warm-up calls of decoder_bench.py are launches too: `calls` counts them
per = {}
for k in sorted(set(f) | set(w)):
    fb = 2.0 * 1024.0 * f.get(k, [0, 0.0])[1]
    wb = 1024.0 * w.get(k, [0, 0.0])[1]
    per[k] = {"launches_per_call": f.get(k, [0])[0] / calls, "fetch_bytes_per_prompt": fb / calls / prompts, "write_bytes_per_prompt": wb / calls / prompts}
tot = sum(v["fetch_bytes_per_prompt"] + v["write_bytes_per_prompt"] for v in per.values())
res = {"bytes_per_prompt": tot, "prompts_per_call": prompts, "calls": calls, "kernels": per,
       "source": f"{tag}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/decoder_bench.py, {prompts} prompts per call, "
                 "FETCH x2 x1024 + WRITE x1024, every decoder kernel summed"}
json.dump(res, open(os.path.join(ROOT, "profiles", f"{tag}_decoder_traffic.json"), "w"), indent=1)
json.dump(res, open(os.path.join(ROOT, "profiles", "decoder_traffic.json"), "w"), indent=1)
print(f"decoder: {tot / 1e6:.1f} MB per prompt over {len(per)} kernels")
for k, v in sorted(per.items(), key=lambda kv: -(kv[1]["fetch_bytes_per_prompt"] + kv[1]["write_bytes_per_prompt"]))[:8]:
    print(f"  {k[:50]:50s} {v['fetch_bytes_per_prompt'] / 1e6:7.2f} MB fetch {v['write_bytes_per_prompt'] / 1e6:7.2f} MB write per prompt")
