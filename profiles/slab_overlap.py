#!/usr/bin/env python3
"""Overlap of the slab step's two streams, from a rocprofv3 --kernel-trace CSV of a run with several ranks in ONE process
(bench.py --gpus N --one-gpu --transport local): how much of the time the halo kernels of the comm streams (pack, ghost
unpack + cell table, (rho, p) copies, header post, the device-to-device copies of the local transport) run concurrently
with a pair kernel (k_density / k_force) of the process, and how many kernels of each kind ran per step and rank.

    python profiles/slab_overlap.py <kernel_trace.csv> <steps> <ranks> [out.json]
"""
import csv, json, sys
from collections import defaultdict

path, steps, ranks = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rows = []
with open(path, newline="") as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sph::", "")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "?"), r.get("Thread_Id", "?")))
rows.sort()
pair_raw = [(s, e) for s, e, n, q, t in rows if n.startswith("k_density") or n.startswith("k_force")]
n_pair = len(pair_raw)
pair = []                                   # union of the pair kernels' intervals (several ranks overlap each other)
for s_, e_ in pair_raw:
    if pair and s_ <= pair[-1][1]:
        pair[-1] = (pair[-1][0], max(pair[-1][1], e_))
    else:
        pair.append((s_, e_))
comm = [(s, e, n) for s, e, n, q, t in rows if n.startswith("k_slab_") or n.startswith("__amd_rocclr_copyBuffer")]


def overlap(a, b):
    return max(0, min(a[1], b[1]) - max(a[0], b[0]))


tot = defaultdict(int); ov = defaultdict(int); cnt = defaultdict(int)
j0 = 0
for s, e, n in comm:
    tot[n] += e - s; cnt[n] += 1
    while j0 < len(pair) and pair[j0][1] < s:
        j0 += 1
    j = j0
    while j < len(pair) and pair[j][0] < e:
        ov[n] += overlap((s, e), pair[j]); j += 1
queues = sorted({q for _, _, _, q, _ in rows})
res = {"steps": steps, "ranks": ranks, "queues_seen": queues,
       "pair_kernel_launches_per_step_and_rank": n_pair / steps / ranks,
       "comm_kernels": {n: {"launches_per_step_and_rank": cnt[n] / steps / ranks, "total_us": tot[n] / 1e3,
                            "concurrent_with_a_pair_kernel_us": ov[n] / 1e3,
                            "share_concurrent": (ov[n] / tot[n]) if tot[n] else 0.0} for n in sorted(tot)}}
text = json.dumps(res, indent=1)
if len(sys.argv) > 4:
    open(sys.argv[4], "w").write(text + "\n")
print(text)
