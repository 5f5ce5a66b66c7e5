"""Derived figures of one configuration from its rocprofv3 summary, merged into its roofline file.

usage: python tools/derive_counters.py <tag>_<config>_rocprof.txt <tag>_<config>_roofline.json [git-head]

Reads the text tools/prof_summary.py wrote (kernel trace + counter passes of tools/profile.sh) and
the JSON tools/config_roofline.py wrote for the same render, and adds to the JSON:
  lanes_active      SQ_THREAD_CYCLES_VALU / (64 * SQ_INSTS_VALU): share of the 64 lanes a vector
                    instruction had switched on, averaged over instructions
  waves_per_simd    4 * SQ_WAVE_CYCLES / (32 * SQ_BUSY_CYCLES): SQ_WAVE_CYCLES counts quad-cycles summed
                    over waves, SQ_BUSY_CYCLES cycles per shader engine of one XCD (8 CUs = 32 SIMDs;
                    a dispatch reports 32 of them)
  valu_issue_share  4 * SQ_ACTIVE_INST_VALU / (32 * SQ_BUSY_CYCLES): share of a SIMD's busy cycles in
                    which a vector instruction was executing
  wait_share        SQ_WAIT_ANY / SQ_WAVE_CYCLES
  l2_hit_rate       TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)
  hbm_bytes         (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE doubled as gfx950 tallies the
                    128-byte requests of 16-byte-per-lane loads at 64 bytes, WRITE_SIZE as reported;
                    per launch, per pixel-sample, per ray, and as GB/s of the kernels' summed time
over the trace kernels (every variant) and the deferred shade kernel, plus the same per kernel name.
The JSON is stamped with the hash of the device sources (bench.kernel_source_sha) and the git head.
"""
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

txt, js = sys.argv[1], sys.argv[2]
head = sys.argv[3] if len(sys.argv) > 3 else os.environ.get("GIT_HEAD", "")
res = json.load(open(js))
counters = defaultdict(lambda: defaultdict(float))   # kernel -> counter -> value (max over passes that repeat it)
times = {}                                            # kernel -> (calls, total_ms)
section = None
for line in open(txt):
    if line.startswith("=="):
        section = line
        continue
    if section and "kernel trace" in section:
        m = re.match(r"(\S.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+\d+\s+\d+\s+\d+\s*$", line)
        if m and m.group(1) != "kernel":
            times[m.group(1).strip()] = (int(m.group(2)), float(m.group(3)))
        continue
    m = re.match(r"\s+(\S.*?)\s+([A-Z][A-Za-z0-9_]+)\s+(\d+)\s+\((\d+) dispatches\)", line)
    if m:
        k, c, v = m.group(1).strip(), m.group(2), float(m.group(3))
        counters[k][c] = max(counters[k][c], v)


def derive(names):
    c = defaultdict(float)
    ms = 0.0
    calls = 0
    for n in names:
        for k, v in counters[n].items():
            c[k] += v
        if n in times:
            calls += times[n][0]
            ms += times[n][1]
    d = {"kernels": sorted(names), "launches": calls, "kernel_ms": ms}
    if c["SQ_INSTS_VALU"]:
        d["valu_instructions"] = c["SQ_INSTS_VALU"]
        d["lanes_active"] = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_INSTS_VALU"])
    if c["SQ_BUSY_CYCLES"]:
        d["waves_per_simd"] = 4.0 * c["SQ_WAVE_CYCLES"] / (32.0 * c["SQ_BUSY_CYCLES"])
        d["valu_issue_share"] = 4.0 * c["SQ_ACTIVE_INST_VALU"] / (32.0 * c["SQ_BUSY_CYCLES"])
    if c["SQ_WAVE_CYCLES"]:
        d["wait_share"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
    if c["TCC_HIT_sum"] + c["TCC_MISS_sum"]:
        d["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    if c["FETCH_SIZE"] or c["WRITE_SIZE"]:
        hbm = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
        d["hbm_bytes"] = hbm
        d["hbm_bytes_per_launch"] = hbm / max(calls, 1)
        if ms:
            d["hbm_gbps"] = hbm / (ms * 1e-3) / 1e9
    return d


hot = [k for k in counters if k.startswith("trace_kernel") or k.startswith("shade_kernel")]
d = derive(hot)
m = re.match(r"\S+ (\d+)x(\d+), (\d+) spp", res.get("config", ""))
if m and "hbm_bytes" in d:
    ps = float(m.group(1)) * float(m.group(2)) * float(m.group(3))
    d["hbm_bytes_per_pixel_sample"] = d["hbm_bytes"] / ps
    d["hbm_bytes_per_ray"] = d["hbm_bytes"] / res["rays"]
    d["valu_instructions_per_ray"] = d.get("valu_instructions", 0.0) / res["rays"]
d["hbm_peak_gbps"] = 8000.0
res["derived"] = d
res["derived_per_kernel"] = {k: derive([k]) for k in sorted(hot)}
res["kernel_source_sha"] = bench.kernel_source_sha()
res["git_head"] = head
res["counters_from"] = os.path.basename(txt)
json.dump(res, open(js, "w"), indent=1)
print(json.dumps(res["derived"], indent=1))
