"""HBM traffic of the trace (and deferred shade) kernels from a tools/prof_summary.py text.

usage: python tools/traffic_from_summary.py profiles/<tag>_rocprof.txt width height spp rays > profiles/<name>.json

FETCH_SIZE and WRITE_SIZE are reported in units of 1024 bytes; on gfx950 FETCH_SIZE tallies the
128-byte requests of 16-byte-per-lane loads at 64 bytes, so it is doubled; WRITE_SIZE is taken as reported.
"""
import json
import re
import sys

path, w, h, spp, rays = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5])
fetch = write = launches = 0
for line in open(path):
    m = re.match(r"\s+(trace_kernel<\S.*?|shade_kernel<\S.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+(\d+)\s+\((\d+) dispatches\)", line)
    if m and m.group(2) == "FETCH_SIZE":
        fetch += int(m.group(3))
        launches += int(m.group(4))
    elif m:
        write += int(m.group(3))
hbm = (2 * fetch + write) * 1024
ps = w * h * spp
print(json.dumps({
    "kernels": f"trace_kernel and deferred shade_kernel launches of one {spp}-spp pass at {w}x{h} ({path})",
    "samples_per_pass": spp, "launches": launches, "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write,
    "hbm_bytes_per_launch": hbm / max(launches, 1), "pixel_samples": ps, "hbm_bytes_per_pixel_sample": hbm / ps,
    "hbm_bytes_per_ray": hbm / rays,
    "note": "separate rocprofv3 --pmc passes (tools/profile.sh); FETCH_SIZE doubled (gfx950 tallies the 128-byte "
            "requests of 16-byte-per-lane loads at 64 bytes); WRITE_SIZE as reported",
}, indent=1))
