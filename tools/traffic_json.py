"""HBM traffic of the trace kernels from the FETCH_SIZE / WRITE_SIZE passes of tools/profile.sh.

usage: python tools/traffic_json.py gpurun_out/prof_<tag> <pixel_samples> > profiles/<name>.json

FETCH_SIZE and WRITE_SIZE are reported in KiB-like units of 1024 bytes; on gfx950 FETCH_SIZE
counts a 128-byte request of a 16-byte-per-lane load as 64 bytes, so it is doubled; WRITE_SIZE
is taken as reported.
"""
import glob
import json
import os
import sqlite3
import sys


def counter(root, name):
    total, launches = 0.0, 0
    for path in glob.glob(os.path.join(root, f"pmc_{name}", "*.db")):
        db = sqlite3.connect(path)
        rows = db.execute("select name, sum(counter_value), count(*) from pmc_events "
                          "where counter_name = ? group by name", (name,)).fetchall()
        for kernel, value, n in rows:
            if "trace_kernel" in kernel or "extend_kernel" in kernel or "shade_kernel" in kernel:
                total += value
                launches += n
    return total, launches


def main(root, pixel_samples):
    fetch, launches = counter(root, "FETCH_SIZE")
    write, _ = counter(root, "WRITE_SIZE")
    hbm = (2.0 * fetch + write) * 1024.0
    print(json.dumps({
        "kernels": "trace_kernel / extend_kernel / shade_kernel launches of one pass",
        "launches": launches,
        "FETCH_SIZE_KB": fetch,
        "WRITE_SIZE_KB": write,
        "hbm_bytes_per_launch": hbm / max(launches, 1),
        "pixel_samples": pixel_samples,
        "hbm_bytes_per_pixel_sample": hbm / pixel_samples,
        "note": "separate rocprofv3 --pmc passes (tools/profile.sh); FETCH_SIZE doubled (gfx950 tallies the "
                "128-byte requests of 16-byte-per-lane loads at 64 bytes); WRITE_SIZE as reported",
    }, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]))
