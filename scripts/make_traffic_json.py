#!/usr/bin/env python
"""profiles/traffic.json from the two PMC summaries of the bench command (scripts/gpu_bench_profile.sh):
HBM-side bytes per launch = 2 x FETCH_SIZE (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE, both in KiB."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse(path, counter):
    out, name = {}, None
    for line in open(path):
        m = re.match(r"^(\S.*?)\s+calls\s+\d+\s+avg", line)
        if m:
            name = m.group(1).strip()
            continue
        m = re.match(r"^\s+%s\s+([0-9.]+)" % counter, line)
        if m and name:
            out[name] = float(m.group(1))
    return out


TAG = sys.argv[1] if len(sys.argv) > 1 else "r03"
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (kernel_source_digest)

fetch = parse(os.path.join(ROOT, "profiles", "%s_bench_pmc_fetch.txt" % TAG), "FETCH_SIZE")
write = parse(os.path.join(ROOT, "profiles", "%s_bench_pmc_write.txt" % TAG), "WRITE_SIZE")
keys = {"k_bwd_b": "k_bwd_b", "k_bwd_a": "k_bwd_a", "k_stress<true, true, false>": "k_stress_fwd_save", "k_velocity<true, false>": "k_velocity_fwd"}
res = {"_how": "rocprofv3 --pmc FETCH_SIZE and (separate pass) --pmc WRITE_SIZE on `python bench.py --steps 1 --warmup 0 --nsteps 400 "
               "--no-cpu-baseline` (profiles/%s_bench_pmc_fetch.txt, %s_bench_pmc_write.txt); FETCH_SIZE doubled per the gfx950 "
               "correction of MI355X_MICROARCH.md (calibrated in round 1 on a kernel with known compulsory bytes: 10 arrays x 9.19 MB = 91.9 MB vs 2 x 45.1 MB "
               "counted), WRITE_SIZE as counted; KiB -> bytes.  Regenerate with scripts/make_traffic_json.py <tag>." % (TAG, TAG),
       "kernel_source_sha256": bench.kernel_source_digest()}
for pat, key in keys.items():
    f = [v for k, v in fetch.items() if pat in k]
    w = [v for k, v in write.items() if pat in k]
    if not f or not w:
        if pat.startswith("k_bwd_"):      # the two-launch step does not run where the persistent loop does
            continue
        sys.exit("missing %s" % pat)
    res[key + "_fetch_kib"] = f[0]
    res[key + "_write_kib"] = w[0]
    res[key + "_bytes_per_launch"] = int(round((2.0 * f[0] + w[0]) * 1024))
# the persistent backward loop: one launch per shot = (nsteps - 1) time steps of the PMC command (bench.py --nsteps 400)
fp_ = [v for k, v in fetch.items() if "k_bwd_persist" in k]
wp_ = [v for k, v in write.items() if "k_bwd_persist" in k]
if fp_ and wp_:
    steps = int(sys.argv[2]) - 1 if len(sys.argv) > 2 else 399
    res["k_bwd_persist_fetch_kib_per_launch"] = fp_[0]
    res["k_bwd_persist_write_kib_per_launch"] = wp_[0]
    res["k_bwd_persist_bytes_per_time_step"] = int(round((2.0 * fp_[0] + wp_[0]) * 1024 / steps))
json.dump(res, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k.endswith("per_launch")}, indent=1))
