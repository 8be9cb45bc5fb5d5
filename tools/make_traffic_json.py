#!/usr/bin/env python3
"""profiles/<prefix>_traffic_<tag>.json from the FETCH_SIZE / WRITE_SIZE tables of tools/summarize_profile.py (separate rocprofv3
passes of `python3 bench.py ...`).  bench.py reads it for roofline.traffic (the counters cannot be read from inside the process).

usage: make_traffic_json.py fetch.txt write.txt particles out.json "source text" [git head] [skip_steps]
"""
import json
import sys

NAMES = {  # kernel symbol -> bench.py's launch label
    "k_neighbor_build<true>": "neighbor_build+density_alpha", "k_neighbor_build<1>": "neighbor_build+density_alpha",
    "k_neighbor_build<2>": "neighbor_build+density_alpha+density_change", "k_neighbor_build<3>": "neighbor_build+density_alpha+divergence_warmstart", "k_neighbor_build<0>": "neighbor_build", "k_nonpressure": "nonpressure_accel_vmax",
    "k_compute_error<false>": "compute_density_error", "k_compute_error<true>": "compute_density_change",
    "k_compute_error<false, false>": "compute_density_error", "k_compute_error<true, false>": "compute_density_change",
    "k_compute_error<false, true>": "velocity_prediction+compute_density_error",
    "k_correct<false, true>": "correct_velocity_with_density_error", "k_correct<false, false>": "correct_velocity_with_divergence_error",
    "k_correct<true, true>": "correct_density_error_warmstart", "k_correct<true, false>": "correct_divergence_error_warmstart",
    "k_correct<false, true, false>": "correct_velocity_with_density_error", "k_correct<false, true, true>": "correct_velocity_with_density_error",
    "k_correct<false, false, false>": "correct_velocity_with_divergence_error",
    "k_correct<true, true, false>": "correct_density_error_warmstart", "k_correct<true, false, false>": "correct_divergence_error_warmstart",
    "k_rank_gather": "gather_attributes", "k_key_count<true>": "advect+cell_count", "k_predict": "velocity_prediction", "k_scatter": "cell_scatter",
}


def kernel_source_sha256():
    import hashlib
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for f in ("sphx_kernels.hip", "sphx_launch.inc", "sphx_internal.hpp", "sphx_sqrt.hpp"):
        h.update(open(os.path.join(root, "yasph2d_amd", "csrc", f), "rb").read())
    return h.hexdigest()


def table(path):
    out = {}
    for line in open(path):
        parts = line.rstrip().split()
        if len(parts) >= 5 and parts[-1].replace(".", "").isdigit() and parts[-2].isdigit():
            name = " ".join(parts[:-4])
            out[name] = int(parts[-2])
    return out


if __name__ == "__main__":
    fetch, write = table(sys.argv[1]), table(sys.argv[2])
    res = {}
    for k, label in NAMES.items():
        if k in fetch or k in write:
            f, w = fetch.get(k, 0), write.get(k, 0)
            res[label] = {"fetch": f, "write": w, "total": f + w}
    scan = {"fetch": fetch.get("k_scan_reduce", 0) + fetch.get("k_scan_apply", 0) + fetch.get("k_scan_onepass", 0),
            "write": write.get("k_scan_reduce", 0) + write.get("k_scan_apply", 0) + write.get("k_scan_onepass", 0)}
    scan["total"] = scan["fetch"] + scan["write"]
    res["cell_scan"] = scan
    doc = {"workload_particles": int(sys.argv[3]), "source": sys.argv[5], "bytes_per_launch": res}
    if len(sys.argv) > 6 and sys.argv[6]:
        doc["git_head"] = sys.argv[6]  # the build the counters were taken from (bench.py quotes it next to roofline.traffic)
    if len(sys.argv) > 7 and sys.argv[7]:
        doc["skip_steps"] = int(sys.argv[7])  # the window the counters belong to (bench.py --skip-steps); absent: from t = 0
    doc["kernel_source_sha256"] = kernel_source_sha256()  # bench.py only quotes a record taken from the kernels it runs
    json.dump(doc, open(sys.argv[4], "w"), indent=1)
    print(json.dumps(res, indent=1)[:400])
