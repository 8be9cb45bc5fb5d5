#!/usr/bin/env python3
"""profiles/<prefix>_valu_<tag>.json from the SQ counter table of tools/pmc_table.py (a rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU
SQ_ACTIVE_INST_VALU ... pass of `python3 bench.py ...`): per kernel, the vector instructions and the cycles a SIMD spent issuing them,
per launch.  bench.py reads it for roofline.valu_issue_frac (the counters cannot be read from inside the process).

usage: make_valu_json.py sq.txt particles out.json "source text" [git head]
"""
import json
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from make_traffic_json import NAMES, kernel_source_sha256  # noqa: E402  (kernel symbol -> bench.py's launch label)


def main():
    lines = open(sys.argv[1]).read().splitlines()
    cols = lines[0].split()[1:]
    res = {}
    for ln in lines[1:]:
        parts = ln.split()
        vals = parts[-len(cols):]
        name = " ".join(parts[:-len(cols)])
        if name not in NAMES:
            continue
        row = dict(zip(cols, (float(v) for v in vals)))
        waves = row.get("WAVES", 0.0)
        if not waves:
            continue
        res[NAMES[name]] = {"kernel": name, "waves": waves, "insts_valu": row.get("INSTS_VALU"), "active_inst_valu": row.get("ACTIVE_INST_V", row.get("ACTIVE_INST_VALU")),
                            "insts_valu_per_wave": row.get("INSTS_VALU", 0.0) / waves}
    doc = {"workload_particles": int(sys.argv[2]), "source": sys.argv[4], "per_launch": res}
    if len(sys.argv) > 5 and sys.argv[5]:
        doc["git_head"] = sys.argv[5]
    doc["kernel_source_sha256"] = kernel_source_sha256()
    json.dump(doc, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(res, indent=1)[:600])


if __name__ == "__main__":
    main()
