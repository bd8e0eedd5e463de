"""profiles/traffic.json from a PMC summary of tools/pmc_profile.sh (the file bench.py reads for roofline.traffic and roofline_valu).
usage: python tools/make_traffic_json.py profiles/r04_final_pmc_summary.json [--round 4]"""
import argparse
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("summary")
    ap.add_argument("--round", default="4")
    ap.add_argument("--isa-table", default=os.path.join(ROOT, "profiles", "r04_hot_loop_isa_table.md"))
    a = ap.parse_args()
    s = json.load(open(a.summary))
    m = lambda k: s[k]["mean_per_launch"]  # noqa: E731
    read_b = int(m("FETCH_SIZE") * 1024 * 2)
    write_b = int(m("WRITE_SIZE") * 1024)
    slow_share = None
    if os.path.exists(a.isa_table):
        t = re.search(r"\| \*\*total \(static\)\*\* \| (\d+) \| (\d+) \| (\d+) \|", open(a.isa_table).read())
        if t:
            f, i, sl = (int(x) for x in t.groups())
            slow_share = round(sl / (f + i + sl), 3)
    out = {
        "workload": "terrain16_1080p",
        "kernel": "trace_stack_kernel",
        "hbm_bytes_per_launch": read_b + write_b,
        "read_bytes": read_b,
        "write_bytes": write_b,
        "valu_wave_instructions_per_launch": int(m("SQ_INSTS_VALU")),
        "salu_wave_instructions_per_launch": int(m("SQ_INSTS_SALU")),
        "valu_slow_group_share_static": slow_share,
        "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU / SQ_INSTS_SALU (separate passes, tools/pmc_profile.sh) on the head build of round {a.round}, "
                  f"{os.path.relpath(a.summary, ROOT)}; slow-group share: static count over the main loop, profiles/r04_hot_loop_isa_table.md",
        "correction": "FETCH_SIZE x 1024 x 2: calibrated with tools/calib_fetch.py on single-dword gathers (profiles/r01_fetch_calibration.csv): 2^21 lines at a "
                      "128-B stride -> 2,097,352 TCC_EA0_RDREQ and FETCH_SIZE 131,084 KB; at a 64-B stride -> 1,048,744 requests, 65,546 KB, i.e. one request per "
                      "128-B line counted as 64 B. WRITE_SIZE x 1024 as is.",
    }
    json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
