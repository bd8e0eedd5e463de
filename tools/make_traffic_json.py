"""profiles/traffic.json from a PMC summary of tools/pmc_profile.sh (the file bench.py reads for roofline.traffic and roofline_valu).
usage: python tools/make_traffic_json.py profiles/r05_final_pmc_summary.json --round 5 --timeline profiles/r05_final_timeline_heavy.json
       --isa-table profiles/r05_hot_loop_isa_table.md"""
import argparse
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("summary")
    ap.add_argument("--round", default="4")
    ap.add_argument("--isa-table", default=os.path.join(ROOT, "profiles", "r05_hot_loop_isa_table.md"))
    ap.add_argument("--kernel-ms", type=float, default=None, help="rocprofv3 / HIP-event average duration of the trace kernel in the same build")
    ap.add_argument("--timeline", default=None, help="tools/wave_timeline.py --json of the heavy (phase-clock) build: clock held in the kernel, rounds, cycles per round")
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
    clock = ledger = None
    if a.timeline and os.path.exists(a.timeline):
        tl = json.load(open(a.timeline))
        clock = tl.get("shader_clock_ghz_in_kernel")
        # issue ledger: wave64 issue cycles of one wave-round by the cost table of tools/issue_rate.hip against the
        # cycles a round takes in the product
        # by MEASURED instruction counts (static per-section counts x frequencies overcount: inlined copies): the static shares of the
        # three groups in the main loop's ISA weight SQ_INSTS_VALU
        shares = None
        if os.path.exists(a.isa_table):
            t = re.search(r"\| \*\*total \(static\)\*\* \| (\d+) \| (\d+) \| (\d+) \|", open(a.isa_table).read())
            if t:
                f, i, sl = (int(x) for x in t.groups())
                shares = {"F": round(f / (f + i + sl), 3), "I": round(i / (f + i + sl), 3), "S": round(sl / (f + i + sl), 3)}
        if shares and a.kernel_ms:
            valu = m("SQ_INSTS_VALU")
            rounds = tl["total_rounds"]
            per_round = valu / rounds
            lo = per_round * (shares["S"] * 4.25 + shares["I"] * 2.3 + max(shares["F"] - shares["S"], 0.0) * 2.3)
            hi = per_round * (shares["S"] * 4.25 + (shares["I"] + shares["F"]) * 2.3)
            waves = m("SQ_WAVES")
            cyc_round = a.kernel_ms * 1e-3 * clock * 1e9 / (rounds / waves)
            ledger = {"valu_wave_instructions_per_wave_round": round(per_round, 1), "static_group_shares": shares,
                      "weighted_issue_cycles_per_wave_round": {"every_f32_pairs_with_a_4_cycle_instruction": round(lo), "no_instruction_pairs": round(hi)},
                      "wave_rounds_per_launch": rounds, "waves": int(waves), "kernel_ms": a.kernel_ms,
                      "cycles_per_wave_round_of_the_product": round(cyc_round), "waves_per_simd": 7,
                      "issue_port_busy": [round(7 * lo / cyc_round, 2), round(min(7 * hi / cyc_round, 9.99), 2)],
                      "timeline_build_cycles_per_round": tl["cycles_per_round"], "wave_iters_per_round": tl["descent"]["wave_iters_per_round"],
                      "note": "costs per wave64 instruction from tools/issue_rate.hip (F 2.3, I 2.3, S 4.25 cycles; an F instruction issues beside an S one for nothing); "
                              "rounds from the timeline build; a figure above 1 means the cost table overstates this mix -- the port cannot be more than busy"}
    out = {
        "workload": "terrain16_1080p",
        "kernel": "trace_stack_kernel",
        "hbm_bytes_per_launch": read_b + write_b,
        "read_bytes": read_b,
        "write_bytes": write_b,
        "valu_wave_instructions_per_launch": int(m("SQ_INSTS_VALU")),
        "salu_wave_instructions_per_launch": int(m("SQ_INSTS_SALU")),
        "valu_slow_group_share_static": slow_share,
        "shader_clock_ghz_in_kernel": clock,
        "issue_ledger": ledger,
        "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU / SQ_INSTS_SALU (separate passes, tools/pmc_profile.sh) on the head build of round {a.round}, "
                  f"{os.path.relpath(a.summary, ROOT)}; slow-group share: static count over the main loop, {os.path.relpath(a.isa_table, ROOT)}; clock and ledger: {a.timeline}",
        "correction": "FETCH_SIZE x 1024 x 2: calibrated with tools/calib_fetch.py on single-dword gathers (profiles/r01_fetch_calibration.csv): 2^21 lines at a "
                      "128-B stride -> 2,097,352 TCC_EA0_RDREQ and FETCH_SIZE 131,084 KB; at a 64-B stride -> 1,048,744 requests, 65,546 KB, i.e. one request per "
                      "128-B line counted as 64 B. WRITE_SIZE x 1024 as is.",
    }
    json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
