#!/usr/bin/env python3
"""Turn the counter passes of tools/profile_all.sh into the summary bench.py reads.

    tools/make_traffic.py <tag> <round> [<tag2> ...]     e.g.  tools/make_traffic.py r02_f1 r02 r02_stress

Reads gpurun_out/<tag>/{pmc_*/, bench_*.json, bench_kernel_stats.csv}, copies the per-kernel summaries to
profiles/<round>/<tag>_*, and writes profiles/<round>/traffic.json: per workload and kernel the HBM bytes per launch
(FETCH_SIZE x 2: gfx950 tallies 128-byte fabric read requests at 64 bytes; WRITE_SIZE as counted; both in KiB), the
vector-instruction issue fraction, vector-L1 accesses per clock per CU, mean resident waves and how wave cycles split
into issuing / stalled at issue / parked in a wait.  The file records the hash of the kernel sources it was taken on;
bench.py ignores it once those sources change.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = {"wf_extend_kernel": "wf_extend_kernel<false", "wf_shade_kernel": "wf_shade_kernel<false", "path_kernel": "path_kernel<false",
           "wf_generate_kernel": "wf_generate_kernel", "accumulate_kernel": "accumulate_kernel"}
N_CU, N_SIMD, N_XCD = 256, 1024, 8


def pass_sums(d):
    """{kernel short name: ({counter: sum}, launches)} of one rocprofv3 --pmc output directory."""
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(set)
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            for short, pat in KERNELS.items():
                if pat in r["Kernel_Name"]:
                    agg[short][r["Counter_Name"]] += float(r["Counter_Value"])
                    calls[short].add(r["Dispatch_Id"])
    return {k: (dict(v), len(calls[k])) for k, v in agg.items()}


def kernel_ms(bench_json, kernel):
    try:
        d = json.load(open(bench_json))
    except Exception:
        return None
    ks = d.get("kernel_share", {})
    r = d.get("roofline", {})
    if r.get("kernel") == kernel:
        return r.get("kernel_ms_mean")
    if kernel == "wf_shade_kernel" and ks.get("shade_launches"):
        return round(ks["shade_ms"] / ks["shade_launches"], 4)
    return None


def main():
    tags, rnd = [sys.argv[1]] + sys.argv[3:], sys.argv[2]
    dst = os.path.join(ROOT, "profiles", rnd)
    os.makedirs(dst, exist_ok=True)
    out_path = os.path.join(dst, "traffic.json")
    doc = {"_how": __doc__.strip().split("\n\n")[-1].replace("\n", " "), "workloads": {}}
    for tag in tags:
        src = os.path.join(ROOT, "gpurun_out", tag)
        bench = json.load(open(os.path.join(src, "bench_default.json")))
        doc["source_hash"] = bench["source_hash"]
        key = bench["workload_key"]
        w = {"bench_value_mrays_per_s": bench["value"], "seconds_per_frame": bench["seconds_per_frame"]}
        passes = {n: pass_sums(os.path.join(src, f"pmc_{n}")) for n in ("fetch", "write", "sq", "stall", "stall2", "l2")}
        for short in KERNELS:
            k = {}
            f, nf = passes["fetch"].get(short, ({}, 0))
            wr, nw = passes["write"].get(short, ({}, 0))
            if nf and nw:
                k["launches"] = nf
                k["fetch_size_kib_sum"] = f.get("FETCH_SIZE", 0.0)
                k["write_size_kib_sum"] = wr.get("WRITE_SIZE", 0.0)
                k["hbm_bytes_per_launch"] = round((2.0 * f.get("FETCH_SIZE", 0.0) / nf + wr.get("WRITE_SIZE", 0.0) / nw) * 1024.0)
                ms = kernel_ms(os.path.join(src, "pmc_fetch", "bench.json"), short)
                if ms:
                    k["kernel_ms_mean"] = ms
            sq, ns = passes["sq"].get(short, ({}, 0))
            if ns and sq.get("GRBM_GUI_ACTIVE"):
                cycles = sq["GRBM_GUI_ACTIVE"] / N_XCD
                k["valu_issue_frac"] = round(min(1.0, sq.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / N_SIMD / cycles), 4)
                k["l1_accesses_per_clk_per_cu"] = round(sq.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0.0) / N_CU / cycles, 4)
                k["l1_miss_requests_per_access"] = round(sq.get("TCP_TCC_READ_REQ_sum", 0.0) / max(1.0, sq.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0.0)), 4)
                k["mean_waves_in_flight"] = round(sq.get("SQ_WAVE_CYCLES", 0.0) * 4.0 / cycles)
                k["resident_wave_limit"] = None
                k["valu_insts_per_wave"] = round(sq.get("SQ_INSTS_VALU", 0.0) / max(1.0, sq.get("SQ_WAVES", 0.0)), 1)
                k["valu_lane_utilisation"] = round(sq.get("SQ_THREAD_CYCLES_VALU", 0.0) / max(1.0, 64.0 * sq.get("SQ_ACTIVE_INST_VALU", 0.0)), 4)
                k["gpu_cycles_per_launch"] = round(cycles / ns)
            st, nst = passes["stall"].get(short, ({}, 0))
            if nst and st.get("SQ_WAVE_CYCLES"):
                wc = st["SQ_WAVE_CYCLES"]
                k["wave_cycle_split"] = {"issuing_any": round(st.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 4),
                                         "stalled_at_issue": round(st.get("SQ_WAIT_INST_ANY", 0.0) / wc, 4),
                                         "parked_in_wait": round(st.get("SQ_WAIT_ANY", 0.0) / wc, 4),
                                         "issuing_vmem": round(st.get("SQ_ACTIVE_INST_VMEM", 0.0) / wc, 4),
                                         "issuing_lds": round(st.get("SQ_ACTIVE_INST_LDS", 0.0) / wc, 4)}
                k["vmem_insts_per_launch"] = round(st.get("SQ_INSTS_VMEM", 0.0) / nst)
                k["lds_insts_per_launch"] = round(st.get("SQ_INSTS_LDS", 0.0) / nst)
            l2, nl2 = passes["l2"].get(short, ({}, 0))
            if nl2 and (l2.get("TCC_HIT_sum", 0.0) + l2.get("TCC_MISS_sum", 0.0)) > 0:
                # L2 (TCC) look-ups of the kernel's launches, all 8 XCDs: hits / (hits + misses); requests per launch
                k["l2_hit_rate"] = round(l2.get("TCC_HIT_sum", 0.0) / (l2.get("TCC_HIT_sum", 0.0) + l2.get("TCC_MISS_sum", 0.0)), 4)
                k["l2_requests_per_launch"] = round(l2.get("TCC_REQ_sum", 0.0) / nl2)
                k["l2_misses_per_launch"] = round(l2.get("TCC_MISS_sum", 0.0) / nl2)
                k["l1_to_l2_read_requests_per_launch"] = round(l2.get("TCP_TCC_READ_REQ_sum", 0.0) / nl2)
            s2, ns2 = passes["stall2"].get(short, ({}, 0))
            if ns2:
                k["stall2_sums"] = {c: v for c, v in s2.items() if c != "GRBM_GUI_ACTIVE"}
            if k:
                w[short] = k
        # the dominant kernel's launch priced three ways against 8 TB/s (the bench line of another workload shows these)
        r = bench.get("roofline", {})
        kk = w.get(r.get("kernel", ""), {})
        if kk.get("hbm_bytes_per_launch") and kk.get("kernel_ms_mean"):
            w["three_fractions"] = {
                "demand_cache_served": r.get("survey_demand_frac"),
                "counter_traffic": round(kk["hbm_bytes_per_launch"] / (kk["kernel_ms_mean"] * 1e-3) / 1e9 / 8000.0, 4),
                "compulsory": r.get("frac"),
                "hbm_bytes_per_launch": kk["hbm_bytes_per_launch"], "algorithmic_bytes_per_launch": r.get("algorithmic_bytes_per_launch"),
                "kernel_ms_mean": kk["kernel_ms_mean"], "l2_hit_rate": kk.get("l2_hit_rate"), "value_mrays_per_s": bench["value"]}
        doc["workloads"][key] = w
        for name in ("bench_default.json", "bench_under_rocprof.json", "bench_kernel_stats.csv", "pmc_fetch.txt", "pmc_write.txt", "pmc_sq.txt",
                     "pmc_stall.txt", "pmc_stall2.txt", "pmc_l2.txt"):
            p = os.path.join(src, name)
            if os.path.exists(p):
                shutil.copy(p, os.path.join(dst, f"{tag}_{name}"))
    json.dump(doc, open(out_path, "w"), indent=1)
    print(json.dumps(doc["workloads"], indent=1)[:3000])


if __name__ == "__main__":
    main()
