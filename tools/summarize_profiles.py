"""Turn a gpurun_out/profN{,f,w,s} rocprofv3 collection into the committed profiles/ artefacts:
  <tag>_kernel_stats.csv   (rocprofv3 --kernel-trace --stats summary, verbatim)
  <tag>_hbm_traffic_per_kernel.json + profiles/hbm_traffic_per_kernel.json (PMC FETCH_SIZE x2 [gfx950 correction] + WRITE_SIZE)
  <tag>_sq_summary.txt     (MFMA-busy fraction, sustained clock, wait breakdown per kernel)
usage: python tools/summarize_profiles.py gpurun_out/prof6 r01g [@suffix]
With a suffix (e.g. @config5: the set-transformer step) the kernels are merged into profiles/hbm_traffic_per_kernel.json
under "<kernel><suffix>" instead of replacing the file (bench.py reads both workloads' dominant kernels from it)."""
import collections
import csv
import json
import os
import shutil
import sys


def load(path, cname):
    agg, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != cname:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace(", false>", ">")   # (round 6: the FLAT template flag)
        agg[k] += float(r["Counter_Value"])
        cnt[k] += 1
    return agg, cnt


def main(base, tag, suffix=""):
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    shutil.copy(os.path.join(base, "kt_kernel_stats.csv"), os.path.join(out, f"{tag}_kernel_stats.csv"))
    fa, fc = load(base + "f/f_counter_collection.csv", "FETCH_SIZE")
    wa, wc = load(base + "w/w_counter_collection.csv", "WRITE_SIZE")
    traffic = {}
    for k in fa:
        if "dib_" not in k:
            continue
        name = k.split("<128")[0] if "fused" in k else (k.split("<")[0] if "attn" in k else k)
        traffic[name] = {"hbm_read_bytes_per_launch": round(2 * fa[k] / fc[k] * 1024),   # gfx950: FETCH_SIZE counts 1/2
                         "hbm_write_bytes_per_launch": round(wa.get(k, 0) / max(wc.get(k, 1), 1) * 1024),
                         "launches_sampled": fc[k]}
    json.dump(traffic, open(os.path.join(out, f"{tag}_hbm_traffic_per_kernel.json"), "w"), indent=1)
    cur = os.path.join(out, "hbm_traffic_per_kernel.json")
    if suffix:
        merged = json.load(open(cur)) if os.path.exists(cur) else {}
        merged = {k: v for k, v in merged.items() if not k.endswith(suffix)}
        merged.update({k + suffix: v for k, v in traffic.items()})
        json.dump(merged, open(cur, "w"), indent=1)
    else:
        keep = {k: v for k, v in (json.load(open(cur)) if os.path.exists(cur) else {}).items() if "@" in k}
        json.dump(dict(traffic, **keep), open(cur, "w"), indent=1)
    rows = list(csv.DictReader(open(base + "s/s_counter_collection.csv")))
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt, dur, seen = collections.Counter(), collections.defaultdict(float), set()
    for r in rows:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace(", false>", ">")   # (round 6: the FLAT template flag) + " grid=" + r["Grid_Size"]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            cnt[k] += 1
            dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    with open(os.path.join(out, f"{tag}_sq_summary.txt"), "w") as fh:
        fh.write("# MFMA_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCDs * 1024 SIMDs); clk = GRBM_GUI_ACTIVE/8/time\n")
        for k, v in agg.items():
            if "gemm" not in k and "fused" not in k and "attn" not in k:
                continue
            n, wcyc = cnt[k], v["SQ_WAVE_CYCLES"]
            util = v["SQ_VALU_MFMA_BUSY_CYCLES"] / max(v["GRBM_GUI_ACTIVE"], 1) * 8 / 1024
            clk = v["GRBM_GUI_ACTIVE"] / n / 8 / (dur[k] / n) / 1e3
            fh.write(f"{k:70s} us={dur[k] / n:8.1f} MFMA_busy={util:5.2f} clk={clk:4.2f}GHz WAIT_ANY={v['SQ_WAIT_ANY'] / wcyc:4.2f} "
                     f"WAIT_INST={v['SQ_WAIT_INST_ANY'] / wcyc:4.2f} ACTIVE={v['SQ_ACTIVE_INST_ANY'] / wcyc:4.2f} "
                     f"LDS_bank_conflict/LDS_active={v['SQ_LDS_BANK_CONFLICT'] / max(v['SQ_ACTIVE_INST_LDS'], 1):4.2f}\n")
    for sfx, fn in (("f/f_counter_collection.csv", "pmc_FETCH_SIZE.csv"), ("w/w_counter_collection.csv", "pmc_WRITE_SIZE.csv")):
        shutil.copy(base + sfx, os.path.join(out, f"{tag}_{fn}"))


if __name__ == "__main__":
    main(*sys.argv[1:4])
