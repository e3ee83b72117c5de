"""Builds profiles/r<NN>_pmc_traffic.json (usage: python dev/pmc_traffic.py r02) from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over dev/pmc_kernels.py
(directories gpurun_out/pmc_tr_<mode>_<rows>_<counter>), see the "_how" entry.
usage: python dev/pmc_traffic.py"""
import csv, glob, json, os, collections, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d, l = 512, 74
KIND = {"k_xp3": "K1", "k_xp_pers": "K1", "k_xp_mfma": "K1", "k_atb3": "K2", "k_atb_mfma": "K2", "k_pow3": "K3"}
ALGO = {"K1": lambda n: 4 * (n * d + n * l + d * l), "K2": lambda n: 4 * (n * d + n * l + d * l), "K3": lambda n: 4 * (n * d + 2 * d * l)}
out = {"_how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (with --kernel-trace only) on "
               "dev/pmc_kernels.py (dev/pmc_pass.sh), per GEMM mode; counter values are KB per launch; gfx950 correction "
               "(MI355X_MICROARCH.md, HBM): FETCH_SIZE reports 1/2 of wide coalesced streaming reads -- calibrated on "
               "k_sum_parts2 (10257.5 KB reported for a 20971520-byte read), WRITE_SIZE is exact. "
               "hbm_bytes_corrected = (2*FETCH_SIZE + WRITE_SIZE)*1024."}
for mode in ("bf16x3", "fp32"):
    for n in (100000, 1000000):
        vals = collections.defaultdict(dict)
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            for f in glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_tr_{mode}_{n}_{counter}", "**", "*counter_collection.csv"), recursive=True):
                acc = collections.defaultdict(list)
                for row in csv.DictReader(open(f, newline="")):
                    for key, kind in KIND.items():
                        if key in row["Kernel_Name"] and row["Counter_Name"] == counter:
                            acc[kind].append(float(row["Counter_Value"]))
                for kind, v in acc.items():
                    vals[kind][counter] = sum(v) / len(v)
        if not vals:
            continue
        ent = {}
        for kind, v in vals.items():
            if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
                ent[kind] = {"FETCH_SIZE_KB": round(v["FETCH_SIZE"], 2), "WRITE_SIZE_KB": round(v["WRITE_SIZE"], 2),
                             "hbm_bytes_corrected": (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024,
                             "algorithmic_bytes": ALGO[kind](n)}
        out.setdefault(f"{n}x{d} l={l}", {})[mode] = ent
rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
try:
    out["measured_at"] = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:
    out["measured_at"] = "unknown"
json.dump(out, open(os.path.join(ROOT, "profiles", f"{rnd}_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
