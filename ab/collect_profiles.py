"""gpurun_out/prof_TAG (ab/profile.sh) -> profiles/TAG_{bench.json,kernel_stats.csv,pmc.csv} + profiles/traffic.json"""
import csv, glob, json, os, sys
tag = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "fused_train16_kernel"     # the dominant kernel of the default bench (substring of its name)
# argv[3]: the key of profiles/issue.json = bench.py's kernel_name() of the timed configuration (default: the substring)
update_traffic = kern.startswith("fused_train16")                       # profiles/traffic.json is the headline kernel's
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
line = [l for l in open(os.path.join(src, "bench.json")) if l.startswith("{")][-1]
open(os.path.join(dst, f"{tag}_bench.json"), "w").write(line)
newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)           # a directory may hold an earlier run's files too
stats = newest(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))
open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w").write(open(stats).read())
out = []
vals = {}
for i in range(1, 6):
    f = newest(os.path.join(src, f"pmc{i}", "*", "*_counter_collection.csv"))
    rows = [r for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"]]
    disp = sorted({int(r["Dispatch_Id"]) for r in rows})
    dur = {}
    for r in rows:
        dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    names = sorted({r["Counter_Name"] for r in rows})
    sets = open(os.path.join(src, f"pmc{i}.set")).read().strip()
    extra = open(os.path.join(src, "args.txt")).read().strip() if os.path.exists(os.path.join(src, "args.txt")) else ""
    out.append(f"# pass {i}: rocprofv3 --pmc {sets} --kernel-trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --stat-launches 0 {extra}   "
               f"({kern} avg {sum(dur.values()) / len(dur):.3f} ms over {len(disp)} launches)")
    for n in names:
        per = {}
        for r in rows:
            if r["Counter_Name"] == n:
                per[int(r["Dispatch_Id"])] = per.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
        v = sum(per.values()) / len(per)
        vals[n] = v
        out.append(f"{n},{v:.6g}")
open(os.path.join(dst, f"{tag}_pmc.csv"), "w").write("\n".join(out) + "\n")
traffic = {"fused_kernel_bytes_per_launch": int(2 * vals["FETCH_SIZE"] * 1024 + vals["WRITE_SIZE"] * 1024),
           "raw": {"FETCH_SIZE_KB": vals["FETCH_SIZE"], "WRITE_SIZE_KB": vals["WRITE_SIZE"], "TCC_EA0_ATOMIC_sum": vals["TCC_EA0_ATOMIC_sum"]},
           "note": f"per launch of {kern} (Layout<1>, MODE_TRAIN_MSE) on the 4K workload; separate --pmc passes (profiles/{tag}_pmc.csv); "
                   "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of coalesced fetch bytes; the kernel's 4-12 B/lane reads are "
                   "outside the calibrated pattern, so this is an upper bound), WRITE_SIZE as is (exact for float atomics)."}
traffic["note"] = traffic["note"].replace("fused_train16_kernel (Layout<1>, MODE_TRAIN_MSE)", kern)
json.dump(traffic, open(os.path.join(dst, "traffic.json" if update_traffic else f"{tag}_traffic.json"), "w"), indent=1)
# what binds the kernel: per-SIMD busy fractions (GRBM_GUI_ACTIVE sums the 8 XCDs' active cycles; 256 CUs x 4 SIMDs; a wave instruction holds the vector ALU 4 cycles)
simd_cycles = vals["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
issue = {"valu_busy": round(4.0 * vals["SQ_ACTIVE_INST_VALU"] / simd_cycles, 4), "mfma_busy": round(vals["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles, 4),
         "wait": round(vals["SQ_WAIT_ANY"] / vals["SQ_WAVE_CYCLES"], 4), "lds_conflict": round(vals["SQ_LDS_BANK_CONFLICT"] / vals["SQ_LDS_IDX_ACTIVE"], 4),
         "insts_valu_per_mfma": round(vals["SQ_INSTS_VALU"] / max(vals["SQ_INSTS_MFMA"], 1.0), 2), "waves": int(vals["SQ_WAVES"]), "pmc": f"profiles/{tag}_pmc.csv",
         "hbm_bytes_per_launch": traffic["fused_kernel_bytes_per_launch"]}
ip = os.path.join(dst, "issue.json")
allrec = json.load(open(ip)) if os.path.exists(ip) else {}
kname = [l for l in open(stats).read().split("\n") if kern in l]
allrec[sys.argv[3] if len(sys.argv) > 3 else kern] = issue
json.dump(allrec, open(ip, "w"), indent=1, sort_keys=True)
print(line[:300]); print("\n".join(out[:12])); print(traffic["fused_kernel_bytes_per_launch"]); print(issue)
