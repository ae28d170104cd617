"""Copies what tools/gpu_round_refresh.sh left under gpurun_out/ into profiles/ (build container).
usage: collect_round_profiles.py <round-tag> <prefix>      e.g.  r02e r02"""
import json, os, shutil, sys

tag, prefix = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "gpurun_out", "refresh_" + tag)
prof = os.path.join(root, "gpurun_out", "prof_" + tag)
ctr = os.path.join(root, "gpurun_out", "ctr_" + tag)
dst = os.path.join(root, "profiles")


def last_line(src, name):
    with open(src) as f:
        line = f.read().strip().splitlines()[-1]
    json.loads(line)
    with open(os.path.join(dst, name), "w") as f:
        f.write(line + "\n")


last_line(os.path.join(out, "bench.json"), prefix + "_bench.json")
for n in ("config5", "config2", "members8", "rodaspr"):
    try:
        last_line(os.path.join(out, "bench_%s.json" % n), "%s_bench_%s.json" % (prefix, n))
    except (OSError, ValueError, IndexError) as ex:
        print("no %s line: %s" % (n, ex))
shutil.copy(os.path.join(prof, "kernel_stats.csv"), os.path.join(dst, prefix + "_bench_kernel_stats.csv"))
shutil.copy(os.path.join(prof, "pmc_summary.json"), os.path.join(dst, prefix + "_bench_pmc_summary.json"))
shutil.copy(os.path.join(prof, "bench_under_rocprof.json"), os.path.join(dst, prefix + "_bench_under_rocprof.json"))
shutil.copy(os.path.join(ctr, "sq_summary.json"), os.path.join(dst, prefix + "_sq_counters.json"))
shutil.copy(os.path.join(out, "levels.txt"), os.path.join(dst, prefix + "_solver_levels_trace.txt"))
shutil.copy(os.path.join(out, "step_doubling_trial.txt"), os.path.join(dst, prefix + "_step_doubling_trial.txt"))
if os.path.exists(os.path.join(out, "stamps.txt")):
    shutil.copy(os.path.join(out, "stamps.txt"), os.path.join(dst, prefix + "_cr_factor_stamps.txt"))
p = json.load(open(os.path.join(prof, "pmc_summary.json")))
s = p["tfk_sweep_fj"]
t = json.load(open(os.path.join(dst, "sweep_traffic.json")))
t.update(FETCH_SIZE_KB=s["FETCH_SIZE_KB_avg"], WRITE_SIZE_KB=s["WRITE_SIZE_KB_avg"], round=prefix,
         hbm_bytes_per_launch=(2 * s["FETCH_SIZE_KB_avg"] + s["WRITE_SIZE_KB_avg"]) * 1024)
json.dump(t, open(os.path.join(dst, "sweep_traffic.json"), "w"), indent=1)
print("sweep traffic %.1f MB per launch" % (t["hbm_bytes_per_launch"] / 1e6))
for k, v in p.items():
    if k.startswith("tfk"):
        f, w = v["FETCH_SIZE_KB_avg"] or 0, v["WRITE_SIZE_KB_avg"] or 0
        print("%-24s read %7.1f MB  written %7.1f MB" % (k, 2 * f * 1024 / 1e6, w * 1024 / 1e6))
