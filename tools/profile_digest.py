#!/usr/bin/env python3
"""dev tool: copy / condense what tools/profile_round.sh wrote under gpurun_out/<dir> into the tracked profiles/rNN_* files.
    python tools/profile_digest.py gpurun_out/r02a r02"""
import csv
import glob
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(ROOT, "profiles")


def line(path):
    ls = [ln for ln in open(path) if ln.startswith("{")]
    return json.loads(ls[-1]) if ls else None


def copy(name, to):
    p = os.path.join(src, name)
    if os.path.exists(p):
        shutil.copyfile(p, os.path.join(dst, f"{tag}_{to}"))


def full_launch_stats(trace_dir, needle):
    """durations (ms) of the full-size launches of the kernels whose name holds `needle`, from the kernel trace"""
    f = glob.glob(os.path.join(src, trace_dir, "*kernel_trace.csv"))
    if not f:
        return None
    rows = [r for r in csv.DictReader(open(f[0])) if needle in r["Kernel_Name"]]
    if not rows:
        return None
    gmax = max(int(r["Grid_Size_X"]) for r in rows)
    d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows if int(r["Grid_Size_X"]) == gmax)
    # (the parity launch behind the timed steps -- 2e5 reads -- fills the same grid as a full-size step: told apart by duration.
    # Round 4's digest averaged it in: "n=56, avg 2.369 ms, min 0.113")
    d = [x for x in d if x >= 0.5 * d[len(d) // 2]]
    return {"n": len(d), "avg_ms": sum(d) / len(d), "median_ms": d[len(d) // 2], "min_ms": d[0], "max_ms": d[-1],
            "kernel": rows[0]["Kernel_Name"].split("(")[0]}


for name in ("bench_default", "bench_under_trace", "bench_sustained3000", "bench_k21", "bench_k63", "bench_hash", "bench_hist20",
             "bench_hist20_rccl1", "bench_packed", "bench_k21_under_trace", "bench_k63_under_trace", "bench_hash_under_trace",
             "bench_hist20_under_trace", "bench_2ranks_shared_gpu", "bench_2ranks_shared_gpu_hist20"):
    d = line(os.path.join(src, name + ".json")) if os.path.exists(os.path.join(src, name + ".json")) else None
    if d is not None:
        with open(os.path.join(dst, f"{tag}_{name}.json"), "w") as f:
            f.write(json.dumps(d) + "\n")
for a, b in (("trace/t_kernel_stats.csv", "kernel_stats.csv"), ("trace_k21/t_kernel_stats.csv", "kernel_stats_k21.csv"),
             ("trace_k63/t_kernel_stats.csv", "kernel_stats_k63.csv"), ("trace_hash/t_kernel_stats.csv", "kernel_stats_hash.csv"),
             ("trace_hist20/t_kernel_stats.csv", "kernel_stats_hist20.csv"), ("pmc_summary.txt", "pmc_summary.txt"),
             ("k_sweep.txt", "k_sweep.txt"), ("len_sweep.txt", "len_sweep.txt"), ("ragged_bench.txt", "ragged_bench.txt"), ("ragged2_bench.txt", "ragged2_bench.txt"),
             ("dirty_bench.txt", "dirty_bench.txt"), ("windows_bench.txt", "windows_bench.txt"), ("windows_len.txt", "windows_len.txt"), ("hist_bench.txt", "hist_bench.txt"),
             ("minimizers_bench.txt", "minimizers_bench.txt"), ("windows2_bench.txt", "windows2_bench.txt"), ("k2_long.txt", "k2_long.txt"), ("settle.txt", "settle.txt"), ("fastx_bench.txt", "fastx_bench.txt"), ("fastq_pipeline.txt", "fastq_pipeline.txt"), ("step_times_cold.txt", "step_times_cold.txt"), ("small_batches.txt", "small_batches.txt")):
    copy(a, b)

out = [f"# Round {tag[1:]} -- rocprofv3 --kernel-trace --stats of `python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --sustain-steps 0` (MI355X, 1 GPU)",
       "", "Collected by `tools/profile_round.sh` on the committed build, condensed by `tools/profile_digest.py`.",
       f"Raw rocprofv3 stats tables: `{tag}_kernel_stats*.csv` (their averages mix the full-size launches with the small parity-check launch).", ""]
for trace_dir, needle, label, bench in (("trace", "scan_bitsliced_kernel<31, 10, 4, false, false", "k=31 (the metric, BASELINE configs[1])", "bench_under_trace"),
                                        ("trace_k21", "scan_bitsliced_kernel<21, 10, 5, false, false", "k=21 (configs[2])", "bench_k21_under_trace"),
                                        ("trace_k63", "scan_bitsliced_kernel<63, 10, 3, false, false", "k=63 (configs[2], [u64;2])", "bench_k63_under_trace"),
                                        ("trace_hash", "scan_bitsliced_kernel<31, 10, 4, false, false", "k=31 + LexHasher fold, 1.25e8 reads (configs[3])", "bench_hash_under_trace")):
    st = full_launch_stats(trace_dir, needle)
    b = line(os.path.join(src, bench + ".json")) if os.path.exists(os.path.join(src, bench + ".json")) else None
    if st and b:
        r = b["roofline"]
        out.append(f"* **{label}** `{st['kernel']}`: n={st['n']} full-size launches, avg {st['avg_ms']:.3f} ms, median {st['median_ms']:.3f}, "
                   f"min {st['min_ms']:.3f}, max {st['max_ms']:.3f} (kernel-trace timestamps, warm-ups included); same run, HIP events in bench.py over "
                   f"the 20 timed steps: avg {r['avg_kernel_ms']:.3f} ms (scan + the idle sweep) -> {r['achieved']:.0f} GB/s algorithmic = "
                   f"**{100 * r['frac']:.1f} %** of 8 TB/s, {100 * r['frac_of_same_run_stream_read']:.1f} % of the same-run read stream ({r['same_run_stream_read_GBps']:.0f} GB/s).")
h = full_launch_stats("trace_hist20", "SinkHistPart")
if h:
    out.append(f"* **histogram 2^20 (configs[4])**: partition pass `{h['kernel'][:60]}...` avg {h['avg_ms']:.3f} ms per chunk launch (n={h['n']}); "
               f"per-step totals in `{tag}_bench_hist20.json`.")
d = line(os.path.join(src, "bench_default.json"))
if d:
    r = d["roofline"]
    out += ["", f"Un-profiled driver command (`{tag}_bench_default.json`): frac **{r['frac']:.3f}**, avg {r['avg_kernel_ms']:.3f} ms; "
            f"`roofline.traffic` = {r['traffic'] / 1e9:.2f} GB per step measured in the run (FETCH_SIZE x {r['traffic_detail']['read_correction']:.4f} "
            f"from the calibration kernel in the same pass + WRITE_SIZE) against 15.00 GB algorithmic; sustained "
            f"{d['sustained']['steps']} steps: {d['sustained']['ms_per_step']:.3f} ms = {d['sustained']['frac']:.3f}."]
s = line(os.path.join(src, "bench_sustained3000.json"))
if s:
    out.append(f"3000 back-to-back steps (`{tag}_bench_sustained3000.json`): {s['sustained']['ms_per_step']:.3f} ms per step = **{s['sustained']['frac']:.3f}** of the roofline.")
out += ["", f"Counter passes (`{tag}_pmc_summary.txt`, separate `--pmc` runs, no tracing flags): see DESIGN.md 4.1 / profiles/HISTORY.md for the digest."]
with open(os.path.join(dst, f"{tag}_kernel_trace_summary.md"), "w") as f:
    f.write("\n".join(out) + "\n")
print("\n".join(out))
