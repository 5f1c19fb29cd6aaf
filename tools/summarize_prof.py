"""Condense a tools/profile_round.sh / profile_config.sh output directory into one small text file for profiles/.
   python tools/summarize_prof.py <src dir> <dst dir> <tag>
Sections: the rocprofv3 --kernel-trace --stats table, then per trace kernel (one-off counting builds skipped) its launch
resources and the per-dispatch mean of every PMC counter (each pass was its own run, never combined with tracing)."""
import collections, csv, glob, os, re, sys
src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
os.makedirs(dst, exist_ok=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from example_gui_opencl_raytracer_amd import api
# which kernels were profiled: bench.py only quotes a summary whose hash names the kernels of the library it has loaded
out = [f"kernel_source_sha256_16: {api.kernel_source_hash()}\nlibrary: {api.library_version()}\n\n"]
for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True):
    out.append(f"## rocprofv3 --kernel-trace --stats ({os.path.basename(f)})\n")
    out.append("".join(l for l in open(f) if "at::native" not in l))
pm = collections.OrderedDict()
info = {}
for f in sorted(glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        m = re.search(r"wt_trace<(\d+)>", k)
        if not m or int(m.group(1)) & 1:                       # skip the one-off counting build (flag bit 0)
            continue
        pm.setdefault(k, collections.OrderedDict()).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        info[k] = {x: r[x] for x in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count")}
out.append("\n## rocprofv3 --pmc (separate passes), per-dispatch mean over the timed launches\n")
for k, counters in pm.items():
    out.append(f"\n### kernel {k}\n{info[k]}\ncounter,mean_per_dispatch,dispatches\n")
    for c, v in counters.items():
        out.append(f"{c},{sum(v) / len(v):.1f},{len(v)}\n")
open(os.path.join(dst, f"{tag}_rocprof_summary.md"), "w").write("".join(out))
print("".join(out))
