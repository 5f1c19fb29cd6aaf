"""Static instruction mix of one compiled trace kernel (disassembly of the gfx950 code object): python tools/kernel_isa.py fast 68 [tag]"""
import collections, os, re, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_stats import code_object, LLVM
which, flags = sys.argv[1], sys.argv[2]
args = [a for a in sys.argv[3:] if not a.startswith("--")]
co = code_object(which, args[0] if args else "")
dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True, check=True).stdout
m = re.search(r"<_ZN\w+8wt_traceILi%s\w*>:\n(.*?)\n\n" % flags, dis, re.S)
body = m.group(1).split("\n")
ops = collections.Counter()
for l in body:
    t = l.split()
    if t and not t[0].endswith(":"):
        ops[t[0]] += 1
tot = sum(ops.values())
cls = collections.Counter()
for o, n in ops.items():
    c = ("valu" if o.startswith("v_") else "salu" if o.startswith("s_") and not o.startswith(("s_load", "s_buffer", "s_waitcnt", "s_cbranch", "s_branch", "s_barrier")) else
         "smem" if o.startswith(("s_load", "s_buffer")) else "lds" if o.startswith("ds_") else "vmem" if o.startswith(("global_", "scratch_", "buffer_", "flat_")) else
         "branch" if o.startswith(("s_cbranch", "s_branch")) else "wait" if o.startswith("s_waitcnt") else "other")
    cls[c] += n
print(f"wt_{which}::wt_trace<{flags}>: {tot} instructions", dict(cls))
print(" top:", ops.most_common(25))
if "--dump" in sys.argv:
    open(f"/tmp/wt_{which}_{flags}.s", "w").write("\n".join(body))
