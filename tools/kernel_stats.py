"""Register / LDS / scratch figures of the compiled trace kernels (from the gfx950 code object inside a build/*.o) and a
hash of that code object -- printed as JSON lines.  Usage: python tools/kernel_stats.py [fast|strict] [tag]"""
import hashlib, json, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def code_object(which="fast", tag=""):
    obj = os.path.join(ROOT, "example_gui_opencl_raytracer_amd", "build", f"whitted_{which}{tag}.o")
    tmp = tempfile.mkdtemp()
    # the host object carries the device code as a fat binary section; extract the bundle and unbundle it
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", obj], check=True)
    co = os.path.join(tmp, "gfx950.co")
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", f"--output={co}",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], check=True)
    return co


def stats(which="fast", tag=""):
    co = code_object(which, tag)
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    out = []
    for blk in notes.split("- .agpr_count")[1:]:
        g = lambda k: re.search(rf"\.{k}:\s*(\S+)", blk)
        name = g("name").group(1)
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        out.append(dict(kernel=dem, vgpr=int(g("vgpr_count").group(1)), vgpr_spill=int(g("vgpr_spill_count").group(1)),
                        sgpr=int(g("sgpr_count").group(1)), sgpr_spill=int(g("sgpr_spill_count").group(1)),
                        lds=int(g("group_segment_fixed_size").group(1)), scratch=int(g("private_segment_fixed_size").group(1))))
    return hashlib.sha256(open(co, "rb").read()).hexdigest()[:16], out


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "fast"
    h, ks = stats(which, sys.argv[2] if len(sys.argv) > 2 else "")
    print(json.dumps(dict(build=which, code_object_sha256_16=h)))
    for k in ks:
        if "wt_trace" in k["kernel"]:
            print(json.dumps(k))
