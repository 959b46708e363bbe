"""ISA statistics for the device code of libpco_gfx: python scripts/isa_stats.py [name-substring [--dump]]
Compiles pco_gfx.hip to gfx950 assembly (device only) and counts instructions per function / kernel."""
import os, re, subprocess, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = "/tmp/pco_isa.s"
def build():
    src = os.path.join(ROOT, "pcodec_amd", "csrc", "pco_gfx.hip")
    deps = [os.path.join(ROOT, "pcodec_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "pcodec_amd", "csrc"))]
    if os.path.exists(OUT) and all(os.path.getmtime(d) < os.path.getmtime(OUT) for d in deps):
        return
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
                           "-Wno-unused-result", "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S", src, "-o", OUT], stderr=subprocess.DEVNULL)
def functions():
    name = None; body = []
    for line in open(OUT):
        m = re.match(r"^(_Z\S+):\s", line)
        if m and name is None:
            name = m.group(1); body = []; continue
        if name is not None:
            if line.startswith(".Lfunc_end"):
                yield name, body; name = None
            else:
                body.append(line)
def demangle(n):
    try: return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip()
    except Exception: return n
if __name__ == "__main__":
    build()
    pat = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else ""
    dump = "--dump" in sys.argv
    for name, body in functions():
        d = demangle(name)
        short = d.split("(")[0]
        if pat and pat not in d: continue
        ins = [l.split()[0] for l in body if l.startswith("\t") and not l.strip().startswith((".", ";"))]
        c = collections.Counter()
        for i in ins:
            if i.startswith("v_"): c["valu"] += 1
            elif i.startswith("s_cbranch") or i.startswith("s_branch"): c["branch"] += 1
            elif i.startswith("s_waitcnt"): c["wait"] += 1
            elif i.startswith("s_"): c["salu"] += 1
            elif i.startswith("global_") or i.startswith("flat_") or i.startswith("buffer_"): c["vmem"] += 1
            elif i.startswith("scratch_"): c["scratch"] += 1
            elif i.startswith("ds_"): c["lds"] += 1
            else: c["other"] += 1
        print(f"{short[:70]:70s} total {len(ins):6d} " + " ".join(f"{k}={v}" for k, v in sorted(c.items())))
        if dump:
            sys.stdout.write("".join(body))
