import os, sys, subprocess
for lib in ("diskrag_amd/libdr_dbgB.so",):
    for name, ci, kind in (("deep96_R32_m16", 2, "2"), ("unit1536_R16_m32", 0, "")):
        env = dict(os.environ); env["DR_LIB"] = os.path.abspath(lib); env["DR_DEBUG"] = "1"
        if kind: env["DR_FORCE_KIND"] = kind
        print("==", lib, name, kind, flush=True)
        try:
            r = subprocess.run([sys.executable, "scripts/dbg_variants.py", name, str(ci), kind or "-"], env=env, timeout=15, capture_output=True, text=True)
            print(r.stdout.strip()[-300:], r.stderr.strip()[-300:], flush=True)
        except subprocess.TimeoutExpired:
            print("TIMEOUT", flush=True)
