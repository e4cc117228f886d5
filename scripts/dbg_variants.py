"""Debug (GPU box): run one golden case per kernel variant under a watchdog."""
import os, sys, subprocess
CASES = [("deep96_R32_m16", 2, "2", st) for st in ("110", "111", "112", "113", "114")]
if len(sys.argv) > 1:
    name, ci, kind = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    sys.path.insert(0, ".")
    import numpy as np
    from tests.conftest import load_golden
    from diskrag_amd import HipIndex, _ffi
    g = load_golden(name); c = g.case(ci)
    mode = {"M1": 1, "M2": 2, "M3": 3, "M4": 4}[c["mode"]]
    ix = HipIndex.create(g.vectors, g.adj if mode <= 2 else g.mem_adj, g.medoid)
    if g.m: ix.set_pq(g.codebook, g.codes)
    flags = 1 if (c["mode"] == "M3" and c["use_pq"]) else 0
    ids, dist, cnt, st = ix.search_batch(c["queries"], c["k"], L=c.get("L", 100), beam_width=c.get("bw", 0) or 0, mode=mode, band_policy=c.get("policy", 0), flags=flags)
    print(name, c["mode"], "kind", kind or "auto", "ids_ok", bool(np.array_equal(ids, c["ids"])), "status", int(st["status"].max()), ix.timing())
else:
    for name, ci, kind, stage in CASES:
        env = dict(os.environ); env["DR_DEBUG"] = "1"; env["DR_DEBUG_STAGE"] = stage; print("== stage", stage, flush=True)
        if kind: env["DR_FORCE_KIND"] = kind
        try:
            errf = open(f'/tmp/dbg_{name}_{kind}.err', 'w')
            try:
                r = subprocess.run([sys.executable, __file__, name, str(ci), kind or "-"], env=env, timeout=15, stdout=subprocess.PIPE, stderr=errf, text=True)
                print(r.stdout.strip()[-400:], flush=True)
            finally:
                errf.close(); print(open(errf.name).read()[-600:], flush=True)
        except subprocess.TimeoutExpired:
            print("TIMEOUT", name, ci, kind, flush=True)
        except Exception as ex:
            print("EXC", ex)
