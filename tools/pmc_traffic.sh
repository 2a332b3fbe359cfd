#!/bin/bash
# HBM traffic of the dominant decode kernel from PMC counters (separate passes, MI355X_MICROARCH.md "HBM"):
#   pass 1: FETCH_SIZE, pass 2: WRITE_SIZE; per-dispatch values -> gpurun_out/<tag>_pmc.json
set -e
TAG=${1:-pmc}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for C in FETCH_SIZE WRITE_SIZE; do
  mkdir -p gpurun_out/$TAG/$C
  rocprofv3 --pmc $C --kernel-trace -d gpurun_out/$TAG/$C -o pmc -- python bench.py --no-cpu-baseline --no-t2i --no-fp8 --no-report --steps 4 --warmup 1 > gpurun_out/$TAG/$C/bench.log 2>&1 || true
done
python - <<PY
import sqlite3, glob, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    dbs = glob.glob(f"gpurun_out/$TAG/{c}/*results.db")
    if not dbs:
        out[c] = None; continue
    db = sqlite3.connect(dbs[0]); cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    q = """select s.kernel_name, count(*), avg(p.value), min(p.value), max(p.value)
           from rocpd_pmc_event p join rocpd_kernel_dispatch d on p.event_id = d.event_id
           join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name order by 3 desc"""
    try:
        rows = list(cur.execute(q))
    except Exception as e:
        rows = []; out[c + "_error"] = str(e); out[c + "_tables"] = [t for t in tabs if "pmc" in t]
    out[c] = [dict(kernel=r[0], dispatches=r[1], avg=r[2], min=r[3], max=r[4]) for r in rows[:12]]
json.dump(out, open("gpurun_out/${TAG}_pmc.json", "w"), indent=1)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for r in (out.get(c) or [])[:6]:
        print(c, r["kernel"][:70], r["dispatches"], round(r["avg"], 1))
print({k: v for k, v in out.items() if k.endswith("error") or k.endswith("tables")})
PY
rm -rf gpurun_out/$TAG
