#!/bin/bash
# Derived PMC metrics of the tiled GEMM on one shape (separate rocprofv3 passes, --kernel-trace only):
# usage (inside gpurun): SHAPE=8192,8192,8192 bash tools/pmc_gemm.sh <tag> [gemm_bench args]
set -e
TAG=${1:-pmcg}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for C in "MfmaUtil" "LdsBankConflict" "LdsUtil" "MemUnitStalled" "SQ_INSTS_LDS SQ_INSTS_MFMA" "SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY"; do
  D=gpurun_out/$TAG/$(echo $C | tr ' ' '_')
  mkdir -p $D
  rocprofv3 --pmc $C --kernel-trace -d $D -o pmc -- python tools/gemm_bench.py "$@" > $D/log.txt 2>&1 || true
  python - "$D" "$C" <<'PY'
import sqlite3, glob, sys
d, names = sys.argv[1], sys.argv[2].split()
dbs = glob.glob(d + "/*results.db")
if not dbs:
    print(names, "no db"); sys.exit(0)
cur = sqlite3.connect(dbs[0]).cursor()
try:
    q = """select s.kernel_name, c.name, count(*), avg(p.value) from rocpd_pmc_event p
           join rocpd_kernel_dispatch d on p.event_id = d.event_id join rocpd_info_kernel_symbol s on d.kernel_id = s.id
           join rocpd_info_pmc c on p.pmc_id = c.id where s.kernel_name like '%gemm_tiled%' group by 1, 2"""
    for r in cur.execute(q):
        print(f"{r[1]:24s} {r[0][:60]:60s} n={r[2]:4d} avg={r[3]:.4g}")
except Exception as e:
    print(names, "query failed:", e)
PY
done
rm -rf gpurun_out/$TAG
