#!/bin/bash
# Derived PMC metrics (MfmaUtil, LdsUtil, LdsBankConflict, MemUnitStalled, VALUBusy ...) of the kernels whose name matches a
# pattern, one rocprofv3 pass per counter set (--kernel-trace only, as the pool requires):
# usage (inside gpurun): bash tools/pmc_kernel.sh <tag> <sql-like-pattern> <command...>
#   e.g. bash tools/pmc_kernel.sh pmca '%attn_prefill%' python tools/attn_ab.py
set -e
TAG=$1; PAT=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for C in "MfmaUtil" "LdsUtil" "LdsBankConflict" "MemUnitStalled" "VALUBusy" "SALUBusy" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"; do
  D=gpurun_out/$TAG/$(echo $C | tr ' ' '_')
  mkdir -p $D
  rocprofv3 --pmc $C --kernel-trace -d $D -o pmc -- "$@" > $D/log.txt 2>&1 || true
  python - "$D" "$PAT" <<'PY'
import sqlite3, glob, sys
d, pat = sys.argv[1], sys.argv[2]
dbs = glob.glob(d + "/*results.db")
if not dbs:
    print(d, "no db"); sys.exit(0)
cur = sqlite3.connect(dbs[0]).cursor()
try:
    q = """select s.kernel_name, c.name, count(*), avg(p.value) from rocpd_pmc_event p
           join rocpd_kernel_dispatch d on p.event_id = d.event_id join rocpd_info_kernel_symbol s on d.kernel_id = s.id
           join rocpd_info_pmc c on p.pmc_id = c.id where s.kernel_name like ? group by 1, 2"""
    for r in cur.execute(q, (pat,)):
        print(f"{r[1]:22s} {r[0][:70]:70s} n={r[2]:5d} avg={r[3]:.4g}")
except Exception as e:
    print(d, "query failed:", e)
PY
done
rm -rf gpurun_out/$TAG
