#!/bin/bash
# Raw PMC counters of the kernels whose name matches a pattern, one rocprofv3 pass per counter set given as "A B C" strings:
# usage (inside gpurun): bash tools/pmc_raw.sh <tag> <sql-like-pattern> "<set1>" "<set2>" ... -- <command...>
set -e
TAG=$1; PAT=$2; shift 2
SETS=()
while [ "$1" != "--" ]; do SETS+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for C in "${SETS[@]}"; do
  D=gpurun_out/$TAG/$(echo $C | tr ' ' '_')
  mkdir -p $D
  rocprofv3 --pmc $C --kernel-trace -d $D -o pmc -- "$@" > $D/log.txt 2>&1 || true
  python - "$D" "$PAT" <<'PY'
import sqlite3, glob, sys
d, pat = sys.argv[1], sys.argv[2]
dbs = glob.glob(d + "/*results.db")
if not dbs:
    print(d, "no db"); print(open(d + "/log.txt").read()[-600:]); sys.exit(0)
cur = sqlite3.connect(dbs[0]).cursor()
try:
    q = """select s.kernel_name, c.name, count(*), sum(p.value) from rocpd_pmc_event p
           join rocpd_kernel_dispatch d on p.event_id = d.event_id join rocpd_info_kernel_symbol s on d.kernel_id = s.id
           join rocpd_info_pmc c on p.pmc_id = c.id where s.kernel_name like ? group by 1, 2"""
    nd = {r[0]: r[1] for r in cur.execute("""select s.kernel_name, count(*) from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s
                                             on d.kernel_id = s.id where s.kernel_name like ? group by 1""", (pat,))}
    for r in cur.execute(q, (pat,)):
        print(f"{r[1]:34s} {r[0][:60]:60s} per-dispatch={r[3] / max(1, nd.get(r[0], 1)):.5g}")
except Exception as e:
    print(d, "query failed:", e)
PY
done
rm -rf gpurun_out/$TAG
