#!/bin/bash
# One profile of the headline decode step that bench.py's `roofline` object audits itself against:
#   pass A  rocprofv3 --kernel-trace --stats of a 512-step graph decode  -> per-kernel in-graph average durations
#   pass B  rocprofv3 --pmc FETCH_SIZE (its own pass, MI355X_MICROARCH.md "HBM": x1024 x2 on gfx950)
#   pass C  rocprofv3 --pmc WRITE_SIZE (uncalibrated; reported raw)
# Everything lands in gpurun_out/<tag>_roofline_profile.json, stamped with the sha256 of the kernel sources + build flags
# (unimedvl_amd/lib/build.stamp): bench.py uses the file only while that stamp equals the library it is running.
# usage (inside gpurun): bash tools/roofline_profile.sh <tag>     then, in the build container:
#   cp gpurun_out/<tag>_roofline_profile.json profiles/ && cp gpurun_out/<tag>_roofline_profile.json profiles/roofline_profile_latest.json
set -e
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS="--no-cpu-baseline --no-t2i --no-fp8 --no-report --no-vit --no-load-path --no-sampled"
mkdir -p gpurun_out/$TAG/trace gpurun_out/$TAG/FETCH_SIZE gpurun_out/$TAG/WRITE_SIZE
rocprofv3 --kernel-trace --stats -d gpurun_out/$TAG/trace -o bench -- python bench.py $ARGS --steps 512 --warmup 8 > gpurun_out/$TAG/trace/bench.log 2>&1 || true
grep '^{' gpurun_out/$TAG/trace/bench.log > gpurun_out/${TAG}_decode_line_under_rocprof.json || true
DB=$(ls gpurun_out/$TAG/trace/*results.db 2>/dev/null | head -1)
python tools/rocpd_stats.py "$DB" gpurun_out/${TAG}_decode_kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace -d gpurun_out/$TAG/$C -o pmc -- python bench.py $ARGS --steps 4 --warmup 1 > gpurun_out/$TAG/$C/bench.log 2>&1 || true
done
# MFMA-bound legs: counters of the dominant kernels of the ViT tower, the image-span prefill and the text-to-image leg
# (one rocprofv3 pass per counter set, --kernel-trace only), attached by bench.py to its vit_encode / t2i objects
if [ -z "$SKIP_STAGE_PMC" ]; then
# per-leg kernel trace (no counters): average duration of every kernel INSIDE the leg, one row per (kernel, grid) = per GEMM shape
for ST in vit prefill t2i edit; do
  D=gpurun_out/$TAG/trace_${ST}
  mkdir -p $D
  REPS=5 STEPS=12 rocprofv3 --kernel-trace --stats -d $D -o stage -- python tools/stage_profile.py $ST > $D/log.txt 2>&1 || true
  DBS=$(ls $D/*results.db 2>/dev/null | head -1)
  [ -n "$DBS" ] && BY_GRID=1 python tools/rocpd_stats.py "$DBS" gpurun_out/${TAG}_${ST}_kernel_stats_by_grid.csv
done
for ST in vit prefill t2i edit; do
  for C in "MfmaUtil" "LdsUtil" "SQ_WAIT_INST_LDS SQ_BUSY_CYCLES"; do
    D=gpurun_out/$TAG/stage_${ST}_$(echo $C | tr ' ' '_')
    mkdir -p $D
    REPS=3 STEPS=12 rocprofv3 --pmc $C --kernel-trace -d $D -o pmc -- python tools/stage_profile.py $ST > $D/log.txt 2>&1 || true
  done
done
fi
python - <<PY
import csv, glob, json, sqlite3
tag = "$TAG"
stage_pmc = {}
for st in ("vit", "prefill", "t2i", "edit"):
    rows = {}
    for d in glob.glob(f"gpurun_out/{tag}/stage_{st}_*"):
        dbs = glob.glob(d + "/*results.db")
        if not dbs:
            continue
        cur = sqlite3.connect(dbs[0]).cursor()
        q = """select s.kernel_name, c.name, count(*), avg(p.value) from rocpd_pmc_event p
               join rocpd_kernel_dispatch d on p.event_id = d.event_id join rocpd_info_kernel_symbol s on d.kernel_id = s.id
               join rocpd_info_pmc c on p.pmc_id = c.id where s.kernel_name like '%gemm_tiled%' or s.kernel_name like '%gemm_w4%' or s.kernel_name like '%attn_prefill%'
               group by 1, 2"""
        try:
            for name, cname, n, avg in cur.execute(q):
                rows.setdefault(name, {})[cname] = dict(n=n, avg=avg)
        except Exception as e:
            rows["error"] = str(e)
    stage_pmc[st] = rows
stage_kernels = {}
for st in ("vit", "prefill", "t2i", "edit"):
    try:
        rows = list(csv.DictReader(open(f"gpurun_out/{tag}_{st}_kernel_stats_by_grid.csv")))
    except OSError:
        continue
    stage_kernels[st] = [dict(name=r["Name"], grid=int(r["GridX"]), calls=int(r["Calls"]), avg_us=float(r["AverageNs"]) / 1e3,
                              total_ms=int(r["TotalDurationNs"]) / 1e6) for r in rows[:24]]
stamp = open("unimedvl_amd/lib/build.stamp").read().strip()
kern = {}
for r in csv.DictReader(open(f"gpurun_out/{tag}_decode_kernel_stats.csv")):
    kern[r["Name"]] = dict(calls=int(r["Calls"]), avg_us=float(r["AverageNs"]) / 1e3, min_us=int(r["MinNs"]) / 1e3, share_pct=float(r["Percentage"]))
pmc = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    dbs = glob.glob(f"gpurun_out/{tag}/{c}/*results.db")
    if not dbs:
        continue
    cur = sqlite3.connect(dbs[0]).cursor()
    q = """select s.kernel_name, count(*), avg(p.value) from rocpd_pmc_event p join rocpd_kernel_dispatch d on p.event_id = d.event_id
           join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name order by 3 desc"""
    try:
        for name, n, avg in cur.execute(q):
            pmc.setdefault(name, {})[c] = dict(dispatches=n, avg_kib=avg)
    except Exception as e:
        pmc["error_" + c] = str(e)
line = {}
try:
    line = json.loads(open(f"gpurun_out/{tag}_decode_line_under_rocprof.json").read().strip().splitlines()[-1])
except Exception:
    pass
out = dict(code_stamp=stamp, tag=tag, kernels=kern, pmc=pmc, stage_pmc=stage_pmc, stage_kernels=stage_kernels,
           bench_line_under_rocprof={k: line.get(k) for k in ("value", "ms_per_step", "steps", "config")},
           correction="traffic = FETCH_SIZE (KiB) x 1024 x 2: gfx950's rocprofv3 tallies the 128-byte requests of a 16 B/lane coalesced stream at 64 B "
                      "(MI355X_MICROARCH.md 'HBM'); WRITE_SIZE is uncalibrated and reported raw (KiB)",
           commands=["rocprofv3 --kernel-trace --stats -- python bench.py %s --steps 512 --warmup 8" % "$ARGS",
                     "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python bench.py %s --steps 4 --warmup 1" % "$ARGS",
                     "rocprofv3 --pmc WRITE_SIZE --kernel-trace -- python bench.py %s --steps 4 --warmup 1" % "$ARGS"])
json.dump(out, open(f"gpurun_out/{tag}_roofline_profile.json", "w"), indent=1)
top = sorted(kern.items(), key=lambda kv: -kv[1]["share_pct"])[:10]
for n, v in top:
    f = pmc.get(n, {}).get("FETCH_SIZE", {}).get("avg_kib")
    print(f"{n[:70]:70s} n={v['calls']:6d} avg={v['avg_us']:8.2f}us {v['share_pct']:6.2f}%  fetch x2 = {(f * 2048 / 1e6 if f else float('nan')):8.2f} MB")
print("stamp", stamp[:16], "line", line.get("value"), line.get("ms_per_step"))
PY
rm -rf gpurun_out/$TAG
