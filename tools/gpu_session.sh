#!/bin/bash
# ONE parameterised driver for what the round-by-round one-shot scripts (tools/r04_*.sh, r05_*.sh, r06_*.sh: in the git history) did
# on the GPU box.  Run through gpurun from the build container:
#     gpurun --timeout 2400 -- 'bash tools/gpu_session.sh <tag> <step> [<step> ...]'
# Everything a step prints lands in gpurun_out/<tag>_session/ (merged back by gpurun).  Steps:
#   suite                       the round-end GPU suite as the driver runs it (pytest -m gpu -x, 25 slowest tests) + __graft_entry__.smoke()
#   tests:<pytest args>         e.g. 'tests:tests/test_attn_lazy_gpu.py -k deterministic -s'
#   bench[:<flags>]             python bench.py <flags> -> <tag>/bench.txt
#   decode[:<flags>]            the decode leg only (256 steps): value / ms per step on one line
#   ab:<VAR=a,b,c>[:<flags>]    the decode leg once per value of an environment knob, e.g. 'ab:UMV_DECODE_NSPLIT=12,24'
#   env:<K=V ...>               export variables for the steps that follow, e.g. 'env:UMV_DECODE_WSPLIT=4 UMV_DECODE_NSPLIT=6'
#   profile                     tools/roofline_profile.sh <tag> (kernel trace + FETCH_SIZE / WRITE_SIZE passes + stage traces)
#   run:<command>               anything else, output to <tag>/run.txt
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp
TAG=${1:?tag}; shift
O=gpurun_out/${TAG}_session; mkdir -p $O       # (tools/roofline_profile.sh owns - and removes - gpurun_out/<tag>/)
DECODE_ONLY="--no-t2i --no-vit --no-vae --no-load-path --no-cpu-baseline --no-fp8 --no-report --no-edit --no-sampled --steps 256"
line() { python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print(d['value'], d['unit'], d['ms_per_step'], 'ms/step', 'step frac', d['roofline'].get('step_frac_of_peak'))"; }
for step in "$@"; do
  kind=${step%%:*}; arg=""; [ "$kind" != "$step" ] && arg=${step#*:}
  t0=$(date +%s)
  case $kind in
    suite)   timeout 2400 python -m pytest tests/ -x -q -m gpu --durations=25 2>&1 | tail -70 > $O/suite.txt
             python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -3 $O/suite.txt $O/smoke.txt ;;
    tests)   timeout 2400 python -m pytest $arg -q -m gpu 2>&1 | tail -60 >> $O/tests.txt; tail -3 $O/tests.txt ;;
    bench)   timeout 1200 python bench.py $arg > $O/bench.txt 2>&1; grep '^{' $O/bench.txt | line ;;
    decode)  timeout 600 python bench.py $DECODE_ONLY $arg 2>/dev/null | line | tee -a $O/decode.txt ;;
    ab)      var=${arg%%=*}; rest=${arg#*=}; vals=${rest%%:*}; flags=""; [ "$vals" != "$rest" ] && flags=${rest#*:}
             for v in ${vals//,/ }; do echo -n "$var=$v: "; env $var=$v timeout 600 python bench.py $DECODE_ONLY $flags 2>/dev/null | line; done | tee -a $O/ab.txt ;;
    env)     export $arg ;;
    profile) bash tools/roofline_profile.sh $TAG > $O/profile.txt 2>&1; tail -5 $O/profile.txt ;;
    run)     timeout 2400 bash -c "$arg" > $O/run.txt 2>&1; tail -20 $O/run.txt ;;
    *)       echo "unknown step $step"; exit 2 ;;
  esac
  echo "[$step] $(( $(date +%s) - t0 )) s" | tee -a $O/steps.txt
done
