#!/bin/bash
# The one recipe for everything that runs on the GPU box:
#   gpurun --timeout S -- 'bash scripts/gpu.sh <tag> <step> [<step> ...]'
# Every step writes under gpurun_out/ with <tag> in the name (merged back; what is worth keeping is copied to profiles/).
# Steps (arguments after the first ':' -- use ',' for spaces inside them):
#   smoke                     __graft_entry__.smoke()
#   test:<pytest args>        python -m pytest <args> -m gpu -q            e.g. test:tests/test_gpu_lloyd.py,-x
#   bench[:<bench.py args>]   the bench line (default flags) + a one-line digest
#   shard                     the 1M-row shard of the headline config (what one of 8 GPUs sees)
#   api[:<args>]              bench.py --api (whole kmeans_cuda() calls)
#   stats[:<bench.py args>]   rocprofv3 --kernel-trace --stats of the bench command -> kernel_stats_<tag>.csv
#   timeline[:<bench.py args>] kernel trace of the bench (default: the 1M-row shard) -> the last 40 launches with their gaps
#   pmc[:<bench.py args>]     PMC passes (one run per counter group, --kernel-trace only) -> pmc_<tag>_summary.json
#   pmc:cmd=<command>         the same passes over any command, e.g. pmc:cmd=python,scripts/config_d.py,--samples,8000000,--shard,0/8
#   hip:<file.hip>[,args]     a standalone .hip program (scripts/mfma_probe.hip, scripts/coarse_probe.hip) built on the box
#                             with $HIPFLAGS and run -> <name>_<tag>.log
#   hippmc:<file.hip>[,args]  the same program under rocprofv3 --pmc (MFMA busy, clock per launch: scripts/probe_pmc.py)
#   configs                   whole calls: config B / mixtures / config C shape with yinyang_t = 0.1 and 0 (scripts/config_b.py)
#   knn[:<config_d.py args>]  config D's share (scripts/config_d.py)
#   scale:<N>                 scripts/scale_check.sh N
#   py:<script,args>          python <script> <args> > gpurun_out/<tag>_<script>.log
#   env:<NAME=VALUE>          export for the steps that follow
set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out; mkdir -p $OUT
TAG=$1; shift
digest() { python3 - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("no bench line:", e); sys.exit(0)
r = d.get("roofline", {})
print({k: d.get(k) for k in ("value", "ms_per_step", "n_gpus")}, "frac", r.get("frac"), "kernel_ms", r.get("kernel_ms"),
      d.get("breakdown_ms_per_step"), "verify", d.get("verify", {}).get("ok"),
      "api", (d.get("api_kmeans_cuda") or {}).get("ms_per_iteration"))
PY
}
for step in "$@"; do
  name=${step%%:*}; args=""; [ "$name" != "$step" ] && args=${step#*:}; args=${args//,/ }
  echo "== $step"
  case $name in
    env) export "$args" ;;
    smoke) timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ;;
    test) timeout 3000 python -m pytest $args -m gpu -q > $OUT/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_$TAG.log | cut -c1-300 ;;
    bench) timeout 900 python bench.py $args > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "bench rc=$?"; digest $OUT/bench_$TAG.json ;;
    shard) timeout 600 python bench.py --samples 1000000 --steps 40 --warmup 10 --no-cpu-baseline --verify-rows 200000 > $OUT/bench1m_$TAG.json 2> $OUT/bench1m_$TAG.err; echo "rc=$?"; digest $OUT/bench1m_$TAG.json ;;
    api) timeout 900 python bench.py --api --steps 20 $args > $OUT/bench_api_$TAG.json 2> $OUT/bench_api_$TAG.err; echo "rc=$?"
         python -c "import json;d=json.loads(open('$OUT/bench_api_$TAG.json').read().strip().splitlines()[-1]);print('api', d['ms_per_step'], [(c['iterations'],round(c['loop_s'],4),round(c['wall_s'],3)) for c in d['calls']])" ;;
    stats) rm -rf $OUT/prof_$TAG
         timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG -o p -- python bench.py ${args:---steps 20 --warmup 5 --no-cpu-baseline --no-verify --no-api-leg} > $OUT/prof_$TAG.log 2>&1; echo "rocprof rc=$?"
         python scripts/rocpd_stats.py $OUT/prof_$TAG/p_results.db $OUT/kernel_stats_$TAG.csv | head -12 | cut -c1-160; rm -rf $OUT/prof_$TAG ;;
    timeline) rm -rf $OUT/prof_$TAG   # the last launches of the bench command as a timeline with gaps
         timeout 900 rocprofv3 --kernel-trace -d $OUT/prof_$TAG -o p -- python bench.py ${args:---samples 1000000 --steps 6 --warmup 4 --no-cpu-baseline --no-verify --no-api-leg} > $OUT/prof_$TAG.log 2>&1; echo "rocprof rc=$?"
         python scripts/rocpd_timeline.py $OUT/prof_$TAG/p_results.db 0 100000 | tail -40 | cut -c1-150 | tee $OUT/timeline_$TAG.log; rm -rf $OUT/prof_$TAG ;;
    pmc) CMD="python bench.py ${args:---steps 10 --warmup 5 --no-cpu-baseline --no-verify --no-api-leg}"; i=0
         case "$args" in cmd=*) CMD="${args#cmd=}" ;; esac   # pmc:cmd=<any command>: the same passes over another program (k-NN: scripts/config_d.py)
         for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
                    "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
           i=$((i+1)); rm -rf /tmp/pmc_$i
           timeout 400 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmc_$i -o pmc -- $CMD > /tmp/pmc_$i.log 2>&1; echo "pmc pass $i ($grp) rc=$?"
         done
         PMC_TAG=$TAG python scripts/pmc_summary.py $OUT/pmc_${TAG}_summary.json | cut -c1-600 ;;
    hip) src=${args%% *}; rest=""; [ "$src" != "$args" ] && rest=${args#* }; exe=scratch/bin/$(basename $src .hip); mkdir -p scratch/bin
         /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -w ${HIPFLAGS:-} $src -o $exe && { echo "## flags: ${HIPFLAGS:-}" >> $OUT/$(basename $src .hip)_$TAG.log; timeout 600 $exe $rest | tee -a $OUT/$(basename $src .hip)_$TAG.log; } ;;
    hippmc) src=${args%% *}; rest=""; [ "$src" != "$args" ] && rest=${args#* }; exe=scratch/bin/$(basename $src .hip); mkdir -p scratch/bin   # a probe under the MFMA-busy / clock counters
         /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -w ${HIPFLAGS:-} $src -o $exe && { rm -rf /tmp/hippmc_$TAG
           timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/hippmc_$TAG -o pmc -- $exe $rest > $OUT/$(basename $src .hip)_pmc_$TAG.log 2>&1
           python scripts/probe_pmc.py /tmp/hippmc_$TAG | tee -a $OUT/$(basename $src .hip)_pmc_$TAG.log; } ;;
    configs) : > $OUT/configs_$TAG.log
         run() { echo "## $1" | tee -a $OUT/configs_$TAG.log; shift; ( "$@" ) 2>&1 | grep -E "kmeans_cuda wall|carried bounds|knn_cuda" | tee -a $OUT/configs_$TAG.log; }
         for rep in 1 2; do
           run "config B: default (yinyang_t=0.1)" timeout 300 python scripts/config_b.py --yinyang 0.1 --verbosity 0
           run "config B: yinyang_t=0" timeout 300 python scripts/config_b.py --yinyang 0 --verbosity 0
           run "4M-row mixture tol 0.01: default" timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0.1 --tolerance 0.01 --verbosity 0
           run "4M-row mixture tol 0.01: yinyang_t=0" timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0 --tolerance 0.01 --verbosity 0
         done
         run "4M-row mixture tol 1e-4: default" timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0.1 --tolerance 0.0001 --verbosity 0
         run "4M-row mixture tol 1e-4: yinyang_t=0" timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0 --tolerance 0.0001 --verbosity 0
         run "config C shape (fp16 angular, 8 virtual shards), k-means++" env KMCUDA_AMD_VIRTUAL_SHARDS=8 timeout 300 python scripts/config_b.py --init k-means++ --metric cos --dtype f16 --yinyang 0.1 --verbosity 0 ;;
    knn) KMCUDA_AMD_KNN_STATS=1 timeout 900 python scripts/config_d.py ${args:---samples 8000000 --shard 0/8} 2>&1 | grep -E "knn_cuda|brute|k-NN filter|pairs" | cut -c1-250 | tee $OUT/knn_$TAG.log ;;
    scale) bash scripts/scale_check.sh $args 2>&1 | tee $OUT/scale_check_$TAG.log | tail -30 ;;
    py) s=${args%% *}; timeout 3000 python $args > $OUT/${TAG}_$(basename $s .py).log 2>&1; echo "rc=$?"; tail -12 $OUT/${TAG}_$(basename $s .py).log | cut -c1-300 ;;
    *) echo "unknown step $step"; exit 2 ;;
  esac
done
