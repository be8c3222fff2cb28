#!/bin/bash
# One parametrised GPU-box job (replaces the per-call scripts of rounds 1-2).  Usage, from the repo root on the box:
#   tools/gpu_job.sh <tag> <step> [<step> ...]        results under gpurun_out/<tag>/
# steps: tests | tests:<pytest -k expr> | bench[:steps] | kstats1 | kstats3 | kstatsg[:calls] | prodprof[:stem] | pmc:<layer> | layer:<bench_layer args> |
#        forward | modules:<net> | sh:<command>
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp
TAG=$1; shift; O=$R/gpurun_out/$TAG; mkdir -p $O
for step in "$@"; do
  name=${step%%:*}; arg=""; [ "$step" != "$name" ] && arg=${step#*:}
  echo "=== $step"
  case $name in
    tests)   if [ -n "$arg" ]; then timeout 1500 python -m pytest tests -m gpu -x -q -k "$arg" 2>&1 | tail -15 | tee $O/tests_k.log
             else timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $O/tests.log; fi ;;
    bench)   timeout 900 python bench.py --steps ${arg:-10} --warmup 3 > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
             tail -1 $O/bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], d.get('breakdown'), d.get('first_calibration_s')); print({k:(v['launches'], round(v['ms'],1), round(v['frac'],3), {s:(q['launches'], round(q['avg_launch_ms']*1e3,1), round(q['frac'],3)) for s,q in v['by_stage'].items()}) for k,v in r.get('by_kernel',{}).items() if v}, r.get('all_int8_sweeps',{}).get('frac'), r.get('whole_search'), (d.get('cpu_baseline') or {}).get('value'))" ;;
    kstats1|kstats3|kstats4)
             n=${name#kstats}
             ( cd /tmp && P4V_SEARCH_STREAMS=$n timeout 600 rocprofv3 --kernel-trace -d $O/prof$n -o b -- python $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-roofline --no-extras > /dev/null 2>&1 )
             python tools/kstats_db.py "$O/prof$n/*.db" > $O/bench_${n}stream_kernel_stats.txt
             [ $n != 1 ] && python tools/kstats_db.py --busy "$O/prof$n/*.db" >> $O/bench_${n}stream_kernel_stats.txt
             head -24 $O/bench_${n}stream_kernel_stats.txt | cut -c1-180; rm -rf $O/prof$n ;;
    kstatsg) # kernel statistics of the grouped search with <arg> concurrent group calls
             n=${arg:-1}
             ( cd /tmp && P4V_GROUP_CALLS=$n timeout 600 rocprofv3 --kernel-trace -d $O/profg$n -o b -- python $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-roofline --no-extras > /dev/null 2>&1 )
             KSTATS_TOP=45 python tools/kstats_db.py "$O/profg$n/*.db" > $O/bench_group${n}_kernel_stats.txt
             python tools/kstats_db.py --busy "$O/profg$n/*.db" >> $O/bench_group${n}_kernel_stats.txt
             head -52 $O/bench_group${n}_kernel_stats.txt | cut -c1-170; tail -8 $O/bench_group${n}_kernel_stats.txt | cut -c1-200; rm -rf $O/profg$n ;;
    prodprof) # profile of the production step, joined with the engine's launch records (tools/prof_join.py); arg = output stem
             stem=${arg:-r6_production_by_stage}; T=/tmp/pp_$TAG; rm -rf $T; mkdir -p $T
             ( cd /tmp && timeout 900 rocprofv3 --kernel-trace -d $T/trace -o t -- python $R/bench.py --profile --steps 3 --warmup 2 --dump-launches $T/launches.json > $O/prodprof_bench.json 2> $O/prodprof_bench.err )
             for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
               tag=$(echo $pass | cut -d' ' -f1)
               ( cd /tmp && timeout 900 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $T/pmc_$tag -o p -- python $R/bench.py --profile --steps 1 --warmup 1 --no-roofline > $T/pmc_$tag.log 2>&1 )
               f=$(find $T/pmc_$tag -name "*counter_collection.csv" | head -1)
               [ -n "$f" ] && { head -1 $f > $T/$tag.csv; grep -E "k_sweep|k_sos|k_bound|k_slice|k_pack" $f >> $T/$tag.csv; } || echo "no counter csv for $tag: $(tail -2 $T/pmc_$tag.log)"
             done
             python tools/prof_join.py --launches $T/launches.json --trace "$T/trace/*.db" --fetch $T/FETCH_SIZE.csv --write $T/WRITE_SIZE.csv \
                    --counters $T/TCC_HIT_sum.csv $T/SQ_VALU_MFMA_BUSY_CYCLES.csv --out $O/$stem 2>&1 | cut -c1-200 | tee $O/prodprof.log
             KSTATS_TOP=40 python tools/kstats_db.py "$T/trace/*.db" > $O/${stem%_by_stage}_kernel_stats.txt; cp $T/launches.json $O/
             python tools/kstats_grid.py "$T/trace/*.db" 60 > $O/${stem%_by_stage}_kernel_time_by_grid.txt ;;
    pmc)     bash tools/pmc_collect.sh $arg $O/pmc_$arg.json > $O/pmc_$arg.log 2>&1; tail -6 $O/pmc_$arg.log | cut -c1-400 ;;
    layer)   timeout 600 python tools/bench_layer.py $arg 2>&1 | tail -12 | tee -a $O/layer.log ;;
    forward) timeout 600 python tools/bench_forward.py 2>&1 | tail -8 | tee $O/forward.log ;;
    modules) timeout 900 python tools/module_times.py $arg 2>&1 | tail -30 | tee $O/modules_$arg.log ;;
    sh)      bash -c "$arg" 2>&1 | tail -40 | tee -a $O/sh.log ;;
    *)       echo "unknown step $step" ;;
  esac
done
