#!/bin/bash
# round 2, call y: the numbers that go into profiles/: bench line, kernel stats (1 and 3 search streams), PMC passes
cd /root/repo; R=/root/repo; O=$R/gpurun_out/r2y; mkdir -p $O; export TMPDIR=/tmp
python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
tail -1 $O/bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['breakdown'], d['first_calibration_s']); r=d['roofline']; print({k:(round(v['ms'],1), round(v['frac'],3)) for k,v in r['by_kernel'].items() if v}, r['all_int8_sweeps']['frac'], r['traffic'], d['cpu_baseline']['value'])"
( cd /tmp && P4V_SEARCH_STREAMS=1 rocprofv3 --kernel-trace -d $O/prof1 -o b -- python $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2>&1 )
python tools/kstats_db.py "$O/prof1/*.db" > $O/bench_1stream_kernel_stats.txt; head -16 $O/bench_1stream_kernel_stats.txt
( cd /tmp && rocprofv3 --kernel-trace -d $O/prof3 -o b -- python $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2>&1 )
python tools/kstats_db.py "$O/prof3/*.db" > $O/bench_3streams_kernel_stats.txt
python tools/kstats_db.py --busy "$O/prof3/*.db" >> $O/bench_3streams_kernel_stats.txt; tail -4 $O/bench_3streams_kernel_stats.txt | cut -c1-200
rm -rf $O/prof1 $O/prof3
bash tools/pmc_collect.sh fc1 $O/pmc_fc1_sweep6.json > $O/pmc_fc1.log 2>&1; tail -4 $O/pmc_fc1.log | cut -c1-400
bash tools/pmc_collect.sh fc2 $O/pmc_fc2_sweep7.json > $O/pmc_fc2.log 2>&1; tail -3 $O/pmc_fc2.log | cut -c1-400
