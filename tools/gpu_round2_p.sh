#!/bin/bash
# round 2, call p: kernel trace of the 3-stream bench: how busy is the GPU during a calibration
cd /root/repo
mkdir -p gpurun_out/r2p
export TMPDIR=/tmp
( cd /tmp && timeout 900 rocprofv3 --kernel-trace -d /root/repo/gpurun_out/r2p/prof -o b -- python /root/repo/bench.py --no-cpu-baseline --no-roofline --steps 3 --warmup 2 > /root/repo/gpurun_out/r2p/bench.json 2> /root/repo/gpurun_out/r2p/bench.err )
tail -1 gpurun_out/r2p/bench.json | cut -c1-400
DB=$(find gpurun_out/r2p/prof -name "*.db" | head -1)
python tools/kstats_db.py $DB | head -24
python tools/kstats_db.py --busy $DB
rm -f $DB
