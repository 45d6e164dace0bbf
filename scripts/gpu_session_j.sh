#!/bin/bash
set +e
export TMPDIR=/tmp
O=gpurun_out/r2j
mkdir -p $O
for v in 1 0; do FP_PLAN=$v python scripts/host_overhead.py 2>&1 | grep "host issue" | sed "s/^/PLAN=$v: /"; done
cd /tmp
# kernel-trace stats: serial (exclusive durations) and concurrent, KITTI and Matterport
( FP_SERIAL=1 FP_PLAN=0 timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/$O/prof_serial_kitti -o p -- python /root/repo/scripts/step_loop.py kitti 5 3 ) > /root/repo/$O/prof_serial_kitti.log 2>&1
( timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/$O/prof_conc_kitti -o p -- python /root/repo/scripts/step_loop.py kitti 5 3 ) > /root/repo/$O/prof_conc_kitti.log 2>&1
( FP_SERIAL=1 FP_PLAN=0 timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/$O/prof_serial_mp -o p -- python /root/repo/scripts/step_loop.py matterport 5 3 ) > /root/repo/$O/prof_serial_mp.log 2>&1
# HBM traffic: two separate PMC passes (FETCH_SIZE / WRITE_SIZE do not fit one pass), counters only
( FP_SERIAL=1 FP_PLAN=0 timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /root/repo/$O/pmc_f_kitti -o pmc -- python /root/repo/scripts/step_loop.py kitti 2 2 ) > /root/repo/$O/pmc_f.log 2>&1
( FP_SERIAL=1 FP_PLAN=0 timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /root/repo/$O/pmc_w_kitti -o pmc -- python /root/repo/scripts/step_loop.py kitti 2 2 ) > /root/repo/$O/pmc_w.log 2>&1
( FP_SERIAL=1 FP_PLAN=0 timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /root/repo/$O/pmc_f_mp -o pmc -- python /root/repo/scripts/step_loop.py matterport 2 2 ) > /root/repo/$O/pmc_f_mp.log 2>&1
( FP_SERIAL=1 FP_PLAN=0 timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /root/repo/$O/pmc_w_mp -o pmc -- python /root/repo/scripts/step_loop.py matterport 2 2 ) > /root/repo/$O/pmc_w_mp.log 2>&1
cd /root/repo
ls $O/prof_serial_kitti/* | head; find $O -name "*.db" | head
du -sh $O
