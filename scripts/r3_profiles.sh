#!/bin/bash
# Everything profiles/r03_* and DESIGN.md section 6 quote, from ONE box: scripts/r3_profiles.sh  (writes gpurun_out/r3p/)
out=gpurun_out/r3p; mkdir -p $out
run() { name=$1; shift; echo "== $name"; timeout -k 10 400 "$@" > $out/$name.log 2>&1; echo "rc=$?"; tail -1 $out/$name.log | cut -c1-400; }
run bench_default python bench.py
run bench_driver python bench.py --steps 20 --warmup 5 --no-cpu-baseline
run bench_fp32 python bench.py --precision fp32 --steps 30 --warmup 10 --no-cpu-baseline
run bench_448 python bench.py --image 448 --tokens 30 --hidden 2048 --steps 20 --warmup 5 --no-cpu-baseline
run bench_eval python bench.py --mode eval --batch 1024 --steps 40 --warmup 10
run bench_host_overlap python bench.py --host-inputs overlap --steps 40 --warmup 10 --no-cpu-baseline
run step_series python scripts/step_series.py
run host_sections python scripts/host_sections.py 8
bash scripts/prof_stats.sh $out/prof_serial --steps 20 --warmup 5 --no-cpu-baseline --serial-streams > $out/prof_serial.log 2>&1; tail -2 $out/prof_serial.log | cut -c1-200
bash scripts/prof_stats.sh $out/prof_overlap --steps 20 --warmup 5 --no-cpu-baseline > $out/prof_overlap.log 2>&1; tail -2 $out/prof_overlap.log | cut -c1-200
timeout -k 10 700 bash scripts/pmc_traffic.sh > $out/pmc_traffic.log 2>&1; tail -8 $out/pmc_traffic.log | cut -c1-200
timeout -k 10 700 bash scripts/pmc_mfma.sh > $out/pmc_mfma.log 2>&1; tail -10 $out/pmc_mfma.log | cut -c1-200
