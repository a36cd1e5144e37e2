# The round's evidence run (on the GPU box): bench line, rocprofv3 kernel stats of the same command, PMC passes (counters only with
# --kernel-trace) for the dominant kernel's fabric traffic / SQ / TCC, condensed into small text summaries for profiles/.
# usage: bash tools/profiling/profile_bench.sh <tag, e.g. r02>
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r03}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout -k 10 600 python bench.py --steps 5 --warmup 2 > $O/bench_1gpu.json 2> $O/bench_1gpu.err
tail -c 400 $O/bench_1gpu.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/stats_run.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_write.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_sq -- python3 bench.py --steps 1 --warmup 0 --spp 16 --no-cpu-baseline > $O/pmc_sq.log 2>&1
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_tcc -- python3 bench.py --steps 1 --warmup 0 --spp 16 --no-cpu-baseline > $O/pmc_tcc.log 2>&1
mkdir -p $O/pmc_traffic && cp -r $O/pmc_fetch $O/pmc_write $O/pmc_traffic/
python3 tools/pmc_traffic.py $O/pmc_traffic synth_room_v1_1920x1080x256 $O/traffic_latest.json pt_persistent_kernel 1 256
mkdir -p $O/pmc_lim && cp -r $O/pmc_sq $O/pmc_tcc $O/pmc_lim/
python3 tools/pmc_limiter.py $O/pmc_lim synth_room_v1_1920x1080x256 $O/limiter_latest.json pt_persistent_kernel 5 || echo "limiter summary failed"
rm -rf $O/pmc_lim
mkdir -p $O/pmc && cp -r $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_tcc $O/pmc/ 2>/dev/null || true
python3 tools/summarize_rocprof.py $O/stats $O/${TAG}_final_rocprof.txt "bench.py --steps 2 --warmup 1 --no-cpu-baseline (1920x1080x256; persistent pipeline: two launches per frame)" > /dev/null
python3 tools/summarize_rocprof.py $O/pmc $O/${TAG}_final_pmc.txt "FETCH_SIZE / WRITE_SIZE passes: bench.py --steps 1 --warmup 0 (full 256 spp, the counting side-render at 4 spp included); SQ / TCC passes: --spp 16" > /dev/null
# BASELINE.json configs[2] (hw6 practice6_2, 1024x1024x256) under the same kernel-stats profiler
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_hw6 -- python3 tools/tuning/p6_probe.py --spp 256 "" > $O/stats_hw6_run.log 2>&1 || echo "hw6 stats pass failed"
python3 tools/summarize_rocprof.py $O/stats_hw6 $O/${TAG}_hw6_config3_rocprof.txt "tools/tuning/p6_probe.py --spp 256 (hw6 practice6_2 1024x1024x256, persistent hw6 pipeline: two launches per frame, two renders)" > /dev/null || true
rm -rf $O/stats_hw6
rm -rf $O/stats $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_tcc $O/pmc $O/pmc_traffic
ls -la $O
