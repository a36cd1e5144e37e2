set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
cd $R
timeout -k 10 500 python bench.py > $O/bench_1gpu.json 2> $O/bench_1gpu.err
tail -c 600 $O/bench_1gpu.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 1 --no-cpu-baseline > $O/stats_run.log 2>&1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 1 --spp 4 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --steps 1 --spp 4 --no-cpu-baseline > $O/pmc_write.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_sq -- python3 bench.py --steps 1 --spp 4 --no-cpu-baseline > $O/pmc_sq.log 2>&1
timeout -k 10 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_tcc -- python3 bench.py --steps 1 --spp 4 --no-cpu-baseline > $O/pmc_tcc.log 2>&1
python3 tools/pmc_traffic.py $O synth_room_v1_1920x1080x256 $O/traffic_latest.json
mkdir -p $O/pmc && cp -r $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_tcc $O/pmc/ 2>/dev/null || true
python3 tools/summarize_rocprof.py $O/stats $O/r01_final_rocprof.txt "bench.py --steps 1 --no-cpu-baseline (1920x1080x256, 1536 rounds)" > /dev/null
python3 tools/summarize_rocprof.py $O/pmc $O/r01_final_pmc.txt "bench.py --steps 1 --spp 4 --no-cpu-baseline (24 rounds, full-size queues): FETCH_SIZE / WRITE_SIZE / SQ / TCC passes" > /dev/null
# keep only the small summaries
rm -rf $O/stats $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_tcc $O/pmc
ls -la $O
