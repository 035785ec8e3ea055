cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# (1) kernel stats of the default bench command (headline + configs record), no CPU baseline to keep the trace short
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/prof_bench.json 2> gpurun_out/prof_bench.err || exit 1
# (2) HBM traffic of the headline step (separate passes, per the guide)
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_fetch -- python bench.py --steps 3 --warmup 1 --no-configs --no-cpu-baseline > gpurun_out/prof_fetch.json 2> gpurun_out/prof_fetch.err || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_write -- python bench.py --steps 3 --warmup 1 --no-configs --no-cpu-baseline > gpurun_out/prof_write.json 2> gpurun_out/prof_write.err || exit 1
# (3) per-launch dump of the profiled leg
SDM_PROF_DUMP=gpurun_out/r02_launches.csv timeout -k 10 200 python bench.py --steps 5 --warmup 3 --no-configs --no-cpu-baseline > gpurun_out/prof_dump.json 2> gpurun_out/prof_dump.err || exit 1
# (4) kernel stats of the C5 bf16 training step
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c5bf16b -- python tools/train_probe.py c5 bf16 4 > gpurun_out/prof_c5bf16b.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c4bf16b -- python tools/train_probe.py c4 bf16 4 > gpurun_out/prof_c4bf16b.log 2>&1 || exit 1
echo done
