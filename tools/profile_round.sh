#!/bin/bash
# Round profile (run on the GPU box from the repo root):  tools/profile_round.sh r02
# 1) the default bench line (with cpu_baseline and extra workloads)
# 2) rocprofv3 --kernel-trace --stats of the same command (without the CPU legs)
# 3) separate --pmc passes (no tracing flags) for HBM traffic and the SQ counters of the query
#    path's kernels (tools/pmc_bench.sh: FETCH_SIZE, WRITE_SIZE, TCC_EA0_RDREQ[_128B], SQ_*;
#    MI355X_MICROARCH.md: on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes, so it is doubled)
# 4) the same with the look-ups inside the gather kernel (pre-pass off), the 8-way shard emulation, the world-1 RCCL run
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02}
cd $R; mkdir -p gpurun_out
timeout -k 10 900 python3 bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_n1.json 2> gpurun_out/${TAG}_bench_n1.err || { tail -5 gpurun_out/${TAG}_bench_n1.err; exit 1; }
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/kt; timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 $R/bench.py --no-cpu --no-extra > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> /tmp/kt.log || { tail -5 /tmp/kt.log; exit 1; }
cd $R
python3 tools/prof_summary.py /tmp/kt > gpurun_out/${TAG}_bench_kernel_trace_summary.txt
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_bench_rocprofv3_kernel_stats.csv
bash tools/pmc_bench.sh ${TAG}_default --no-extra || exit 1
NIQKI_LOOKUP_PREPASS=0 bash tools/pmc_bench.sh ${TAG}_noprepass --no-extra || exit 1
cd /tmp
rm -rf /tmp/kt2; NIQKI_LOOKUP_PREPASS=0 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2 -- python3 $R/bench.py --no-cpu --no-extra > $R/gpurun_out/${TAG}_noprepass_bench_under_rocprof.json 2> /tmp/kt2.log || exit 1
cd $R
python3 tools/prof_summary.py /tmp/kt2 | grep -E "gather_kernel|lookup_kernel|lookup_rows_kernel|probe_kernel|order_kernel|^==|kernel " > gpurun_out/${TAG}_noprepass_kernel_trace_summary.txt
timeout -k 10 600 python3 bench.py --shard-of 8 --no-cpu > gpurun_out/${TAG}_shard_of_8.json 2> gpurun_out/${TAG}_shard_of_8.err || exit 1
# the weak-scaling shape of --gpus 8 on rank 0 (every rank brings 4096 queries), and all 8 ranks of the group on this one GPU
timeout -k 10 600 python3 bench.py --shard-of 8 --batch 32768 --ring 2 --steps 5 --warmup 2 --no-cpu --no-extra > gpurun_out/${TAG}_shard_of_8_weak.json 2> gpurun_out/${TAG}_shard_of_8_weak.err || exit 1
timeout -k 10 600 python3 tools/bench_group_local.py --steps 3 > gpurun_out/${TAG}_group_local_8_shards.json 2> gpurun_out/${TAG}_group_local.err || exit 1
hipcc -O3 --offload-arch=gfx950 tools/ubench_partial_write.hip -o /tmp/ubench_partial_write 2>/dev/null && /tmp/ubench_partial_write > gpurun_out/${TAG}_ubench_partial_write.txt
NIQKI_FORCE_DIST=1 timeout -k 10 600 python3 bench.py --no-cpu --no-extra > gpurun_out/${TAG}_force_dist_world1.json 2> gpurun_out/${TAG}_force_dist_world1.err || exit 1
hipcc -O3 --offload-arch=gfx950 tools/ubench_sector.hip -o /tmp/ubench_sector 2>/dev/null && /tmp/ubench_sector > gpurun_out/${TAG}_ubench_sector.txt
cd /tmp
for c in TCC_EA0_RDREQ TCC_EA0_RDREQ_128B; do
  rm -rf /tmp/pmc_$c; timeout -k 10 120 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -- /tmp/ubench_sector > /dev/null 2>&1
  python3 $R/tools/prof_summary.py /tmp/pmc_$c | grep -E "sector_kernel" | sed 's/  */ /g' >> $R/gpurun_out/${TAG}_ubench_sector.txt
done
cd $R; head -14 gpurun_out/${TAG}_bench_kernel_trace_summary.txt | cut -c1-170
python3 -c "
import json
j=json.load(open('gpurun_out/${TAG}_bench_n1.json'))
print('value %.0f ms/step %.2f gather %.3f frac %.3f' % (j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'], j['roofline']['frac']))
print(json.dumps(j['cpu_baseline']))
print(json.dumps(j['sketch_kernel']))"
