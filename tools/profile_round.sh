#!/bin/bash
# Round profile (run on the GPU box from the repo root):  tools/profile_round.sh r03
# 1) the default bench line (with cpu_baseline and extra workloads)
# 2) rocprofv3 --kernel-trace --stats of the timed steps alone (bench.py --no-legs), and of the pipelined mode
# 3) separate --pmc passes (no tracing flags) for HBM traffic and the SQ counters of the query
#    path's kernels (tools/pmc_bench.sh: FETCH_SIZE, WRITE_SIZE, TCC_EA0_RDREQ[_128B], SQ_*;
#    MI355X_MICROARCH.md: on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes, so it is doubled)
# 4) the same with the look-ups inside the gather kernel (pre-pass off), the 8-way shard emulation, the world-1 RCCL run
# Two parts (a gpurun call is limited to 20 minutes):  tools/profile_round.sh r05 line   |   tools/profile_round.sh r05 rest
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r03}
PART=${2:-all}
cd $R; mkdir -p gpurun_out
if [ "$PART" = "line" ] || [ "$PART" = "all" ]; then
timeout -k 10 1000 python3 bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_n1.json 2> gpurun_out/${TAG}_bench_n1.err || { tail -5 gpurun_out/${TAG}_bench_n1.err; exit 1; }
python3 -c "
import json
j=json.load(open('gpurun_out/${TAG}_bench_n1.json'))
print('value %.0f ms/step %.2f gather %.3f frac %.3f (algorithmic %.3f, layout min %.3f)' % (j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'], j['roofline']['frac'], j['roofline']['frac_algorithmic'], j['roofline']['frac_layout_min']))
print(json.dumps(j['cpu_baseline'])[:1500])"
[ "$PART" = "line" ] && exit 0
fi
cd /tmp; export TMPDIR=/tmp
# the traced command runs ONLY the index build, the warm-up and the timed steps (--no-legs), sketch kernel and query one
# after the other (--no-overlap): a kernel's average in the summary is that of launches which had the device to
# themselves, like the ones roofline.avg_launch_ms averages
rm -rf /tmp/kt; timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 $R/bench.py --no-legs --no-overlap --steps 20 --warmup 5 > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> /tmp/kt.log || { tail -5 /tmp/kt.log; exit 1; }
cd $R
python3 tools/prof_summary.py /tmp/kt > gpurun_out/${TAG}_bench_kernel_trace_summary.txt
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_bench_rocprofv3_kernel_stats.csv
# separately labelled: the default step (next batch sketched beside the query: niqki_sketch_ahead / niqki_query_ahead),
# default and priority streams -- the gather-path kernels share the CUs with the sketch kernel there, their times are no
# roofline figures (the trace also holds the few one-after-the-other steps bench.py takes its roofline from)
cd /tmp
for pm in "" "--priority-streams"; do
  rm -rf /tmp/ktp; timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktp -- python3 $R/bench.py --no-legs $pm --steps 20 --warmup 5 > $R/gpurun_out/${TAG}_bench_overlapped${pm}.json 2> /tmp/ktp.log || { tail -5 /tmp/ktp.log; exit 1; }
  python3 $R/tools/prof_summary.py /tmp/ktp > $R/gpurun_out/${TAG}_overlapped${pm}_kernel_trace_summary.txt
done
cd $R
bash tools/pmc_bench.sh ${TAG}_default || exit 1
# two processes on this one GPU (bench.py --gpus 2 starts its own ranks when no launcher did): gloo for bench.py's own barrier, the library's ipc transport
# for the exchange; with and without the next batch's sketch kernel beside the exchange
for ov in "" "--no-overlap"; do
  timeout -k 10 600 python3 bench.py --gpus 2 --steps 9 --warmup 2 --no-cpu --no-extra $ov > gpurun_out/${TAG}_bench_n2_ipc_one_gpu${ov}.json 2> gpurun_out/${TAG}_bench_n2.err || { tail -5 gpurun_out/${TAG}_bench_n2.err; exit 1; }
done
timeout -k 10 600 python3 bench.py --shard-of 8 --no-cpu > gpurun_out/${TAG}_shard_of_8.json 2> gpurun_out/${TAG}_shard_of_8.err || exit 1
# the weak-scaling shape of --gpus 8 on rank 0 (every rank brings 4096 queries), and all 8 ranks of the group on this one GPU
timeout -k 10 600 python3 bench.py --shard-of 8 --batch 32768 --ring 2 --steps 5 --warmup 2 --no-cpu --no-extra > gpurun_out/${TAG}_shard_of_8_weak.json 2> gpurun_out/${TAG}_shard_of_8_weak.err || exit 1
timeout -k 10 600 python3 tools/bench_group_local.py --steps 3 > gpurun_out/${TAG}_group_local_8_shards.json 2> gpurun_out/${TAG}_group_local.err || exit 1
NIQKI_FORCE_DIST=1 timeout -k 10 600 python3 bench.py --no-cpu --no-extra > gpurun_out/${TAG}_force_dist_world1.json 2> gpurun_out/${TAG}_force_dist_world1.err || exit 1
# the walk's access pattern alone, and which instruction kinds issue side by side (built by hand into tools/bin/)
[ -x tools/bin/ubench_lines ] && timeout -k 10 60 tools/bin/ubench_lines > gpurun_out/${TAG}_ubench_lines.txt 2>&1
[ -x tools/bin/ubench_mix ] && timeout -k 10 60 stdbuf -oL tools/bin/ubench_mix > gpurun_out/${TAG}_ubench_mix.txt 2>&1
# round 5: the short-read path's kernels, a 500 000-genome index, (the file path on 2048 files is part of the bench line's cli_files)
bash tools/profile_reads4.sh ${TAG} > /dev/null 2>&1
timeout -k 10 600 python3 bench.py --genomes 500000 --no-legs --steps 6 --warmup 2 > gpurun_out/${TAG}_bench_500k_genomes.json 2> gpurun_out/${TAG}_bench_500k.err
# LAST (it rebuilds the library of this scratch copy with clock reads in the gather kernel): phase clocks of a gather workgroup
touch niqki_amd/csrc/nq_query.hip && make -C niqki_amd/csrc HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DNQ_GATHER_CLOCK" > gpurun_out/${TAG}_clock_build.log 2>&1 && {
  timeout -k 10 300 python3 tools/gather_clock.py --shard-of 8 --no-cpu --no-extra --steps 3 2>&1 | grep -v "^{\|amdgpu.ids" > gpurun_out/${TAG}_gather_phase_clocks_shard_of_8.txt
  timeout -k 10 300 python3 tools/gather_clock.py --no-cpu --no-extra --steps 3 2>&1 | grep -v "^{\|amdgpu.ids" > gpurun_out/${TAG}_gather_phase_clocks_whole_range.txt
}
cd $R
head -14 gpurun_out/${TAG}_bench_kernel_trace_summary.txt | cut -c1-170
[ -f gpurun_out/${TAG}_bench_under_rocprof.json ] && python3 -c "
import json
j=json.load(open('gpurun_out/${TAG}_bench_under_rocprof.json'))
print('traced run: value %.0f ms/step %.2f gather %.3f' % (j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms']))"
