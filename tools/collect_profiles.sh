#!/bin/bash
# Runs ON THE GPU BOX (gpurun): every measurement artefact of a round, each rocprofv3 pass on its own
# (counters never share a run with --stats; FETCH_SIZE and WRITE_SIZE in separate passes: TCC has 4 slots).
#   tools/collect_profiles.sh <tag> [nohost|hostonly]     -> gpurun_out/<tag>_*   (nohost: without the 60 s capacity runs of the stream program; hostonly: only those)
set -u
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out
BENCH="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --sustain-seconds 0 --no-extra-legs"
if [ "${2:-}" != "hostonly" ]; then
python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench rc=$?"
python3 bench.py --gpus 1 --launcher --no-cpu-baseline > $OUT/${TAG}_bench_launcher.json 2> $OUT/${TAG}_bench_launcher.err; echo "launcher rc=$?"
# the default command (timed region + sustained leg + the three extra legs: value_incl_h2d, frontend_method1, configs4_iq) under the kernel trace:
# every kernel of every leg in one stats file; and the same without the extra legs, for the agreement of the dominant kernel's average
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_stats -o s --output-format csv -- python3 bench.py --no-cpu-baseline > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_stats.err; echo "stats rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_stats_main -o s --output-format csv -- python3 bench.py --no-cpu-baseline --no-extra-legs > $OUT/${TAG}_bench_main_under_rocprof.json 2> $OUT/${TAG}_stats_main.err; echo "stats main rc=$?"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/${TAG}_pmc -o fetch --output-format csv -- $BENCH > /dev/null 2>> $OUT/${TAG}_pmc.err; echo "fetch rc=$?"
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/${TAG}_pmc -o write --output-format csv -- $BENCH > /dev/null 2>> $OUT/${TAG}_pmc.err; echo "write rc=$?"
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace -d $OUT/${TAG}_pmc -o sq1 --output-format csv -- $BENCH > /dev/null 2>> $OUT/${TAG}_pmc.err; echo "sq1 rc=$?"
timeout -k 10 200 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d $OUT/${TAG}_pmc -o sq2 --output-format csv -- $BENCH > /dev/null 2>> $OUT/${TAG}_pmc.err; echo "sq2 rc=$?"
python3 tools/measure_extra.py > $OUT/${TAG}_measure_extra.txt 2> $OUT/${TAG}_measure_extra.err; echo "extra rc=$?"
fi
if [ "${2:-}" != "nohost" ]; then
# the stream program at real-time pace over 60 s of signal per stream (looped synthetic signal): every stream's hop due together
# (the aligned worst case) and every stream on its own hop phase
for n in 4096 4608; do timeout -k 5 330 python3 tools/host_scale.py --streams $n --hops 278 --loop-hops 20 --feeders 16 > $OUT/${TAG}_host_scale_${n}_aligned_60s.json 2> $OUT/${TAG}_host_scale_${n}_aligned_60s.err; echo "host scale $n aligned rc=$?"; done
for n in 5376 5888; do timeout -k 5 330 python3 tools/host_scale.py --streams $n --hops 278 --loop-hops 20 --phase-spread-ms 216 --phase-per-stream --feeders 16 > $OUT/${TAG}_host_scale_${n}_perstream_60s.json 2> $OUT/${TAG}_host_scale_${n}_perstream_60s.err; echo "host scale $n per-stream rc=$?"; done
fi
ls $OUT/${TAG}_stats $OUT/${TAG}_pmc 2>/dev/null
