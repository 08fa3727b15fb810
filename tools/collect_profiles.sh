#!/bin/bash
# Runs ON THE GPU BOX (gpurun): every measurement artefact of a round, each rocprofv3 pass on its own
# (counters never share a run with --stats; FETCH_SIZE and WRITE_SIZE in separate passes: TCC has 4 slots).
#   tools/collect_profiles.sh <tag>     -> gpurun_out/<tag>_*
set -u
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out
BENCH="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --sustain-seconds 0"
python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench rc=$?"
python3 bench.py --gpus 1 --launcher --no-cpu-baseline > $OUT/${TAG}_bench_launcher.json 2> $OUT/${TAG}_bench_launcher.err; echo "launcher rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_stats -o s --output-format csv -- python3 bench.py --no-cpu-baseline > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_stats.err; echo "stats rc=$?"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/${TAG}_pmc -o fetch --output-format csv -- $BENCH > /dev/null 2>> $OUT/${TAG}_pmc.err; echo "fetch rc=$?"
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/${TAG}_pmc -o write --output-format csv -- $BENCH > /dev/null 2>> $OUT/${TAG}_pmc.err; echo "write rc=$?"
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace -d $OUT/${TAG}_pmc -o sq1 --output-format csv -- $BENCH > /dev/null 2>> $OUT/${TAG}_pmc.err; echo "sq1 rc=$?"
timeout -k 10 200 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d $OUT/${TAG}_pmc -o sq2 --output-format csv -- $BENCH > /dev/null 2>> $OUT/${TAG}_pmc.err; echo "sq2 rc=$?"
python3 tools/measure_extra.py > $OUT/${TAG}_measure_extra.txt 2> $OUT/${TAG}_measure_extra.err; echo "extra rc=$?"
if [ "${2:-}" != "nohost" ]; then
for n in 1024 4096; do timeout -k 5 200 python3 tools/host_scale.py --streams $n --hops 20 > $OUT/${TAG}_host_scale_${n}.json 2> $OUT/${TAG}_host_scale_${n}.err; echo "host scale $n rc=$?"; done
# 4096 streams whose hop clocks are NOT aligned (each of the eight feeders has its own phase inside the 216 ms period), and the same
# 4096 streams over two device loops on the one GPU of the box (--devices=0,0: two handles, two ingest and two post-processing threads)
timeout -k 5 200 python3 tools/host_scale.py --streams 4096 --hops 20 --phase-spread-ms 216 > $OUT/${TAG}_host_scale_4096_spread.json 2> $OUT/${TAG}_host_scale_4096_spread.err; echo "host scale 4096 spread rc=$?"
timeout -k 5 200 python3 tools/host_scale.py --streams 4096 --hops 20 --devices 0,0 > $OUT/${TAG}_host_scale_4096_two_loops.json 2> $OUT/${TAG}_host_scale_4096_two_loops.err; echo "host scale 4096 two loops rc=$?"
fi
ls $OUT/${TAG}_stats $OUT/${TAG}_pmc
