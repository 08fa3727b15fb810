cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for b in 16 8; do
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/l3_$b -o fetch --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --llr-block $b > /dev/null 2>> gpurun_out/l3.err
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/l3_$b -o write --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --llr-block $b > /dev/null 2>> gpurun_out/l3.err
done
echo done
