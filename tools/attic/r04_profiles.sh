# Round-4 evidence (run on the GPU box through gpurun; results under gpurun_out/r04/, copied into profiles/ afterwards):
# the default bench line, the RCCL protocol with one rank, rocprofv3 kernel statistics (C2 / C3 at 1 M / C5), PMC passes of the C2 step, the other configurations' lines
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
cd $R
python bench.py > $O/bench_C2_default.json 2> $O/bench_C2_default.err
for port in 29517 29533; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $port bench.py --gpus 1 --force-collectives --no-extras --no-cpu-baseline > $O/bench_C2_force_collectives.json 2> $O/bench_C2_force_collectives.err
  echo "force-collectives rc=$? bytes=$(stat -c %s $O/bench_C2_force_collectives.json)"
  [ -s $O/bench_C2_force_collectives.json ] && break
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_C2 -o C2 -- python $R/bench.py --no-extras --no-cpu-baseline --steps 20 --repeats 0 > $O/prof_C2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_C5 -o C5 -- python $R/bench.py --config C5 --batch 16 --no-extras --no-cpu-baseline --steps 10 --repeats 0 > $O/prof_C5.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_C3 -o C3 -- python $R/bench.py --config C3 --db 1000000 --no-extras --no-cpu-baseline --steps 10 --repeats 0 > $O/prof_C3.log 2>&1
cd $R
bash tools/pmc_conv.sh C2
bash tools/pmc_conv.sh C5
python bench.py --config C5 --batch 16 > $O/bench_C5_B16.json 2> $O/bench_C5_B16.err
python bench.py --config C4 --batch 16 --no-cpu-baseline > $O/bench_C4_B16.json 2> $O/bench_C4_B16.err
python bench.py --config C1 --batch 16 --no-cpu-baseline > $O/bench_C1_B16.json 2> $O/bench_C1_B16.err
python bench.py --config C3 --db 1000000 --no-cpu-baseline > $O/bench_C3_1M.json 2> $O/bench_C3_1M.err
python tools/dbbuild_bench.py > $O/dbbuild.log 2>&1
python tools/train_bench.py C3 4 10 > $O/train_C3_B4.json 2> $O/train.err
find $O -name "*kernel_trace.csv" -delete
find $R/gpurun_out/pmc_C2 $R/gpurun_out/pmc_C5 -name "*kernel_trace.csv" -delete
ls -la $O $O/prof_C2 | head -40
