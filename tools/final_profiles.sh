set -x
R=$GRAFT_REPO_ROOT
cd $R
python bench.py > gpurun_out/final_C2.json 2> gpurun_out/final_C2.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --force-collectives --no-extras --no-cpu-baseline > gpurun_out/final_C2_coll.json 2> gpurun_out/final_C2_coll.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final_C2 -o C2 -- python $R/bench.py --no-extras --no-cpu-baseline --steps 20 > $R/gpurun_out/prof_final_C2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final_C5 -o C5 -- python $R/bench.py --config C5 --batch 16 --no-extras --no-cpu-baseline --steps 10 > $R/gpurun_out/prof_final_C5.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final_C3 -o C3 -- python $R/bench.py --config C3 --db 1000000 --no-extras --no-cpu-baseline --steps 10 > $R/gpurun_out/prof_final_C3.log 2>&1
cd $R
bash tools/pmc_conv.sh C2
ls gpurun_out/pmc_C2
python bench.py --config C5 --batch 16 > gpurun_out/bench_r3_C5.json 2> gpurun_out/bench_r3_C5.err
python bench.py --config C4 --batch 16 --no-cpu-baseline > gpurun_out/bench_r3_C4.json 2> gpurun_out/bench_r3_C4.err
python bench.py --config C1 --batch 16 --no-cpu-baseline > gpurun_out/bench_r3_C1.json 2> gpurun_out/bench_r3_C1.err
python bench.py --config C3 --db 1000000 --no-cpu-baseline > gpurun_out/bench_r3_C3_1M.json 2> gpurun_out/bench_r3_C3_1M.err
python tools/dbbuild_bench.py > gpurun_out/dbbuild_r3.log 2>&1
python tools/train_bench.py C3 4 10 > gpurun_out/train_r3.json 2> gpurun_out/train_r3.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final_train -o train -- python $R/tools/train_bench.py C3 4 5 > $R/gpurun_out/prof_final_train.log 2>&1
cd $R
