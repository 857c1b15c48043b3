# Round-end profiles (GPU box, through gpurun): bench lines of every config, rocprofv3 kernel statistics of the C2 / C3 / C5 bench commands, PMC passes of C2.
# Everything lands under gpurun_out/final/; tools/collect_profiles.py copies the summaries into profiles/r06_*.
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
cd $R
python bench.py > $O/bench_C2_default.json 2> $O/bench_C2_default.err
python bench.py --gpus 1 --force-collectives --no-extras --no-cpu-baseline > $O/bench_C2_force_collectives.json 2> $O/bench_C2_force_collectives.err
python bench.py --isotropic-db --no-cpu-baseline --kernels-top 40 > $O/bench_C2_isotropic_db.json 2> $O/bench_C2_isotropic_db.err
python bench.py --fp32-store --no-extras --no-cpu-baseline > $O/bench_C2_fp32_store.json 2> $O/bench_C2_fp32_store.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_C2 -o C2 -- python $R/bench.py --no-extras --no-cpu-baseline --steps 20 > $O/prof_C2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_C5 -o C5 -- python $R/bench.py --config C5 --batch 16 --no-extras --no-cpu-baseline --steps 10 > $O/prof_C5.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_C3 -o C3 -- python $R/bench.py --config C3 --db 1000000 --no-extras --no-cpu-baseline --steps 10 > $O/prof_C3.log 2>&1
cd $R
bash tools/pmc_conv.sh C2
python bench.py --config C5 --batch 16 > $O/bench_C5_B16.json 2> $O/bench_C5_B16.err
python bench.py --config C4 --batch 16 --no-cpu-baseline > $O/bench_C4_B16.json 2> $O/bench_C4_B16.err
python bench.py --config C1 --batch 16 --no-cpu-baseline > $O/bench_C1_B16.json 2> $O/bench_C1_B16.err
python bench.py --config C3 --db 1000000 --no-cpu-baseline > $O/bench_C3_1M.json 2> $O/bench_C3_1M.err
python tools/dbbuild_bench.py > $O/dbbuild.log 2>&1
python tools/train_bench.py C3 4 10 > $O/train_C3_B4.json 2> $O/train_C3_B4.err
find $O -name '*_kernel_trace.csv' -delete
find $O -name '*.db' -delete
du -sh $O
