# Run on the GPU box from the repo root (gpurun -- bash tools/collect_profiles.sh [tag]): the bench line, the bench under
# rocprofv3 --kernel-trace --stats, separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ counters) over tools/kbench.py, the PMC
# summary bench.py quotes its roofline.traffic from, the batch-128 bench, the other BASELINE configurations
# (tools/config_bench.py, tools/c3_bench.py + its kernel stats and counters) and the training step.
# Then, in the container: python tools/install_profiles.py --tag r03 --src gpurun_out/r03p
set -e
TAG=${1:-r03}
P=gpurun_out/${TAG}p
mkdir -p $P
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --steps 3 --warmup 2 > $P/bench.json 2> $P/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > $P/bench_under_rocprof.json 2> $P/trace.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $P/$c -o run -- python3 tools/kbench.py --reps 2 > $P/$c.log 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $P/${c}_k2 -o run -- python3 tools/kbench.py --only k2 --reps 3 > $P/${c}_k2.log 2>&1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --kernel-trace --output-format csv -d $P/sq -o run -- python3 tools/kbench.py --reps 2 > $P/sq.log 2>&1
for f in $P/FETCH_SIZE_k2/*/*counter_collection.csv $P/FETCH_SIZE_k2/*counter_collection.csv; do [ -f "$f" ] && cp $f $P/FETCH_SIZE/k2_counter_collection.csv; done
for f in $P/WRITE_SIZE_k2/*/*counter_collection.csv $P/WRITE_SIZE_k2/*counter_collection.csv; do [ -f "$f" ] && cp $f $P/WRITE_SIZE/k2_counter_collection.csv; done
python tools/make_pmc_profile.py --fetch $P/FETCH_SIZE --write $P/WRITE_SIZE --sq $P/sq --out $P/pmc_kernels.json --note "round 3 kernels (conv_c2 / conv_g2 / conv_h as round 2 + run-time knots_len in conv_h, branch-free activations and s_setprio in conv_c2 / conv_g2); kbench at the pipeline's 256-sample slab; K2 at its 33-sample slab"
python bench.py --steps 3 --warmup 1 --batch 128 --no-cpu-baseline --no-other-configs > $P/bench_batch128.json 2> $P/bench128.err || true
python tools/config_bench.py > $P/config_bench.txt 2>&1 || true
python tools/c3_bench.py > $P/c3_bench.txt 2>&1 || true
rocprofv3 --kernel-trace --stats --output-format csv -d $P/c3trace -o run -- python3 tools/c3_prof.py > $P/c3trace.log 2>&1 || true
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $P/c3sq -o run -- python3 tools/c3_prof.py > $P/c3sq.log 2>&1 || true
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $P/c3$c -o run -- python3 tools/c3_prof.py > $P/c3$c.log 2>&1 || true
done
python tools/train_bench.py > $P/train_bench.txt 2>&1 || true
python tools/train_step_profile.py >> $P/train_bench.txt 2>&1 || true
rocprofv3 --kernel-trace --stats --output-format csv -d $P/traintrace -o run -- python3 tools/train_step_profile.py > $P/traintrace.log 2>&1 || true
rocprofv3 --kernel-trace --stats --output-format csv -d $P/trainc3 -o run -- python3 tools/train_prof_c3.py > $P/trainc3.log 2>&1 || true
find $P -name "*kernel_stats.csv" | head -5
tail -c 400 $P/bench.json
