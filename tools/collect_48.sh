# Counters of the three kernels on BASELINE config 5's lattice (48^4: rows of 24 pairs = one and a half MFMA tiles), 50 samples per
# launch (= the sites of 253 samples of 32^4).  gpurun -- bash tools/collect_48.sh ; then copy gpurun_out/r03p48/pmc48.json to
# profiles/r03_pmc_lat48_kernels.json
set -e
P=gpurun_out/r03p48
mkdir -p $P
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/kbench.py --lattice 48,48,48,48 --batch 50 --reps 3 > $P/kbench48.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $P/$c -o run -- python3 tools/kbench.py --lattice 48,48,48,48 --batch 50 --reps 2 > $P/$c.log 2>&1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --kernel-trace --output-format csv -d $P/sq -o run -- python3 tools/kbench.py --lattice 48,48,48,48 --batch 50 --reps 2 > $P/sq.log 2>&1
python tools/make_pmc_profile.py --fetch $P/FETCH_SIZE --write $P/WRITE_SIZE --sq $P/sq --slab 50 --lattice 48,48,48,48 --out $P/pmc48.json --note "48^4 (config 5's lattice), 50 samples per launch; the last segment of every row is half empty" > /dev/null
tail -3 $P/kbench48.txt
