# Counters of a training step (tools/train_step_profile.py: 32^4, 2 spline layers, batch 16; tools/train_prof_c3.py: 16^3, 8 spline
# layers, batch 256): gpurun -- bash tools/collect_train_pmc.sh ; then python tools/install_train_pmc.py
set -e
P=gpurun_out/r03ptrain
mkdir -p $P
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $P/$c -o run -- python3 tools/train_step_profile.py > $P/$c.log 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $P/c3$c -o run -- python3 tools/train_prof_c3.py > $P/c3$c.log 2>&1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $P/sq -o run -- python3 tools/train_step_profile.py > $P/sq.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $P/c3sq -o run -- python3 tools/train_prof_c3.py > $P/c3sq.log 2>&1
ls $P
