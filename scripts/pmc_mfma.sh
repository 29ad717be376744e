#!/bin/bash
# MFMA utilisation evidence per kernel family (BASELINE.json north_star: "MFMA utilisation for the conv, GEMM and
# classifier kernels"): two --pmc passes over the bench (serialised by the profiler), summed per family.
#   mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs)   (the gfx94x MfmaUtil formula;
#               the GRBM counter comes back summed over the XCDs)
#   valu_per_mfma = SQ_INSTS_VALU / SQ_INSTS_MFMA,  lds conflict cycles per LDS instruction
# usage: scripts/pmc_mfma.sh [OUT.json]   (writes gpurun_out/pmc_mfma.json; copy to profiles/)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_mfma
rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/a.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/b -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/b.log 2>&1
python3 scripts/pmc_mfma_summarise.py "${1:-gpurun_out/pmc_mfma.json}" $OUT
