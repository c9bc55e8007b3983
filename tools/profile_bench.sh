#!/bin/bash
# Profiles bench.py's hot path with rocprofv3 on the GPU box (run through gpurun) AT THE BENCH'S OWN
# CONFIGURATION (default --pairs 32768).  Pass 1: kernel trace + stats.  Passes 2-5: PMC counters,
# each in a run of its own.  Pass 6: kernel trace of a run with the extras.
# Then: python3 tools/summarize_prof.py gpurun_out/prof profiles/r03
set -e
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
ARGS="bench.py --steps 6 --warmup 3 --no-extra --cpu-pairs 0"
echo "$ARGS" > $OUT/command.txt
timeout -s KILL 240 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python3 $ARGS > $OUT/trace.json 2> $OUT/trace.err
timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_mfma -o p -- python3 $ARGS > $OUT/pmc_mfma.json 2> $OUT/pmc_mfma.err || echo "pmc_mfma pass failed"
timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE -d $OUT/pmc_sq -o p -- python3 $ARGS > $OUT/pmc_sq.json 2> $OUT/pmc_sq.err || echo "pmc_sq pass failed"
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p -- python3 $ARGS > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o p -- python3 $ARGS > $OUT/pmc_write.json 2> $OUT/pmc_write.err
# the popcount formulation of north_star (knob bit 25 = 33554432): kernel trace, VALU counters, HBM traffic
PARGS="bench.py --steps 3 --warmup 3 --no-extra --cpu-pairs 0 --variant 33554432"
timeout -s KILL 240 rocprofv3 --kernel-trace --stats -d $OUT/pop_trace -o t -- python3 $PARGS > $OUT/pop_trace.json 2> $OUT/pop_trace.err || echo "pop_trace pass failed"
timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE -d $OUT/pop_sq -o p -- python3 $PARGS > $OUT/pop_sq.json 2> $OUT/pop_sq.err || echo "pop_sq pass failed"
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE -d $OUT/pop_fetch -o p -- python3 $PARGS > $OUT/pop_fetch.json 2> $OUT/pop_fetch.err || echo "pop_fetch pass failed"
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE -d $OUT/pop_write -o p -- python3 $PARGS > $OUT/pop_write.json 2> $OUT/pop_write.err || echo "pop_write pass failed"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_extras -o t -- python3 bench.py --steps 1 --warmup 1 --cpu-pairs 0 --pairs 4096 > $OUT/trace_extras.json 2> $OUT/trace_extras.err
find $OUT -type f | head -50
