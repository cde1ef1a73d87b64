#!/bin/bash
# rocprofv3 counters of the stuffing kernel (and, beside it, the coding kernel) over a short bench.py run: what the
# stuffing kernel executes, which is what it takes from the coding kernel it runs beside.
# usage (on the GPU box): tools/pmc_stream.sh <outdir> [bench.py args]
out=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/$out
EXTRA="$*"
run() { name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $R/$out/$name -- python3 $R/bench.py --steps 2 --warmup 1 --batch 128 --resident 64 --per-launch 64 --no-cpu-baseline --no-e2e $EXTRA > $R/$out/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $R/$out/$name.log; exit 1; }; }
run cls SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_VALU_INT32
run occ SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
python3 $R/tools/pmc_summary.py $R/$out k_stream > $R/$out/summary.txt
python3 $R/tools/pmc_summary.py $R/$out k_coded >> $R/$out/summary.txt
cat $R/$out/summary.txt
