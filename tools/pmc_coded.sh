#!/bin/bash
# rocprofv3 counters of the coding kernel over a short bench.py run, one pass (one process) per counter set.
# usage: tools/pmc_coded.sh <outdir> [bench.py args]
out=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/$out
EXTRA="$*"
run() { name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $R/$out/$name -- python3 $R/bench.py --steps 2 --warmup 1 --batch 8 --resident 4 --per-launch 4 --no-cpu-baseline --no-e2e $EXTRA > $R/$out/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $R/$out/$name.log; exit 1; }; }
run cls SQ_INSTS_VALU SQ_INSTS_VALU_CVT SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_THREAD_CYCLES_VALU
run occ SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU
run lds SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA
run fetch FETCH_SIZE GRBM_GUI_ACTIVE
run write WRITE_SIZE
python3 $R/tools/pmc_summary.py $R/$out k_coded > $R/$out/summary.txt
cat $R/$out/summary.txt
