#!/bin/bash
# rocprofv3 counter passes for the per-kernel HBM traffic and the SQ wait/occupancy picture of the bench configuration
# (ONE context, B lockstep chains, delaySteps D).  Run on the GPU box from the repository root:
#     bash scripts/pmc_collect.sh 128 32 gpurun_out/pmc_r02
# One counter group per run (FETCH_SIZE and WRITE_SIZE do not fit one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots"); the
# profiled program is python3 itself (no wrapper in between).  Batched contexts only: a TCC-counter pass over a
# single-chain context (scripts/probe_sweep.py) crashed (round 1) or hung (round 2) the profiled process, DESIGN.md section 10.
set -e
B=${1:-128}; D=${2:-32}; OUT=$(realpath -m ${3:-gpurun_out/pmc}); ROOTDIR=$(pwd)
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp DQMC_DELAY_STEPS=$D
run() {   # name counters...
    local name=$1; shift
    timeout -k 10 200 rocprofv3 --pmc "$@" -d "$OUT/$name" -o "$name" --output-format csv -- python3 "$ROOTDIR/scripts/probe_batch.py" 16 10 1 qr "$B" > "$OUT/$name.log" 2>&1
    echo "$name rc=$?"; grep "sweeps/s" "$OUT/$name.log" || true
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA
run lds SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
