#!/bin/bash
# round 5, GPU call 20: the large parity sweeps on the LAST build (nontemporal stores, k_errlog's new letter columns and event prefetch):
# GPU == oracle chunk by chunk — 13 modes of the record path x 60 000 reads, metagenome worker calls 4 x 20 000, transcriptome 5 x 30 000
cd "$(dirname "$0")/../.."
O=gpurun_out/r05w; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( timeout 900 python scripts/parity_sweep.py 60000 2>&1 | tail -16; timeout 600 python scripts/parity_meta_big.py 2>&1 | tail -6; timeout 600 python scripts/parity_trx_big.py 2>&1 | tail -7 ) | tee $O/parity_sweeps.log
