#!/bin/bash
# in-kernel phase profile of the forward attention kernel (see attn_phase_prof.py); build first on the dev box:
#   bash scratch/build_abl.sh attention_ops prof "-DATTN_PROF"
cd $GRAFT_REPO_ROOT
BUTD_HIP_LIB=$GRAFT_REPO_ROOT/scratch/exp/libabl_prof.so python scratch/attn_phase_prof.py
