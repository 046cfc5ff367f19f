#!/bin/bash
export SOAK_SEL='one_pass or bn_backward or batch_norm or deferred or data_parallel or graph or allocator or train_step or stream_and_resume or bn_pair or handoff or statistics or igemm or k_tail or parity or two_first or loss_launch or plane or umap_negatives or tail_gradient'
export SOAK_TIMEOUT=1000
bash scripts/soak.sh 16 3
