for s in 11 12; do python scripts/fuzz_gemm_bf16.py $s 60 | tail -2; done
for s in 21 22; do python scripts/fuzz_bn_pair.py $s 36 | tail -3; done
python scripts/fuzz_bn_backward.py 31 20 | tail -2
for s in 41 42; do python scripts/fuzz_bn_conv.py $s 24 | tail -3; done
python scripts/fuzz_bn_handoff.py 32 30 | tail -2
python scripts/fuzz_conv_bf16.py 33 60 | tail -2
python scripts/fuzz_attention.py 34 30 | tail -2
python scripts/fuzz_parity.py 35 2>&1 | tail -2
python scripts/fuzz_rowops.py 36 2>&1 | tail -2
python __graft_entry__.py smoke 2>&1 | tail -2
