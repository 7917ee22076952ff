#!/bin/bash
# Tuning sweep of the GF(65537) one-pass kernel: cache hints on loads / stores, twiddle window depth, first-round stagger.
# Variant libraries are built into _variants/ (see DESIGN.md section 4.2 (1a)); run on the GPU box.
for rep in 1 2; do
for v in default ld0_st2 ld2_st0 ld0_st0 tww24 tww40; do
  if [ $v = default ]; then unset GALOIS_AMD_LIB; else export GALOIS_AMD_LIB=$PWD/_variants/lib_$v.so; fi
  for s in 0 2 4; do
    echo "$v stagger=$s: $(GFA_NTT_FERMAT_STAGGER=$s python tools/fermat_time.py 1024 4096 2>/dev/null | grep batch | awk '{print $3, $8}' | tr '\n' ' ')"
  done
done; done
