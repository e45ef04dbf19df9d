#!/bin/bash
# the f16x2 forward GEMM with its HBM round trips ablated (compile-time hooks in gemm_emu.hip, off in the product build): run from the
# repo root on the GPU box AFTER tools/build_variant.sh built ab/lib_{nostore,noloada,both}.so here (hipcc cross-compiles)
for v in "" nostore noloada both; do
  if [ -z "$v" ]; then python tools/mb_h2_ablate.py 2>/dev/null | tail -1; else HOISDF_LIB=$PWD/ab/lib_$v.so python tools/mb_h2_ablate.py 2>/dev/null | tail -1; fi
done
