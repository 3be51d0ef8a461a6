#!/bin/bash
# the valence / stock-dialect rows of tools/stock_timing.py under library builds and switches: bash tools/val_ab.sh
for spec in "default:" "default:DSA_OCT_STREAMS=0" "build_abl/lib_oct0.so:"; do
  IFS=: read -r lib e1 <<< "$spec"
  if [ "$lib" = default ]; then unset DSA_LIB; else export DSA_LIB=$PWD/$lib; fi
  [ -n "$e1" ] && export $e1
  echo "== $spec"
  python tools/stock_timing.py 4096 2>&1 | cut -c1-120
  [ -n "$e1" ] && unset ${e1%%=*}
done
