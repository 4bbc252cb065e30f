#!/bin/bash
# k_collect_small between two builds (scripts/build_rev.sh / build_variant.sh), by what is stored, at small batch sizes
L=${LIBS:-"build/lib_head3.so build/lib_cur.so"}
for n in ${SIZES:-1024 4096 8192}; do for st in ${STREAMS:-none mask obs all}; do
python scripts/ab_inproc.py $n 32 $st $L 2>&1 | grep median | cut -c1-100
done; done
