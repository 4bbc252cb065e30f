set -e
scripts/build_variant.sh small0 -DGBL_FORCE_COLLECT_SMALL=0 > /dev/null
scripts/build_variant.sh small1 -DGBL_FORCE_COLLECT_SMALL=4 > /dev/null
for n in 1024 4096 8192; do
for st in ply plymask all; do
python scripts/ab_inproc.py $n 32 $st build/lib_small0.so build/lib_small1.so 2>&1 | grep median
done; done
