set -e
scripts/build_variant.sh small0 -DGBL_FORCE_COLLECT_SMALL=0 > /dev/null
scripts/build_variant.sh small1 -DGBL_FORCE_COLLECT_SMALL=4 > /dev/null
for st in none scalars mask maskscalars obs obsscalars all; do
python scripts/ab_inproc.py 4096 32 $st build/lib_small0.so build/lib_small1.so 2>&1 | grep median
done
