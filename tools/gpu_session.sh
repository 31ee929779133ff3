# scratch: the command list of the current gpurun call (tools/README.md); the round's profile set is tools/profile_round.sh
O=$GRAFT_REPO_ROOT/gpurun_out/r5y; mkdir -p $O
cd $GRAFT_REPO_ROOT/tools/_variants/old_tree
for v in p6 p7 p8 p9 p10 p11 l4 l6 l7; do for m in "L=1000 n=150000" "L=1000 n=200000 fb=0" "L=300 n=400000" "L=1000 n=150000 mode=ragged2L"; do echo "## variant $v: $m"; KMX_LIB_VARIANT=$v timeout 600 python3 dev_bisect_sum.py $m 2>&1 | grep -v amdgpu; done; done > $O/old_variants_2.txt 2>&1
grep -c "whole: ok" $O/old_variants_2.txt; grep -B2 -A12 "MISMATCH\|DIFFER" $O/old_variants_2.txt | head -80
