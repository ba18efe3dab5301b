#!/bin/bash
# results of an A/B library against the tree's, array by array:  bash profiles/r05/ab_bits_lib.sh <variant> [rows]
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
V=$1; N=${2:-20000}
rm -rf $O/bits_a $O/bits_b; mkdir -p $O/bits_a $O/bits_b
TRX_LIB=$R/profiles/ab_libs/libtrx_$V.so python profiles/r05/ab_bits.py $N $O/bits_a > $O/bits_$V.txt 2>$O/bits_$V.err
python profiles/r05/ab_bits.py $N $O/bits_b > $O/bits_tree.txt 2>$O/bits_tree.err
python profiles/r05/ab_bits_compare.py $O/bits_a $O/bits_b | grep -v "rows differing 0 of" | tail -15
grep calc_probs $O/bits_$V.txt > $O/bits_cp_a.txt; grep calc_probs $O/bits_tree.txt > $O/bits_cp_b.txt
diff $O/bits_cp_a.txt $O/bits_cp_b.txt > /dev/null && echo "calc_probs: same bits" || echo "calc_probs: DIFFERENT"
rm -rf $O/bits_a $O/bits_b
