#!/bin/bash
# carried cells (TRX_CARRY_CELLS) against the build without: results and times, one job
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
mkdir -p $O/bits_a $O/bits_b
TRX_LIB=$R/profiles/ab_libs/libtrx_nocarry.so python profiles/r05/ab_bits.py 30000 $O/bits_a > $O/bits_nocarry.txt 2>$O/bits_nocarry.err
python profiles/r05/ab_bits.py 30000 $O/bits_b > $O/bits_carry.txt 2>$O/bits_carry.err
python profiles/r05/ab_bits_compare.py $O/bits_a $O/bits_b > $O/bits_compare.txt 2>&1
grep calc_probs $O/bits_nocarry.txt > $O/bits_cp_a.txt; grep calc_probs $O/bits_carry.txt > $O/bits_cp_b.txt
diff $O/bits_cp_a.txt $O/bits_cp_b.txt >> $O/bits_compare.txt; echo "calc_probs diff rc $?" >> $O/bits_compare.txt
rm -rf $O/bits_a $O/bits_b
