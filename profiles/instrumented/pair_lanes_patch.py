"""Instrumented build: lanes of the pair loop's trips by what they do (profiles/pair_lanes.py reads the counters).
   g_lanes[0] trips, [1] lanes with a pair, [2] lanes whose pair is on the disc (enter the Mandel-Agol stage),
   [3] trips with no lane on the disc (the stage is skipped), [4] pairs of contact cells (all S sub-exposures)"""
import sys
d = sys.argv[1]
p = d + '/trx_kernels.hip'; s = open(p).read()
old = '''                    for (int p0 = 0; p0 < total; p0 += 64) {
                        const int p = p0 + lane;
                        if (p < total) {'''
assert old in s
s = s.replace(old, '''                    for (int p0 = 0; p0 < total; p0 += 64) {
                        const int p = p0 + lane;
                        {
                            bool on = false, heavyp = false;
                            if (p < total) {
                                const int d_ = (int)pdesc[p];
                                const int h_ = d_ & 63;
                                const unsigned meta_ = cs.meta[h_];
                                const RowC& c_ = LONG ? cu : rows[meta_ & 0xffu];
                                const int ht_ = (int)((meta_ >> 8) & 0xffu) - 1;
                                heavyp = ht_ < 0;
                                double sE_ = cs.sE[h_], cE_ = cs.cE[h_];
                                const int s_ = s0 + (d_ >> 6);
                                const double frac_ = (ht_ < 0) ? fma((double)(s_ + 1) - 0.5, a.rS, -0.5) : tier_xw[2 * (ht_ * kTierMaxNodes + s_)];
                                kepler_full(c_.nmot * ((cs.t[h_] + a.exptime * frac_) - c_.t0) + c_.Mtr, c_.e, sE_, cE_);
                                const double ce_ = cE_ - c_.e;
                                const double X_ = fma(c_.ax, ce_, c_.bx * sE_), Y_ = fma(c_.ay, ce_, c_.by * sE_);
                                const double yc_ = Y_ * c_.cosi;
                                on = Y_ >= 0.0 && fma(X_, X_, yc_ * yc_) < (1.0 + c_.k) * (1.0 + c_.k);
                            }
                            const unsigned long long m1 = __ballot(p < total), m2 = __ballot(on), m3 = __ballot(heavyp);
                            if (lane == 0) {
                                atomicAdd(&g_lanes[0], 1ull);
                                atomicAdd(&g_lanes[1], (unsigned long long)__popcll(m1));
                                atomicAdd(&g_lanes[2], (unsigned long long)__popcll(m2));
                                if (m2 == 0ull) atomicAdd(&g_lanes[3], 1ull);
                                atomicAdd(&g_lanes[4], (unsigned long long)__popcll(m3));
                            }
                        }
                        if (p < total) {''')
old = 'int trx_set_debug_node_counts(int on)'
assert old in s
s = s.replace(old, '''int trx_dbg_lanes(unsigned long long* out, int reset)
{
    unsigned long long z[8] = {0};
    if (out) hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lanes), sizeof(z));
    if (reset) hipMemcpyToSymbol(HIP_SYMBOL(g_lanes), z, sizeof(z));
    return 0;
}

''' + old, 1)
old = 'namespace {\n'
i = s.index(old)
s = s[:i] + old + '__device__ unsigned long long g_lanes[8];\n' + s[i + len(old):]
open(p, 'w').write(s)
