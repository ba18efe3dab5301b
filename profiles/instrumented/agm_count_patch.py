import sys
d=sys.argv[1]
p=d+'/trx_device.hpp'; s=open(p).read()
old='''__device__ __forceinline__ double cel_pair(double kc, double a1, double b1, double pp, double a2,
                                           double b2)
{'''
assert old in s
s=s.replace(old,'''__device__ unsigned long long g_dbg[16];
'''+old+'''
    int my_steps = 0, done = 0;
    const double kc_in = kc;''')
old2='''        if (fabs(g - q) <= g * 5e-7) return true;'''
assert old2 in s
s=s.replace(old2,'''        my_steps += 1;
        if (fabs(g - q) <= g * 5e-7) return true;''')
old3='''#pragma unroll 1
    for (int it = 0; it < 20; ++it) {
        if (step()) break;
        if (step()) break;
    }
    const double d1 = em * (em + p1), d2 = em * (em + pp);'''
assert old3 in s
s=s.replace(old3,'''    const unsigned long long act = __ballot(1);
    int wave_steps = 0;
    {
        // steps every lane needs (run individually), and the wave's maximum
        bool fin = false;
        for (int it = 0; it < 40; ++it) {
            if (!fin) fin = step();
            if (__ballot(!fin) == 0ull) break;
        }
        int m = my_steps;
        for (int o = 32; o > 0; o >>= 1) { int x = __shfl_xor(m, o); m = m > x ? m : x; }
        wave_steps = (m + 1) & ~1;      // two steps per trip
    }
    atomicAdd(&g_dbg[0], (unsigned long long)my_steps);
    atomicAdd(&g_dbg[3 + (my_steps > 10 ? 10 : my_steps)], 1ull);
    if ((int)__lane_id() == __ffsll((long long)act) - 1) {
        atomicAdd(&g_dbg[1], (unsigned long long)wave_steps * 64ull);
        atomicAdd(&g_dbg[2], (unsigned long long)wave_steps * (unsigned long long)__popcll(act));
    }
    const double d1 = em * (em + p1), d2 = em * (em + pp);''')
open(p,'w').write(s)
p=d+'/trx_kernels.hip'; s=open(p).read()
old='int trx_set_debug_node_counts(int on)'
assert old in s
s=s.replace(old,'''int trx_dbg_read(unsigned long long* out, int reset)
{
    unsigned long long z[16] = {0};
    if (out) hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg), sizeof(z));
    if (reset) hipMemcpyToSymbol(HIP_SYMBOL(g_dbg), z, sizeof(z));
    return 0;
}

'''+old,1)
open(p,'w').write(s)
