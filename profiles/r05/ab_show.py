import json,sys
for f in sys.argv[1:]:
    d=json.loads([l for l in open(f) if l.startswith('{')][-1])
    r=d['roofline']
    print(f.split('/')[-1], "value %.4e launch %.3f ms frac %.4f | n100 %.3e n200 %.3e irr %.3e (%.3f ms) | allsub %.2f ms"%(d['value'], r['mean_launch_ms'], r['frac'], d['shapes']['n100']['evals_per_s'], d['shapes']['n200']['evals_per_s'], d['shapes']['n2000_irregular']['evals_per_s'], d['shapes']['n2000_irregular'].get('mean_launch_ms',0), r['all_subexposures']['mean_launch_ms']))
