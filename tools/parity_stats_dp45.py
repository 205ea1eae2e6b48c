import sys, glob, json, os, numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,os.path.join(ROOT,'light-path-tracer_amd')); os.chdir(ROOT)
import ltrace
for f in sorted(glob.glob('tests/golden/rays_dp45_*.npz')):
    g=np.load(f); meta=json.loads(str(g['meta'])); n=g['alpha'].size
    fa=np.full(n,np.nan); w=np.zeros(n,dtype=np.int64); st=np.zeros(n,dtype=np.int8); ev=np.zeros(n,dtype=np.uint32)
    ltrace.trace_batch_kerr(meta['M'],meta['a'],meta['r_obs'],g['alpha'],g['theta'],np.pi/2,max(5000.0,6*meta['r_obs']),g['refine'],fa,w,integrator='dp45',precision=64,out_status=st,out_rhs_evals=ev)
    same=(st==1)==(g['status']==1); esc=same&(st==1); d=np.abs(fa[esc]-g['final_alpha'][esc])
    print(os.path.basename(f)[:34].ljust(34), 'flips',(~same).sum(),'median %.2e p90 %.2e p99 %.2e max %.2e'%(np.median(d),np.quantile(d,.9),np.quantile(d,.99),d.max()), 'evals %.2f vs %.2f, differing %d'%(ev.mean(), g['rhs_evals'].mean(), (ev!=g['rhs_evals']).sum()))
