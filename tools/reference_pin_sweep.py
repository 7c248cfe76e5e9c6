#!/usr/bin/env python3
"""CPU experiment behind tests/test_oracle_pins.py::test_reference_fixture_is_reproduced_two_sided: statistics of the reference's layout file
(test/DRB1-3123_unsorted.og.lay) against the oracle's layouts over seeds, thread counts and a sweep of the generating parameters.
Result (profiles/r03/reference_pin_sweep.txt): only runs WITHOUT the cooling phase (cooling_start = 1) reproduce the file."""
import sys, os, json
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np, odgi_amd as oa
from oracle import oracle as orc
G=os.path.join(ROOT,'tests','golden')
g=oa.Graph.from_gfa(os.path.join(G,'DRB1-3123_unsorted.gfa')); og=orc.Graph.from_product(g)
lay=oa.Layout.load(os.path.join(G,'DRB1-3123_unsorted.og.lay'))
p=oa.LayoutParams.defaults(g)
terms=orc.trace_terms(og, orc.params_from(p), 12345, 64, 0, True, 4000).reshape(-1,4)   # Zipf-sampled pairs (cooling sampler)
sh=np.asarray(g.step_handle).astype(np.int64); pos=np.asarray(g.step_pos).astype(np.int64); nl=np.asarray(g.node_len).astype(np.int64)
pf=np.asarray(g.path_first).astype(np.int64)
def stats(X,Y):
    X=np.asarray(X,dtype=np.float64); Y=np.asarray(Y,dtype=np.float64)
    out={}
    out['stress']=orc.path_stress_exhaustive(og,X,Y)
    pn,pb=orc.path_distance(og,X,Y); out['per_node']=pn; out['per_bp']=pb
    # adjacent steps
    ks=np.concatenate([np.arange(pf[i],pf[i+1]-1) for i in range(len(pf)-1)])
    a=sh[ks]; b=sh[ks+1]; d=(pos[ks+1]-pos[ks]).astype(np.float64)
    r=np.hypot(X[a]-X[b],Y[a]-Y[b])/np.maximum(d,1e-9)
    out['adj']=[float(np.percentile(r,q)) for q in (10,50,90)]
    ka,kb,oa_,ob=terms[:,0].astype(np.int64),terms[:,1].astype(np.int64),terms[:,2].astype(np.int64),terms[:,3].astype(np.int64)
    ea=(sh[ka]&~1)|oa_; eb=(sh[kb]&~1)|ob
    pa=pos[ka]+np.where((sh[ka]&1)!=oa_, nl[sh[ka]>>1],0); pb_=pos[kb]+np.where((sh[kb]&1)!=ob, nl[sh[kb]>>1],0)
    d=np.abs(pa-pb_).astype(np.float64); m=d>0
    r=np.hypot(X[ea]-X[eb],Y[ea]-Y[eb])[m]/d[m]
    out['zipf']=[float(np.percentile(r,q)) for q in (10,50,90)]
    C=np.cov(np.stack([X,Y])); ev=np.linalg.eigvalsh(C); out['aspect']=float(np.sqrt(ev[1]/max(ev[0],1e-12)))
    out['extent']=float(np.sqrt(ev[1]))
    return out
print('fixture',json.dumps(stats(lay.X,lay.Y)),flush=True)
for th in (1,2,4):
    for seed in (7,8,9):
        X0,Y0=oa.initial_layout(g,'d',seed=seed)
        X,Y,st=orc.layout_hogwild(og,orc.params_from(p),th,X0,Y0)
        print('oracle threads',th,'seed',seed,json.dumps(stats(X,Y)),flush=True)
print('--- sweep')
import dataclasses
def run(**kw):
    q=oa.LayoutParams.defaults(g, **kw)
    X0,Y0=oa.initial_layout(g,'d',seed=7)
    X,Y,st=orc.layout_hogwild(og,orc.params_from(q),2,X0,Y0)
    s=stats(X,Y)
    print(kw, round(s['stress'],4), [round(v,3) for v in s['adj']], [round(v,3) for v in s['zipf']], round(s['per_node'],3), round(s['per_bp'],4), flush=True)
S=g.n_steps
for kw in (dict(), dict(iter_max=20), dict(iter_max=15), dict(iter_max=10), dict(min_term_updates=5*S), dict(min_term_updates=2*S), dict(min_term_updates=S), dict(eps=0.1), dict(eps=1.0), dict(theta=0.9), dict(theta=0.5), dict(cooling_start=1.0), dict(cooling_start=0.8), dict(cooling_start=0.2), dict(space_max=100), dict(iter_max=100), dict(eta_max=float(S)**2)):
    try: run(**kw)
    except Exception as e: print(kw,'failed',e)
for init in ('g','u','h'):
    X0,Y0=oa.initial_layout(g,init,seed=7)
    X,Y,st=orc.layout_hogwild(og,orc.params_from(p),2,X0,Y0); s=stats(X,Y)
    print('init',init, round(s['stress'],4), [round(v,3) for v in s['adj']], [round(v,3) for v in s['zipf']], round(s['per_node'],3), round(s['per_bp'],4))
