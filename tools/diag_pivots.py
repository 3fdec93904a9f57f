"""How many pivots does the active-set iteration take on the bench batch?  A numpy restatement of the device
iteration (most violated row first, the 1-D problem along its line bounded by the two stored rows) on the oracle's
rows of one configuration-4 batch: pivots per QP, and per group of QPs that iterate together in a wavefront
(16 with four lanes per QP).  CPU only: python tools/diag_pivots.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sccav_cbf_amd import scenarios
from oracle import oracle as orc
orc.build()
A=65536
w=scenarios.config4(A=A, seed=1238)
p=orc.make_params(alpha=1.0, R=(0.5,0,0,0.5))
res=orc.solve_batch(p, w["state"], w["u_ref"], w["obs"], w["obs_type"], want_rows=True)
u,st,rows=res[0],res[1],res[2]
rows=np.asarray(rows)  # (M,A,4)? a0,a1,b,h
M=rows.shape[0]
a0,a1,b=rows[...,0],rows[...,1],rows[...,2]
# w-space: R=0.5 I -> Lc = sqrt(.5) I ; t = a / l ; rhs = b - a.un
l=np.sqrt(0.5)
ur=w["u_ref"]; un0=ur[:,0]; un1=np.arctan(np.tan(ur[:,1])*0.5)
t0=a0/l; t1=a1/l; rhs=b-(a0*un0+a1*un1)
n=np.sqrt(t0*t0+t1*t1); nz=n>1e-150
at0=np.where(nz,t0/np.where(nz,n,1),0); at1=np.where(nz,t1/np.where(nz,n,1),0); bt=np.where(nz,rhs/np.where(nz,n,1),np.where(rhs>0,np.inf,rhs))
W0=np.zeros(A);W1=np.zeros(A)
sa0=np.zeros((2,A));sa1=np.zeros((2,A));sb=-np.ones((2,A))
status=np.zeros(A,int); its=np.zeros(A,int)
ar=np.arange(A)
for it in range(200):
    viol=bt-at0*W0-at1*W1
    thr=1e-12*(abs(W0)+abs(W1))+1e-300
    k=np.argmax(viol,axis=0); vmax=viol[k,ar]
    go=(status==0)&(vmax>thr)
    if not go.any(): break
    its+=go
    c0=at0[k,ar];c1=at1[k,ar];cb=bt[k,ar]
    zero=(c0==0)&(c1==0)
    p0=c0*cb;p1=c1*cb;d0=-c1;d1=c0
    tlo=np.full(A,-np.inf);thi=np.full(A,np.inf);inf=np.zeros(A,bool)
    lsel=np.zeros(A,bool);hsel=np.zeros(A,bool)
    for j in range(2):
        cs=sa1[j]*d1+sa0[j]*d0; ee=sb[j]-sa0[j]*p0-sa1[j]*p1
        npar=abs(cs)>1e-14
        with np.errstate(all='ignore'): t=ee/cs
        up=cs>0
        tl=np.where(npar&up,t,-np.inf); th=np.where(npar&~up,t,np.inf)
        l1=tl>tlo; h1=th<thi
        if j==1: lsel,hsel=l1,h1
        tlo=np.where(l1,tl,tlo); thi=np.where(h1,th,thi)
        inf|=~npar&(ee>1e-10*(abs(sb[j])+abs(cb))+1e-300)
    with np.errstate(all='ignore'):
        inf|=tlo>thi+1e-10*(abs(tlo)+abs(thi))
    bad=zero|inf
    uselo=tlo>0; usehi=~uselo&(thi<0)
    t=np.where(uselo,tlo,np.where(usehi,thi,0))
    keep1=(uselo&lsel)|(usehi&hsel)
    upd=go&~bad; mv=upd&~keep1
    with np.errstate(all='ignore'):
        W0=np.where(upd,p0+t*d0,W0);W1=np.where(upd,p1+t*d1,W1)
    for arr,c in ((sa0,c0),(sa1,c1),(sb,cb)):
        arr[1]=np.where(mv,arr[0],arr[1]); arr[0]=np.where(upd,c,arr[0])
    status=np.where(go&bad,1,status)
print("status agree with oracle:", (status==np.asarray(st)).mean(), "infeasible", (status==1).mean())
print("pivots per QP: mean %.2f"%its.mean(), "hist", np.bincount(its)[:14])
for g in (8,16,32,64):
    mx=its.reshape(-1,g).max(axis=1)
    print("per %2d QPs: mean of max %.2f  (steps incl. final check %.2f)"%(g,mx.mean(),mx.mean()+1), np.bincount(mx)[:14])
feas=status==0
print("feasible mean", its[feas].mean(), "infeasible mean", its[~feas].mean())
