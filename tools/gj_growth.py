"""Offline check of the growth-watched, pivot-free Gauss-Jordan of lii_iekf.hip (gj12_loop) on matrices A = I + P11 G, B = P[:12, :]
taken from simulated LIO / LO sequences through the oracle: how often does the per-column growth test send an elimination to the
pivoting fallback, how large is the growth, and how far is the pivot-free gain from the exactly (mpmath) computed one, next to
the threshold-pivoted gain's distance."""
import numpy as np, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O
from harness import synth
import mpmath as mp
mp.mp.dps = 50
O.lib()

def gj_nopivot(A, B, gmax=64.0):
    M=np.hstack([A,B]).astype(np.float64); n=A.shape[0]
    g=np.abs(M[:, :n]).max(axis=0); g0=g.copy()
    ok=True
    for k in range(n):
        piv=M[k,k]
        if not np.isfinite(piv) or piv==0: return None, np.inf, False
        rowk=M[k,:]/piv
        m=M[:,k].copy()
        M=M-np.outer(m,rowk); M[k,:]=rowk
        g=np.maximum(g,np.abs(M[:, :n]).max(axis=0))
    ok=bool(np.all(g<=gmax*np.maximum(g0,1.0)))
    return M[:,n:], float((g/np.maximum(g0,1.0)).max()), ok

def gj_threshold(A,B,tau=0.25):
    M=np.hstack([A,B]).astype(np.float64); n=A.shape[0]
    for k in range(n):
        col=np.abs(M[k:,k])
        if not col[0]>=tau*col[1:].max(initial=0.0):
            p=k+int(np.argmax(col)); M[[k,p]]=M[[p,k]]
        rowk=M[k,:]/M[k,k]; m=M[:,k].copy()
        M=M-np.outer(m,rowk); M[k,:]=rowk
    return M[:,n:]

def exact(A,B):
    X=(mp.matrix(A.tolist())**-1)*mp.matrix(B.tolist())
    return np.array([[float(X[i,j]) for j in range(B.shape[1])] for i in range(B.shape[0])])

def tri_to_G(ne):
    G=np.zeros((12,12)); G[np.triu_indices(12)]=ne[:78]; return G+G.T-np.diag(np.diag(G))

def run(hall, sensor, nscans, label, imu_en=True, pstart=None):
    map_pts = hall.surface_points(0.15, noise=0.01, seed=3)
    tree=O.Tree("oracle"); tree.build(map_pts)
    st=O.state_init()
    P=O.StateView(st).cov.copy() if pstart is None else pstart.copy()
    nfb=0; gm=0; en=0; et=0
    for s in range(nscans):
        R=synth.rot_zyx(0.02*np.sin(s),-0.01,0.3+0.05*s); p=np.array([0.5+0.1*s,-0.4,0.1])
        scan=synth.make_scan(hall,sensor,R,p,noise=0.02,seed=5+s)
        v=O.StateView(st); v.rot_end[:]=R; v.pos_end[:]=p
        st0=O.state_boxplus(st,np.r_[0.004,-0.003,0.005,0.02,-0.015,0.01,np.zeros(18)])
        r=tree.iterate_once(scan,st0,search=True,imu_en=imu_en,threads=8)
        G=tri_to_G(r["out91"])
        A=np.eye(12)+P[:12,:12]@G; B=P[:12,:]
        Xn,g,ok=gj_nopivot(A,B); Xt=gj_threshold(A,B); Xe=exact(A,B)
        sc=np.sqrt(np.abs(np.diag(P)))[:12,None]*np.sqrt(np.abs(np.diag(P)))[None,:]
        nfb+= (not ok); gm=max(gm,g)
        en=max(en, np.abs(Xn-Xe).max()/np.abs(Xe).max()); et=max(et, np.abs(Xt-Xe).max()/np.abs(Xe).max())
        Pinv=np.linalg.inv(P); Pinv[:12,:12]+=G
        P=np.linalg.inv(Pinv); P=0.5*(P+P.T)
        Q=np.zeros((24,24)); Q[0:3,0:3]=np.eye(3)*1e-6; Q[3:6,3:6]=np.eye(3)*1e-6; Q[12:15,12:15]=np.eye(3)*1e-4; Q[15:24,15:24]=np.eye(9)*1e-8
        F=np.eye(24); F[3:6,12:15]=np.eye(3)*0.1; F[0:3,15:18]=-np.eye(3)*0.1; F[12:15,18:21]=-np.eye(3)*0.1
        P=F@P@F.T+Q
    print(f"{label}: {nscans} eliminations, fallbacks {nfb}, largest per-column growth {gm:.2f}, gain error vs exact: pivot-free {en:.1e}, threshold-pivoted {et:.1e}")

hall = synth.Hall(size=(20.0, 16.0, 6.0), n_boxes=6, seed=3)
run(hall,"tiny",25,"hall / 2 k points, LIO")
run(hall,"vlp16",10,"hall / 30 k points, LIO")
run(hall,"tiny",10,"hall / 2 k points, LO",imu_en=False)
cor = synth.Hall(size=(60.0, 3.0, 3.0), n_boxes=0, seed=4)
run(cor,"tiny",15,"corridor / 2 k points, LIO")
run(cor,"vlp16",8,"corridor / 30 k points, LIO")
# a prior with wildly different scales (pose collapsed, velocity / bias blocks of order 1)
P0=np.eye(24); P0[:6,:6]*=1e-8; P0[6:12,6:12]*=1e-6
run(hall,"vlp16",6,"hall / 30 k points, LIO, collapsed pose prior",pstart=P0)
rng=np.random.default_rng(1)
Q,_=np.linalg.qr(rng.normal(size=(24,24))); P1=Q@np.diag(10.0**rng.uniform(-9,0,24))@Q.T
run(hall,"vlp16",6,"hall / 30 k points, LIO, random dense prior spectrum 1e-9..1",pstart=P1)
