"""Random-shape, random-WEIGHTS parity sweep of the whole forward (HIP path vs the CPU oracle; run on the GPU box:
python profiles/fuzz_whole.py <seed> <trials>).  Each trial draws a weight regime (vcrnet_amd.weights.regime_weights), a weight
seed and, for the trained-like regime, the scale factor (1.5 .. 4)."""
import torch, numpy as np, sys, time
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import vcrnet_amd
from vcrnet_amd import native, synth, composed
from test_hip_forward import build_net
import oracle
rs=np.random.RandomState(int(sys.argv[1]) if len(sys.argv)>1 else 0)
nets={}
worst=(0,0)
t0=time.time()
for trial in range(int(sys.argv[2]) if len(sys.argv)>2 else 12):
    k=int(rs.choice([5,20,20,40])); B=int(rs.randint(2,9)); N=int(rs.randint(k+1,900)) if rs.rand()<0.8 else int(rs.choice([k+1,k+2,1024,1025,1279]))
    kind=str(rs.choice(["whole","dist","identity"]))
    kw={"whole":{},"dist":dict(vcp_nn="dist"),"identity":dict(pointer="identity")}[kind]
    regime=str(rs.choice(["default","seed4321","trained","trained","randemb"])); wseed=int(rs.randint(0,10**6)); scale=float(rs.uniform(1.5,4.0))
    net,w=build_net(regime=regime,seed=wseed,scale=scale,**kw); net.emb_nn.k=k; net.knn_waves=8 if trial%2 else 0
    src,tgt,_,_,_=synth.make_batch(int(rs.randint(0,1000)),B,N)
    s,t=torch.from_numpy(src),torch.from_numpy(tgt)
    rec={}
    cfg=oracle.OracleConfig(k=k,record=rec, **({"vcp_nn":"dist"} if kind=="dist" else {}), **({"pointer":"identity"} if kind=="identity" else {}))
    ref=oracle.vcrnet_forward(w,s,t,cfg)
    with torch.no_grad(): out=net(s.cuda(),t.cuda())
    amb=torch.zeros(B,dtype=torch.bool)
    if N>k+1:
        for side,xyz in (("emb_src",s),("emb_tgt",t)):
            for feat in (rec[side]["x64"],xyz):
                top=torch.topk(oracle.neg_sqdist_knn(feat),k+2,dim=-1).values
                amb|=(top[...,k]==top[...,k+1]).any(1)
    keep=torch.ones_like(amb)
    dR=float((out[2].cpu()-ref[2])[keep].abs().max()) if keep.any() else 0
    dt=float((out[3].cpu()-ref[3])[keep].abs().max()) if keep.any() else 0
    # conditioning of the rigid solve: R moves by ~ |dH| / (s1 + s2) of H's two smallest singular values; a cloud whose
    # soft correspondences collapse onto a line (tiny N, random feature extractor) is one the reference itself does not
    # reproduce between 1 and 8 threads (seed 7, trial 3: 1e-3) -- reported, not counted
    sv=torch.linalg.svdvals(rec["H"].double()); gap=float(((sv[:,1]+sv[:,2])/sv[:,0]).min())
    ok=dR<=1e-4 and dt<=(3e-5 if N<=128 else 1e-5)
    flag="" if ok else ("  (ill-conditioned H: not counted)" if gap<0.15 else "  <<<<<< FAIL")
    if not ok and gap>=0.15:
        # the reference arithmetic's own sensitivity on this input: the oracle against its float64 twin (same weights, same
        # clouds).  The rule: a deviation within the BASELINE tolerance PLUS three times that spread is the input's, not the kernels'.
        cfg64=oracle.OracleConfig(k=k, **({"vcp_nn":"dist"} if kind=="dist" else {}), **({"pointer":"identity"} if kind=="identity" else {}))
        r64=oracle.vcrnet_forward({kk:v.double() for kk,v in w.items()},s.double(),t.double(),cfg64)
        sR=float((r64[2].float()-ref[2]).abs().max()); st=float((r64[3].float()-ref[3]).abs().max())
        if dR<=1e-4+3*sR and dt<=1e-5+3*st: flag=f"  (oracle vs its float64 twin: dR {sR:.2e} dt {st:.2e}; rule applied: BASELINE tolerance + 3 x that spread (dR <= {1e-4+3*sR:.2e}, dt <= {1e-5+3*st:.2e}): not counted)"
        else: flag+=f"  (oracle vs its float64 twin: dR {sR:.2e} dt {st:.2e})"
    print(f"{kind:8s} {regime:8s} wseed={wseed:6d} scale={scale:.2f} B={B} N={N:5d} k={k:2d} amb={int(amb.sum())} dR={dR:.2e} dt={dt:.2e} sv-gap={gap:.3f}{flag}", flush=True)
print("elapsed",time.time()-t0)
