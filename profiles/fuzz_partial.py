import torch, numpy as np, sys, time
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import vcrnet_amd
from vcrnet_amd import native, synth, composed
from vcrnet_amd.module import vcrnetIter
from test_hip_forward import build_net
import oracle
rs=np.random.RandomState(int(sys.argv[1]) if len(sys.argv)>1 else 0)
# each trial draws a weight regime, a weight seed and (trained-like regime) a scale factor
for trial in range(int(sys.argv[2]) if len(sys.argv)>2 else 10):
    regime=str(rs.choice(["default","seed4321","trained","trained","randemb"])); wseed=int(rs.randint(0,10**6)); scale=float(rs.uniform(1.5,4.0))
    net,w=build_net(regime=regime,seed=wseed,scale=scale,partial=True, overlap2=synth.OVERLAP2_0575)
    B=int(rs.randint(1,5)); Nfull=int(rs.choice([64,100,256,333,500,1024,1333]))
    src,tgt,_,_,_=synth.make_batch(int(rs.randint(0,1000)),B,Nfull,partial=True)
    s,t=torch.from_numpy(src).cuda(),torch.from_numpy(tgt).cuda()
    N=s.shape[2]
    with torch.no_grad():
        f=net(s,t); c=composed.forward_composed(net,s,t)
        it=vcrnetIter(net,s,t,iter=3)
    K=f[0].shape[2]; same=0
    for b in range(B):
        pf={tuple(x) for x in torch.cat((f[0][b],f[1][b]),0).t().cpu().numpy().round(6).tolist()}
        pc={tuple(x) for x in torch.cat((c[0][b],c[1][b]),0).t().cpu().numpy().round(6).tolist()}
        same+=len(pf&pc)
    ref=oracle.vcrnet_forward(w,s.cpu(),t.cpu(),oracle.OracleConfig(partial=True,overlap2=synth.OVERLAP2_0575))
    okshape = ref[0].shape==f[0].shape
    det=torch.det(it[2]).cpu()
    dRo=float((f[2].cpu()-ref[2]).abs().max())
    bad = (not okshape) or torch.isnan(it[2]).any().item() or (det-1).abs().max()>1e-4 or same<0.85*K*B
    print(f"{regime:8s} wseed={wseed:6d} scale={scale:.2f} B={B} N={N:4d} K={K:3d} same pairs {same}/{K*B} shape_ok={okshape} det={det.numpy().round(5)} dR_vs_oracle={dRo:.1e}{'  <<<< FAIL' if bad else ''}",flush=True)
