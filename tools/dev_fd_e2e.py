import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import precondition_amd as pa
from tests import cpu_backend
dev = torch.device("cuda:0")
np.set_printoptions(precision=4, linewidth=220, suppress=True)
z=np.load(os.path.join(ROOT,'tests/golden/e2e.npz')); idx=json.load(open(os.path.join(ROOT,'tests/golden/e2e_index.json')))
c=[c for c in idx if c['name']=='tree_c_fd_r4'][0]
name,n=c['name'],c['n_params']; kw=dict(c['kwargs']); bs=kw.pop('block_size')
opt=pa.distributed_shampoo(c['lr'],bs,batch_axis_name=None,**kw)
copt=pa.distributed_shampoo(c['lr'],bs,batch_axis_name=None,_backend_for_testing=cpu_backend,**kw)
params=tuple(torch.tensor(z[f"{name}__param{i}"],device=dev) for i in range(n)); st=opt.init(params)
cparams=tuple(torch.tensor(z[f"{name}__param{i}"]) for i in range(n)); cst=copt.init(cparams)
for t in range(c["steps"]):
    grads=tuple(torch.tensor(z[f"{name}__grad{i}_t{t}"],device=dev) for i in range(n))
    upd,st=opt.update(grads,st,params)
    cupd,cst=copt.update(tuple(g.cpu() for g in grads),cst,cparams)
    print('step',t,[float(np.linalg.norm(upd[i].cpu().numpy()-z[f"{name}__upd{i}_t{t}"])/np.linalg.norm(z[f"{name}__upd{i}_t{t}"])) for i in range(n)])
    for i in range(n):
      for j in range(len(st.stats[i].preconditioners)):
        m=st.stats[i].preconditioners[j].cpu().numpy(); r=cst.stats[i].preconditioners[j].numpy()
        sm=st.stats[i].statistics[j].cpu().numpy(); sr=cst.stats[i].statistics[j].numpy()
        bad = not np.allclose(m[:, -2:], r[:, -2:], rtol=5e-3, atol=1e-3*np.abs(r[:, -2:]).max()) if m.shape[0]!=m.shape[1] else np.linalg.norm(m-r)>3e-2*np.linalg.norm(r)
        if bad or not np.allclose(sm, sr, rtol=1e-4, atol=1e-5*np.abs(sr).max()):
          print('  MISMATCH p',i,j,m.shape,'stat diff',np.abs(sm-sr).max(),'inv',m[:4,-2],r[:4,-2],'ct',m[:2,-1],r[:2,-1],'defl',m[-4:,-1],r[-4:,-1],'hz',m[-1,-2],r[-1,-2])
