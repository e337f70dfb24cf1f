import copy, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from glnn_amd import ops
from glnn_amd.models import Model
from glnn_amd.student import StudentEngine
dev="cuda:0"
KNOBS=("GLNN_GEMM_LAT","GLNN_GEMM_TN_LAT","GLNN_STUDENT_ONE_CALL","GLNN_STUDENT_LAT_BN_BWD")
for dims,norm,p in (([128,256,256,40],"batch",0.2),([100,72,47],"batch",0.5),([130,64,64,7],"none",0.0),([128,1024,1024,40],"batch",0.5)):
    torch.manual_seed(5)
    base=Model(dict(model_name="MLP",num_layers=len(dims)-1,feat_dim=dims[0],hidden_dim=dims[1],label_dim=dims[-1],dropout_ratio=p,norm_type=norm,device=dev))
    x=ops.as_feat(torch.randn(2048,dims[0],device=dev)); t=ops.as_feat(torch.log_softmax(torch.randn(2048,dims[-1],device=dev),1))
    res=[]
    for mode in "01":
        for k in KNOBS: os.environ[k]=mode
        m2=copy.deepcopy(base); m2.train(); opt=torch.optim.Adam(m2.parameters(),lr=0.001)
        eng=StudentEngine(m2,opt,512)
        losses=[]
        for i,m in enumerate([512,300,33,512,1,511,64,512]):
            eng.step(x, torch.arange(i*5,i*5+m,device=dev), ops.LOSS_KL, t, 1.0); losses.append(eng.loss_out.clone())
        torch.cuda.synchronize()
        res.append((torch.stack(losses).flatten(), [v.clone() for v in m2.state_dict().values()], int(eng.sync_counters.abs().sum()) if eng.sync_counters is not None else 0))
    dl=float((res[0][0]-res[1][0]).abs().max())
    dp=max(float((a.float()-b.float()).abs().max()) for a,b in zip(res[0][1],res[1][1]))
    print(dims,norm,"max loss diff",dl,"max state diff",dp,"counters",res[0][2],res[1][2], "losses", [round(float(v),4) for v in res[1][0]])
