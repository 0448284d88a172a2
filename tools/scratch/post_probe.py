import os, sys, numpy as np, torch
ROOT='/root/repo'; sys.path.insert(0, ROOT)
import yolo_fastest_amd as yf
dev=torch.device("cuda:0")
res,batch=512,64
io=yf.io_params_for(res)
m=yf.YoloFastest(io).to(dev).eval()
m.load_state_dict(torch.load(os.path.join(ROOT,"yolo-fastest-and-embedded-deployment_amd/assets/weights","yolo_fastest_512x640_epoch27.pth"),map_location=dev))
H,W=io["input_shape"][:2]
m(torch.zeros(1,1,H,W,device=dev))
hl,hs=[],[]
for f in range(batch):
    g=np.random.default_rng(f)
    for (h,w),dst in (((H//16,W//16),hl),((H//32,W//32),hs)):
        t=np.empty((3,8,h,w),np.float32)
        t[:,0:2]=g.normal(0,1,(3,2,h,w)); t[:,2:4]=g.normal(0,0.5,(3,2,h,w)); t[:,4]=g.normal(-1,1.5,(3,h,w)); t[:,5:8]=g.normal(0,2,(3,3,h,w))
        dst.append(t.reshape(24,h,w))
pred=(torch.from_numpy(np.stack(hl)).to(dev),torch.from_numpy(np.stack(hs)).to(dev))
for conf,nms,tag in ((0.5,0.2,"dense default"),(0.9999,0.2,"almost no candidates"),(0.5,1.0,"no suppression"),(0.95,0.2,"~few candidates"),(0.8,0.2,"fewer")):
    post=yf.YOLO_post_process(conf,nms,3,3,io["anchors"],io["input_shape"]).bind(m)
    raw=post.detect_raw(pred,kmax=2048); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): raw=post.detect_raw(pred,kmax=2048)
    e1.record(); torch.cuda.synchronize()
    print(f"{tag:24s} {e0.elapsed_time(e1)/10:.3f} ms  survivors {raw['counts'].float().mean().item():.1f}")
