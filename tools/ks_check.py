import torch, sys
sys.path.insert(0,'/root/repo')
import flashattention_kernel_project_amd as fa
g=torch.Generator(device='cuda').manual_seed(3)
worst=0
ALG=int(sys.argv[1]) if len(sys.argv)>1 else 29
for dt in (torch.float16, torch.bfloat16):
    for (bh,n) in ((32,1024),(3,128),(5,256),(2,2048),(7,384),(300,512)):
        for spread in (1.0, 2.5):
            q,k,v=(torch.randn(bh,n,64,generator=g,device='cuda') for _ in range(3))
            q,k,v=(q*spread).to(dt),(k*spread).to(dt),v.to(dt)
            s=(q.float()@k.float().transpose(1,2))*0.125
            want=torch.softmax(s,-1)@v.float()
            got=fa.fa_forward(q,k,v,algo=ALG)
            ref27=fa.fa_forward(q,k,v,algo=27)
            e=float((got-want).abs().max()); e27=float((ref27-want).abs().max())
            worst=max(worst,e)
            print(dt,bh,n,spread,'err29 %.2e err27 %.2e'%(e,e27), 'finite', bool(torch.isfinite(got).all()))
print('worst',worst)
