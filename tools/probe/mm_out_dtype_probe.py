import torch
dev='cuda:0'
torch.manual_seed(0)
for (m,k,n) in ((128,1536,3072),(128,1536,1536),(128,512,1536),(128,1536,512)):
    a=torch.randn(m,k,device=dev).bfloat16(); w=torch.randn(n,k,device=dev).bfloat16()
    ref=a.float()@w.float().t()
    y=torch.mm(a,w.t(),out_dtype=torch.float32)
    print('mm NT',m,k,n, float((y-ref).norm()/ref.norm()))
    dy=torch.randn(m,n,device=dev).bfloat16()
    ref2=dy.float()@w.float(); y2=torch.mm(dy,w,out_dtype=torch.float32)
    print('mm NN', float((y2-ref2).norm()/ref2.norm()))
    ref3=dy.float().t()@a.float(); y3=torch.mm(dy.t(),a,out_dtype=torch.float32)
    print('mm TN', float((y3-ref3).norm()/ref3.norm()))
a=torch.randn(16,32,384,device=dev).bfloat16(); b=torch.randn(16,384,32,device=dev).bfloat16()
print('bmm', float((torch.bmm(a,b,out_dtype=torch.float32)-torch.bmm(a.float(),b.float())).norm()/torch.bmm(a.float(),b.float()).norm()))
