import torch, torch.nn.functional as F
B,H,Lq,Lk,d=40,8,100,196,32
q=torch.randn(B,Lq,H*d,device='cuda',requires_grad=True); k=torch.randn(B,Lk,H*d,device='cuda',requires_grad=True); v=torch.randn(B,Lk,H*d,device='cuda',requires_grad=True)
mask=torch.rand(B,1,Lq,Lk,device='cuda')>0.3
qq=q.view(B,Lq,H,d).transpose(1,2); kk=k.view(B,Lk,H,d).transpose(1,2); vv=v.view(B,Lk,H,d).transpose(1,2)
o=F.scaled_dot_product_attention(qq,kk,vv,attn_mask=mask)
print('o', o.shape, o.stride(), o.transpose(1,2).is_contiguous())
g=torch.randn(B,Lq,H*d,device='cuda')
go=g.view(B,Lq,H,d).transpose(1,2)
dq,dk,dv=torch.autograd.grad(o,(qq,kk,vv),go)
print('dq',dq.stride(),'dk',dk.stride(),'dv',dv.stride())
print(dq.transpose(1,2).reshape(B*Lq,H*d).data_ptr()==dq.data_ptr())
