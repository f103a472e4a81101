import os, sys, numpy as np, torch
sys.path.insert(0, '.')
from evoke_amd import ops
from evoke_amd.model_pretrain_finetune import FineTune
from oracle import functional as O, spec as S
from tests.helpers import ARGS, V, load_procedural, load_tokenizer
torch.set_num_threads(16)
model = FineTune(dict(ARGS), load_tokenizer(), 'iu_xray'); load_procedural(model, S.finetune_spec(V)); model.eval(); ops.set_dropout_enabled(False)
g = torch.Generator().manual_seed(21); B, L, Li = 8, 60, 30
images = torch.randn(2*B,3,224,224,generator=g); ids = torch.randint(5,V-2,(B,L),generator=g); ids[:,0]=V-2; masks=torch.ones(B,L,dtype=torch.long)
for i in range(B):
    ln=L-4*i; ids[i,ln-1]=V-1; ids[i,ln:]=0; masks[i,ln:]=0
inc=torch.randint(5,V-2,(B,Li),generator=g); inc[:,0]=1; incm=torch.ones(B,Li,dtype=torch.long)
pids=np.array(['p%d_s%d'%(i%B,i%B) for i in range(2*B)])
store={}
dec=model.text_decoder
h1=dec.model.decoder.norm.register_forward_hook(lambda m,i,o: store.__setitem__('x', i[0].detach().float()))
h2=dec.logit.register_forward_hook(lambda m,i,o: store.__setitem__('logits', o.detach().float()))
with torch.no_grad():
    hip=model(images.cuda(),ids.cuda(),masks.cuda(),pids,inc,incm,mode='train')['all_loss'].item()
    P={k:v.detach().float().cpu() for k,v in model.state_dict().items() if not k.endswith('position_ids')}
    ref=O.finetune_forward_train(P,images,ids,masks,pids,inc,incm)['all_loss'].item()
x=store['x'].cpu()   # bf16-valued input of the final norm
gam=P['text_decoder.model.decoder.norm.gamma']; bet=P['text_decoder.model.decoder.norm.beta']
mean=x.mean(-1,keepdim=True); std=x.std(-1,keepdim=True)
y=gam*(x-mean)/(std+1e-6)+bet
W=P['text_decoder.logit.weight']; b=P['text_decoder.logit.bias']
lg=y@W.t()+b
def nll(lg):
    lp=torch.log_softmax(lg,-1); tgt=torch.zeros(B,L,dtype=torch.long); tgt[:,:L-1]=ids[:,1:]; w=torch.zeros(B,L); w[:,:L-1]=masks[:,1:].float()
    return float(-(lp.gather(2,tgt.unsqueeze(-1)).squeeze(-1)*w).sum()/w.sum())
print('hip %.6f ref %.6f diff %.2e'%(hip,ref,abs(hip-ref)))
print('f32 tail from hip pre-norm x: %.6f diff vs ref %.2e'%(nll(lg), abs(nll(lg)-ref)))
yb=y.to(torch.bfloat16).float(); Wb=W.to(torch.bfloat16).float()
print('bf16(y), bf16(W) logits: %.6f diff %.2e'%(nll(yb@Wb.t()+b), abs(nll(yb@Wb.t()+b)-ref)))
print('hip logits loss %.6f'%nll(store['logits'].cpu()[..., :V+1]))
