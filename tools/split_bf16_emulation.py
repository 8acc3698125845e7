import torch, numpy as np
torch.manual_seed(0)
def split(x):
    hi = x.to(torch.bfloat16).to(torch.float32)
    lo = (x - hi).to(torch.bfloat16).to(torch.float32)
    return hi, lo
def mm_split(x, w):      # x [B,K] f32, w [K,F] f32 ; 3 products, f32 accumulate (emulate with f64 accumulate of exact bf16 products then round once: optimistic) 
    xh, xl = split(x); wh, wl = split(w)
    return (xh.double() @ wh.double() + xh.double() @ wl.double() + xl.double() @ wh.double()).float()
def mm_split_f32acc(x, w):   # f32 accumulation order effects: use float32 matmuls
    xh, xl = split(x); wh, wl = split(w)
    return xh @ wh + (xh @ wl + xl @ wh)
def net(x, W, mm, act):
    w1,b1,w2,b2,w3,b3 = W
    h = torch.relu(mm(x, w1) + b1)
    h = torch.relu(mm(h, w2) + b2)
    return act(mm(h, w3) + b3)
def mm64(x, w): return (x.double() @ w.double())
def mm32(x, w): return x @ w
B = 4096
for name,(d,h1,h2,no),act,scale in [("softmax16",(6,300,300,16),lambda y: torch.softmax(y,-1),0.2),
                               ("gauss",(6,400,400,4),lambda y: torch.cat([torch.tanh(y[...,:2]),torch.sigmoid(y[...,2:])],-1),0.2),
                               ("critic",(6,200,200,1),lambda y:y,0.3),
                               ("critic_big",(6,200,200,1),lambda y:y,1.0)]:
    r = lambda *s: (torch.rand(*s)*2-1)*scale
    W = (r(d,h1), r(h1), r(h1,h2)*(1 if scale<1 else 0.3), r(h2), r(h2,no), r(no))
    x = torch.rand(B,d)*6-3
    ref = net(x.double(), [t.double() for t in W], mm64, act)
    for nm,mm in (("f32",mm32),("split",mm_split),("split_f32acc",mm_split_f32acc)):
        y = net(x, W, mm, act).double()
        err = (y-ref).abs(); tol = 1e-5+1e-5*ref.abs()
        print(f"{name:10s} {nm:13s} max abs err {err.max():.2e}  max err/tol {(err/tol).max():.2f}  |ref| max {ref.abs().max():.2f}")
print("---- 3-way truncation split (hi, mid, lo), 6 products")
def split3(x):
    xi = x.view(torch.int32)
    hi = (xi & -65536).view(torch.float32)
    rem = x - hi
    mid = (rem.view(torch.int32) & -65536).view(torch.float32)
    lo = rem - mid
    assert torch.equal(lo, lo.to(torch.bfloat16).to(torch.float32))
    return hi, mid, lo
def mm_split3(x, w):
    xh, xm, xl = split3(x.contiguous()); wh, wm, wl = split3(w.contiguous())
    return xh @ wh + (xh @ wm + xm @ wh) + (xh @ wl + xl @ wh + xm @ wm)
torch.manual_seed(0)
for name,(d,h1,h2,no),act,scale in [("softmax16",(6,300,300,16),lambda y: torch.softmax(y,-1),0.2),
                               ("gauss",(6,400,400,4),lambda y: torch.cat([torch.tanh(y[...,:2]),torch.sigmoid(y[...,2:])],-1),0.2),
                               ("critic",(6,200,200,1),lambda y:y,0.3),
                               ("critic_big",(6,200,200,1),lambda y:y,1.0)]:
    r = lambda *s: (torch.rand(*s)*2-1)*scale
    W = (r(d,h1), r(h1), r(h1,h2)*(1 if scale<1 else 0.3), r(h2), r(h2,no), r(no))
    x = torch.rand(B,d)*6-3
    ref = net(x.double(), [t.double() for t in W], mm64, act)
    for nm,mm in (("f32",mm32),("split3",mm_split3)):
        y = net(x, W, mm, act).double()
        err = (y-ref).abs(); tol = 1e-5+1e-5*ref.abs()
        print(f"{name:10s} {nm:13s} max abs err {err.max():.2e}  max err/tol {(err/tol).max():.2f}")
print("---- 2-way float16 split (hi = f16(v), lo = f16(v - hi)), 3 products; ftz = subnormal parts flushed to zero")
def split_h(x, ftz=False, rtz_hi=False):
    hi = x.to(torch.float16)
    if ftz: hi = torch.where(hi.abs() < 2.0 ** -14, torch.zeros_like(hi), hi)
    hi = hi.to(torch.float32)
    lo = (x - hi).to(torch.float16)
    if ftz: lo = torch.where(lo.abs() < 2.0 ** -14, torch.zeros_like(lo), lo)
    return hi, lo.to(torch.float32)
def make_mm_h(ftz):
    def mm(x, w):
        xh, xl = split_h(x, ftz); wh, wl = split_h(w, ftz)
        return xh @ wh + (xh @ wl + xl @ wh)
    return mm
torch.manual_seed(0)
for name,(d,h1,h2,no),act,scale in [("softmax16",(6,300,300,16),lambda y: torch.softmax(y,-1),0.2),
                               ("gauss",(6,400,400,4),lambda y: torch.cat([torch.tanh(y[...,:2]),torch.sigmoid(y[...,2:])],-1),0.2),
                               ("critic",(6,200,200,1),lambda y:y,0.3),
                               ("critic_big",(6,200,200,1),lambda y:y,1.0),
                               ("gauss_small_w",(6,400,400,4),lambda y: torch.cat([torch.tanh(y[...,:2]),torch.sigmoid(y[...,2:])],-1),0.02)]:
    r = lambda *s: (torch.rand(*s)*2-1)*scale
    W = (r(d,h1), r(h1), r(h1,h2)*(1 if scale<1 else 0.3), r(h2), r(h2,no), r(no))
    x = torch.rand(B,d)*6-3
    ref = net(x.double(), [t.double() for t in W], mm64, act)
    for nm,mm in (("f32",mm32),("split3",mm_split3),("f16x2",make_mm_h(False)),("f16x2 ftz",make_mm_h(True))):
        y = net(x, W, mm, act).double()
        err = (y-ref).abs(); tol = 1e-5+1e-5*ref.abs()
        print(f"{name:14s} {nm:13s} max abs err {err.max():.2e}  max err/tol {(err/tol).max():.2f}")
