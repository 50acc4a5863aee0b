import random, sys
random.seed(1)
p_n = [(0.653,'n'),(0.314,'l'),(0.033,'s')]
p_l = [(0.63,'n'),(0.30,'l'),(0.07,'s')]
def nxt(tab):
    r=random.random(); a=0
    for p,s in tab:
        a+=p
        if r<a: return s
    return tab[-1][1]
def sim(N, shade_min, ovh, iters=200000, W=64, policy="max"):
    st=['s']*N
    cost={'n':155+ovh,'l':150+ovh,'s':1000+ovh}
    tot=0; lanes={'n':0,'l':0,'s':0}; batches={'n':0,'l':0,'s':0}; rays=0
    for it in range(iters):
        cnt={'n':0,'l':0,'s':0}
        for s in st: cnt[s]+=1
        if cnt['s']>=shade_min or (cnt['n']==0 and cnt['l']==0): S='s'
        else:
            if policy=="max": S='n' if cnt['n']>=cnt['l'] else 'l'
            else:
                # prefer full batches; leaf when >= W else node
                S='l' if cnt['l']>=W or cnt['n']==0 else 'n'
        k=0
        for i in range(N):
            if st[i]==S and k<W:
                k+=1
                if S=='n': st[i]=nxt(p_n)
                elif S=='l': st[i]=nxt(p_l)
                else: st[i]='n'; rays+=1
        tot+=cost[S]; lanes[S]+=k; batches[S]+=1
    return tot/rays, {s: lanes[s]/max(1,batches[s]) for s in lanes}
for N in (64,96,128,160,192,256):
    for sm in (24,32,48):
        for pol in ("max","leaf64"):
            c,f=sim(N,sm,45,iters=40000,policy=pol)
            print("N=%3d shade_min=%2d %-6s wave-instr/ray %.1f  fill n %.1f l %.1f s %.1f" % (N,sm,pol,c,f['n'],f['l'],f['s']))
