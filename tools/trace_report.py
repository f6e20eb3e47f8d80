import numpy as np, sys
U0 = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
buf=np.load('gpurun_out/trace.npy')
for b in range(4):
    t=buf[b].astype(np.int64)
    print("--- slot",b)
    per=(t[:,1:,0]-t[:,:-1,0])&0xffffffff
    print("period median per wave", np.median(per,axis=1))
    kb=(np.arange(64)+U0)%3
    def seg(name,a,b_,sel=None,nxt=False):
        if nxt: d=(t[:,1:,b_]-t[:,:-1,a])&0xffffffff
        else: d=(t[:,:,b_]-t[:,:,a])&0xffffffff
        if sel is not None: d=d[:,sel[:d.shape[1]]]
        print(f"  {name:28s} w0-3 {d[:4].mean():7.0f}  w4-7 {d[4:].mean():7.0f}  median {np.median(d):7.0f} p90 {np.percentile(d,90):7.0f}")
    seg("first half",0,1); seg("wait vm",1,3); seg("barrier",3,4); seg("second half",4,5); seg("issue",5,6)
    seg("filter slot kb<2",6,2,kb!=2); seg("filter slot kb=2",6,2,kb==2); seg("-> next top",2,0,nxt=True)
    if b == 3:
        # timeline of one unit, relative to barrier release (stamp 4 min over waves)
        for uu in (10, 11, 12):
            base = t[:, uu, 4].min()
            print("  unit", uu, "kb", kb[uu]); 
            for w in range(8): print("    w%d" % w, " ".join(f"{int(t[w,uu,k]-base):6d}" for k in (0,1,3,4,5,6,2)))
