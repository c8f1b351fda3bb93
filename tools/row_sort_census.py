import numpy as np, sys
sys.path.insert(0,'/root/repo')
from isfusion_amd import synthetic
def down(c, shape, ks=(3,3,3), st=(2,2,2), pd=(1,1,1)):
    # c [n,4] (b,z,y,x); output o exists if any input i with o*st - pd <= i <= o*st - pd + ks - 1
    outs=[]
    oshape=[(shape[d]+2*pd[d]-ks[d])//st[d]+1 for d in range(3)]
    for kz in range(ks[0]):
      for ky in range(ks[1]):
        for kx in range(ks[2]):
            k=(kz,ky,kx)
            o=np.empty_like(c); o[:,0]=c[:,0]; ok=np.ones(len(c),bool)
            for d in range(3):
                num=c[:,d+1]+pd[d]-k[d]
                ok&=(num%st[d]==0)
                od=num//st[d]
                ok&=(od>=0)&(od<oshape[d])
                o[:,d+1]=od
            outs.append(o[ok])
    o=np.unique(np.concatenate(outs),axis=0)
    return o, oshape
def submasks(c, shape):
    # tap masks of a SubM 3x3x3 conv over sorted coords c
    key=lambda a: ((a[:,0]*shape[0]+a[:,1])*shape[1]+a[:,2])*shape[2]+a[:,3]
    k=key(c); order=np.argsort(k); ks=k[order]
    m=np.zeros(len(c),np.uint32); t=0
    for dz in (-1,0,1):
      for dy in (-1,0,1):
        for dx in (-1,0,1):
            n=c.copy(); n[:,1]+=dz; n[:,2]+=dy; n[:,3]+=dx
            ok=(n[:,1]>=0)&(n[:,1]<shape[0])&(n[:,2]>=0)&(n[:,2]<shape[1])&(n[:,3]>=0)&(n[:,3]<shape[2])
            kk=key(n); pos=np.searchsorted(ks,kk); pos[pos>=len(ks)]=len(ks)-1
            has=ok&(ks[pos]==kk)
            m|=(has.astype(np.uint32)<<t); t+=1
    return m
def steps(masks, tile=128):
    n=len(masks); tot=0; grp=0
    for s in range(0,n,tile):
        mm=masks[s:s+tile]
        tot+=bin(int(np.bitwise_or.reduce(mm))).count('1')
    for s in range(0,n,16):
        grp+=bin(int(np.bitwise_or.reduce(masks[s:s+16]))).count('1')
    return tot, grp
B=4
cs=[]
rg=np.array([-54.0,-54.0,-5.0]); vs=np.array([0.075,0.075,0.2])
for b in range(B):
    p=synthetic.lidar_sweeps(1234+2000+b,300000)
    c=np.floor((p[:,:3]-rg)/vs).astype(np.int64)
    ok=((c>=0)&(c<np.array([1440,1440,40]))).all(1)
    c=c[ok][:,::-1]  # z,y,x
    cs.append(np.concatenate([np.full((len(c),1),b),c],1))
c0=np.unique(np.concatenate(cs),axis=0); shape=[41,1440,1440]
c1,s1=down(c0,shape); c2,s2=down(c1,s1); c3,s3=down(c2,s2)
c4,s4=down(c3,s3,(3,1,1),(2,1,1),(0,0,0))
print("levels",len(c0),len(c1),len(c2),len(c3),len(c4),s1,s2,s3,s4)
for name,c,sh in (("L2",c2,s2),("L3",c3,s3),("L4",c4,s4)):
    m=submasks(c,sh)
    pairs=sum(bin(int(x)).count('1') for x in m)
    t0,g0=steps(m)
    # global sort by mask within 4 parts (contiguous quarters), keeping parts
    n=len(m); res=[]
    for parts in (4,1):
        tt=0; gg=0
        for pi in range(parts):
            seg=m[pi*n//parts:(pi+1)*n//parts]
            pc=np.array([bin(int(x)).count('1') for x in seg])
            o=np.lexsort((seg,pc)) if False else np.argsort(seg, kind='stable')
            a,b_=steps(seg[o]); tt+=a; gg+=b_
        res.append((parts,tt,gg))
    print(name,"rows",n,"pairs/row %.1f"%(pairs/n),"tile-taps now",t0,"(%.1f per tile of 27)"%(t0/np.ceil(n/128)),"group-taps now",g0, "| sorted by mask:",res)

print("--- key variants at L3 / L2 (4 parts)")
def evalkey(m, keyf, parts=4):
    n=len(m); tt=gg=0
    for pi in range(parts):
        seg=m[pi*n//parts:(pi+1)*n//parts]
        o=np.argsort(keyf(seg), kind='stable')
        a,b_=steps(seg[o]); tt+=a; gg+=b_
    return tt,gg
def perm_bits(m, order):
    out=np.zeros_like(m, dtype=np.uint64)
    for newpos,old in enumerate(order):
        out|=((m>>np.uint32(old))&1).astype(np.uint64)<<np.uint64(newpos)
    return out
zp=list(range(18,27)); z0=list(range(9,18)); zm=list(range(0,9))
for name,c,sh in (("L3",c3,s3),("L2",c2,s2)):
    m=submasks(c,sh)
    print(name,"raw",evalkey(m,lambda s:s))
    print(name,"[z0 | z- | z+ms]",evalkey(m,lambda s:perm_bits(s,z0+zm+zp)))
    print(name,"[z0 | z+ | z-ms]",evalkey(m,lambda s:perm_bits(s,z0+zp+zm)))
    # coarse class: (any z+, any z-) then popcount then mask
    def coarse(s):
        ap=((s>>18)&0x1ff)!=0; am=(s&0x1ff)!=0
        pc=np.array([bin(int(x)).count('1') for x in s])
        return (ap.astype(np.uint64)<<40)|(am.astype(np.uint64)<<39)|(pc.astype(np.uint64)<<32)|s.astype(np.uint64)
    print(name,"coarse(z+,z-),popc,mask",evalkey(m,coarse))
    # count-based: popcount per z-block
    def blocks(s):
        f=lambda x: np.array([bin(int(v)).count('1') for v in x])
        return (f((s>>18)&0x1ff).astype(np.uint64)<<44)|(f(s&0x1ff).astype(np.uint64)<<40)|s.astype(np.uint64)
    print(name,"popc(z+),popc(z-),mask",evalkey(m,blocks))

print("--- short keys (4 parts L3/L4, 8 parts L2)")
for name,c,sh,parts in (("L3",c3,s3,4),("L2",c2,s2,8),("L4",c4,s4,4)):
    m=submasks(c,sh)
    full=evalkey(m,lambda s:s,parts)
    zz=evalkey(m,lambda s:(((s>>18)&0x1ff).astype(np.uint64)<<9)|(s&0x1ff).astype(np.uint64),parts)
    # 16-bit: z+ 8 bits (drop centre? no: drop corner bit 0) + z- 8 bits
    z16=evalkey(m,lambda s:((((s>>18)&0x1ff)>>1).astype(np.uint64)<<8)|((s&0x1ff)>>1).astype(np.uint64),parts)
    # counts only: popcount z+, popcount z-, popcount z0 (4+4+4 bits)
    f=lambda x: np.array([bin(int(v)).count('1') for v in x]).astype(np.uint64)
    pc=evalkey(m,lambda s:(f((s>>18)&0x1ff)<<8)|(f(s&0x1ff)<<4)|f((s>>9)&0x1ff),parts)
    none=steps(m)
    print(name,"unsorted",none,"full27",full,"z+z- 18b",zz,"16b",z16,"popcounts 12b",pc)

print("--- narrow levels")
for name,c,sh,parts in (("L1",c1,s1,8),("L0",c0,shape,8)):
    m=submasks(c,sh)
    none=steps(m)
    z16=evalkey(m,lambda s:((((s>>18)&0x1ff)>>1).astype(np.uint64)<<8)|((s&0x1ff)>>1).astype(np.uint64),parts)
    full=evalkey(m,lambda s:s,parts)
    pairs=sum(bin(int(x)).count('1') for x in m)
    print(name,"rows",len(m),"pairs/row %.1f"%(pairs/len(m)),"unsorted tile-taps, group-taps",none,"16b",z16,"full",full, "tiles",int(np.ceil(len(m)/128)))

print("--- strided conv outputs (taps = inputs present in the 3x3x3 window at stride 2)")
def stridemasks(cin, shape_in, cout, ks=(3,3,3), st=(2,2,2), pd=(1,1,1)):
    key=lambda a: ((a[:,0]*shape_in[0]+a[:,1])*shape_in[1]+a[:,2])*shape_in[2]+a[:,3]
    kin=np.sort(key(cin))
    m=np.zeros(len(cout),np.uint32); t=0
    for kz in range(ks[0]):
      for ky in range(ks[1]):
        for kx in range(ks[2]):
            n=cout.copy()
            n[:,1]=cout[:,1]*st[0]-pd[0]+kz; n[:,2]=cout[:,2]*st[1]-pd[1]+ky; n[:,3]=cout[:,3]*st[2]-pd[2]+kx
            ok=(n[:,1]>=0)&(n[:,1]<shape_in[0])&(n[:,2]>=0)&(n[:,2]<shape_in[1])&(n[:,3]>=0)&(n[:,3]<shape_in[2])
            kk=key(n); pos=np.searchsorted(kin,kk); pos[pos>=len(kin)]=len(kin)-1
            m|=((ok&(kin[pos]==kk)).astype(np.uint32)<<t); t+=1
    return m
for name,ci,si,co,parts in (("L2->L3 (128->256)",c2,s2,c3,4),("L1->L2 (64->128)",c1,s1,c2,8)):
    m=stridemasks(ci,si,co)
    pairs=sum(bin(int(x)).count('1') for x in m)
    print(name,"rows",len(m),"pairs/row %.1f"%(pairs/len(m)),"unsorted",steps(m),"full sort",evalkey(m,lambda s:s,parts),"tiles",int(np.ceil(len(m)/128)))

print("--- ordering variants at L3 (4 parts): frequency-ordered bits, greedy")
m=submasks(c3,s3)
freq=[int(((m>>np.uint32(t))&1).sum()) for t in range(27)]
order_rare_ms=sorted(range(27),key=lambda t:freq[t])          # rare bits most significant? perm_bits puts order[0] at bit 0 (LSB)
print("freq",freq)
print("common bits LSB.. rare MSB", evalkey(m,lambda s:perm_bits(s,sorted(range(27),key=lambda t:-freq[t]))))
print("rare bits LSB.. common MSB", evalkey(m,lambda s:perm_bits(s,order_rare_ms)))

print("--- one-pass keys (<= 10 bits incl. part bits: 4 parts -> 8 key bits; 8 parts -> 7 key bits)")
def k16(s): return ((((s>>19)&0xff).astype(np.uint64))<<8)|((s>>1)&0xff).astype(np.uint64)
for name,c,sh,parts,kb in (("L3",c3,s3,4,8),("L2",c2,s2,8,7),("L4",c4,s4,4,8)):
    m=submasks(c,sh)
    f=lambda x: np.array([bin(int(v)).count('1') for v in x]).astype(np.uint64)
    # a: 4 bits popcount(z+ 9) | 4 bits popcount(z- 9)
    a=evalkey(m,lambda s:(np.minimum(f((s>>18)&0x1ff),15)<<4)|np.minimum(f(s&0x1ff),15),parts)
    # b: z+ edge pattern: (any of row ky=0, ky=1, ky=2 in z+ ; same z-) 3+3 bits + 2 bits in-plane corner info
    def rows3(x): return (((x&0x7)!=0).astype(np.uint64))|((((x>>3)&0x7)!=0).astype(np.uint64)<<1)|((((x>>6)&0x7)!=0).astype(np.uint64)<<2)
    b=evalkey(m,lambda s:(rows3((s>>18)&0x1ff)<<3)|rows3(s&0x1ff),parts)
    # c: top kb bits of the 16-bit key
    cc=evalkey(m,lambda s:k16(s)>>np.uint64(16-kb),parts)
    # d: centre column taps: z+ (ky=1: 3 bits) z- (ky=1: 3 bits) + z+ centre, ...
    print(name,"unsorted",steps(m),"16b",evalkey(m,k16,parts),"| popc(z+),popc(z-) 8b",a,"| 3 rows z+,z- 6b",b,"| top",kb,"bits of 16b",cc)

print("--- coarse6 + detail")
for name,c,sh,parts in (("L3",c3,s3,4),("L2",c2,s2,8),("L4",c4,s4,4)):
    m=submasks(c,sh)
    co=lambda s:(rows3((s>>18)&0x1ff)<<3)|rows3(s&0x1ff)
    a=evalkey(m,lambda s:(co(s)<<np.uint64(16))|k16(s),parts)
    b=evalkey(m,lambda s:(co(s)<<np.uint64(9))|((s>>9)&0x1ff).astype(np.uint64),parts)   # coarse + in-plane 9 bits
    # coarse by columns too: 3 rows + 3 cols per plane = 12 bits
    def cols3(x): return (((x&0x49)!=0).astype(np.uint64))|(((x&0x92)!=0).astype(np.uint64)<<1)|(((x&0x124)!=0).astype(np.uint64)<<2)
    c12=evalkey(m,lambda s:(rows3((s>>18)&0x1ff)<<9)|(rows3(s&0x1ff)<<6)|(cols3((s>>18)&0x1ff)<<3)|cols3(s&0x1ff),parts)
    print(name,"coarse6|k16",a,"coarse6|inplane9",b,"rows+cols 12b",c12)

print("--- strided with coarse keys")
for name,ci,si,co_,parts in (("L2->L3",c2,s2,c3,4),("L1->L2",c1,s1,c2,8)):
    m=stridemasks(ci,si,co_)
    co=lambda s:(rows3((s>>18)&0x1ff)<<3)|rows3(s&0x1ff)
    co9=lambda s:(rows3((s>>18)&0x1ff)<<6)|(rows3((s>>9)&0x1ff)<<3)|rows3(s&0x1ff)
    print(name,"unsorted",steps(m),"full27",evalkey(m,lambda s:s,parts),"coarse6",evalkey(m,co,parts),"coarse9 (3 planes x 3 rows)",evalkey(m,co9,parts),
          "coarse9|k16", evalkey(m,lambda s:(co9(s)<<np.uint64(16))|k16(s),parts))
