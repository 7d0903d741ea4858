# LDS cycles of the linear-head kernel's access patterns as a function of the pitches
G128 = [[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
G128 += [[l+32 for l in g] for g in G128]
def cyc_b128(addr):      # addr(lane) in dwords; banks (a)%64, 4 wide
    tot=0
    for g in G128:
        use={}
        for l in g:
            a=addr(l)
            for d in range(4):
                use.setdefault((a+d)%64,set()).add(a+d)
        tot+=max(len(v) for v in use.values())
    return tot
def cyc_b32(addr, nb=32):
    tot=0
    for g in (range(0,32),range(32,64)):
        use={}
        for l in g:
            a=addr(l); use.setdefault(a%nb,set()).add(a)
        tot+=max(len(v) for v in use.values())
    return tot
def rowpos(b): return (b&~15)+((b&3)<<2)+((b>>2)&3)
def report(FP,WP,NT=2):
    fa=cyc_b128(lambda l:(l&15)*WP+4*(l>>4))
    ftw=cyc_b32(lambda l:(4*(l>>4))*FP+rowpos(l&15))
    QN=4*NT
    xsw=max(cyc_b32(lambda l:(((wv*64+l)%QN)*4)*FP+rowpos((wv*64+l)//QN)) for wv in range(8))
    dwr=cyc_b128(lambda l:(l&15)*FP+4*(l>>4))
    adr=cyc_b32(lambda l:(4*(l>>4))*WP+(l&15))
    print("FP=%d WP=%d: fwd A b128 %d (4 ideal) | Ft^T write b32 %d (2) | dO^T write b32 %d (2) | dW operand b128 %d (4) | Adam W b32 %d (2)"%(FP,WP,fa,ftw,xsw,dwr,adr))
for FP in (112,116,120,124,128,132,136):
    report(FP,196)
for WP in (196,200,204,208,212,216,220,224,228):
    report(116,WP)
