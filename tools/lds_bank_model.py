"""Dev tool: LDS cycles per 16-lane group of the A-operand ds_read_b128 of the split box kernels (csrc/conv3d_split.hip) under a halo-box layout
slot(z, y, x) = z * SZ + y * SY + x, from the instruction's fixed lane groups and the 64 x 4-byte banks (MI355X_MICROARCH.md, LDS section): a group costs
as many cycles as the largest number of DIFFERENT addresses that share a 16-byte slot position mod 256 bytes.  Averages over the 7 k-steps (4 taps each)
and the 4 groups; prints the layouts of rounds 2-3 (10, 100) and the alternatives searched.

    python tools/lds_bank_model.py
"""
import itertools
GROUPS=[list(range(0,4))+list(range(12,16))+list(range(20,28)),
        list(range(4,12))+list(range(16,20))+list(range(28,32)),
        [32+i for i in list(range(0,4))+list(range(12,16))+list(range(20,28))],
        [32+i for i in list(range(4,12))+list(range(16,20))+list(range(28,32))]]
def cost(SY,SZ,vox):   # vox(ri)->(x,y,z) offsets within the m-tile
    tot=0; n=0
    for s in range(7):
        for grp in GROUPS:
            slots={}
            for lane in grp:
                g,ri=lane>>4,lane&15
                tp=min(4*s+g,26)
                dz,dy,dx=tp//9-1,(tp//3)%3-1,tp%3-1
                x,y,z=vox(ri)
                a=(z+dz)*SZ+(y+dy)*SY+(x+dx)
                slots.setdefault(a%16,set()).add(a)
            tot+=max(len(v) for v in slots.values()); n+=1
    return tot/n
maps={'8x2y':lambda ri:(ri&7,ri>>3,0),'8x2z':lambda ri:(ri&7,0,ri>>3),'4x4y':lambda ri:(ri&3,ri>>2,0),'4x2y2z':lambda ri:(ri&3,(ri>>2)&1,ri>>3),
      '4x(2y)interleave':lambda ri:((ri&3)+4*((ri>>3)&1),(ri>>2)&1,0)}
for name,m in maps.items():
    best=[]
    for SY in range(10,17):
        for pad in range(0,17):
            SZ=10*SY+pad
            best.append((cost(SY,SZ,m),SY,SZ))
    best.sort()
    print(name,'current(10,100): %.2f'%cost(10,100,m),'best:',best[:4])
m=maps['8x2y']
print('y-major: y stride 104, z stride 10:', cost(104,10,m), ' (y 120, z 11):', cost(120,11,m), ' (y 104, z 10) for 4^3-like? n/a')
for ys in (104,105,106,107,108,112,120):
    for zs in (10,11,12):
        if ys>=10*zs: print(ys,zs,round(cost(ys,zs,m),3))
