"""Bank-conflict model of the register FFT engine's LDS exchanges (csrc/fft_reg.hpp).

gfx950 LDS as measured (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, profiles/pmc_r02.json): 32 banks of 4 bytes, one
clock serves 32 lanes of a b32 access or 16 lanes of a b64 access; lanes of a group that share a bank (pair)
serialise.  For every stage layout of a plan the script prints the average clocks per 16-lane group of the
ds_read_b64 / ds_write_b64 of that layout (1.0 = conflict-free) for the two last-stage thread mappings of
RegFft::pos (PDEOPT_FFT_LAST_IDENTITY).  A Strang row pass (dit + dif: layouts 2 1 1 0 0 1 1 2 at N = 512) comes
out at 16 clocks per 8 ideal = 0.50 conflict share with the frequency-major mapping -- the measured figure --
and 10 with the thread-major one.
usage: python tools/lds_bank_model.py"""
import collections
def plan(N):
    l=N.bit_length()-1
    if N==1024: return [16,8,8]
    n8=l//3; tail=1<<(l%3)
    return [8]*n8+([tail] if tail>1 else [])
def run(N, identity, PTS=None):
    radix=plan(N); PTS=PTS or (16 if N>512 else 8); TT=N//PTS; NP=N+N//8+1
    def sublen(i):
        ns=N
        for k in range(i): ns//=radix[k]
        return ns
    def pos_of(k):
        p=0;w=1
        for i,R in enumerate(radix):
            S=sublen(i)//R; p+=((k//w)%R)*S; w*=R
        return p
    def pos(stage,j,slot):
        R=radix[stage]; Ns=sublen(stage); S=Ns//R
        u=slot//R; m=slot%R
        if S==1: return ((j+u*TT)*R+m) if identity else pos_of(j+u*TT)+m
        bf=j+u*TT; blk=bf//S; jj=bf%S
        return blk*Ns+jj+m*S
    pad=lambda p: p+(p>>3)
    out=[]
    for st in range(len(radix)):
        tot=ideal=0
        for slot in range(PTS):
            addrs=[]
            for lane in range(64):
                f,j=divmod(lane,TT) if TT<64 else (0,lane)
                addrs.append(f*NP+pad(pos(st,j,slot)))
            for g in range(0,64,16):
                c=collections.Counter(a%16 for a in addrs[g:g+16]); tot+=max(c.values()); ideal+=1
        out.append(tot/ideal)
    return out
for N in (64,128,256,512,1024):
    print(N, plan(N), "current", run(N,False), "identity", run(N,True))
