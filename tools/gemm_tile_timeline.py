"""Per-tile timeline of the Diffuse GEMM from a SC_GEMM_CLOCK=<file> dump (n = 8192):
   SC_GEMM_CLOCK=tiles.txt python tests/probes/diffuse_only.py 8192 2 random
   python tools/gemm_tile_timeline.py tiles.txt
Prints cycles / effective clock per XCD and per generation of tiles."""
import numpy as np, sys
def tilemap(nt):
    out=[]
    npch=(nt+7)//8
    for pi in range(npch):
        for pj in range(pi,npch):
            for ti in range(pi*8,min(nt,pi*8+8)):
                for tj in range(max(ti,pj*8),min(nt,pj*8+8)):
                    out.append((ti,tj))
    return out
tm=tilemap(64)
for name in sys.argv[1:]:
    d=np.loadtxt(name)
    b=d[:,0].astype(int); cyc=d[:,1]; ticks=d[:,2]; start=d[:,3]
    full=len(b); chunk=full//8
    tile=(b&7)*chunk+(b>>3)
    ti=np.array([tm[t][0] for t in tile]); tj=np.array([tm[t][1] for t in tile])
    start=(start-start.min())/100.0; dur=ticks/100.0; clk=cyc/ticks*100
    print(name, 'cycles/tile mean %.3fM min %.3fM max %.3fM' % (cyc.mean()/1e6,cyc.min()/1e6,cyc.max()/1e6))
    print(' dur us mean %.0f min %.0f max %.0f; per-tile clock MHz mean %.0f min %.0f max %.0f' % (dur.mean(),dur.min(),dur.max(),clk.mean(),clk.min(),clk.max()))
    xcd=b&7
    for x in range(8):
        m=xcd==x
        print('  XCD %d: cycles mean %.3fM, clock mean %.0f (min %.0f max %.0f), last end %.0f us' % (x,cyc[m].mean()/1e6,clk[m].mean(),clk[m].min(),clk[m].max(),(start[m]+dur[m]).max()))
    # generation (by start order within xcd)
    m=xcd==0
    order=np.argsort(start[m]); 
    c=cyc[m][order]; k=clk[m][order]; s0=start[m][order]; du=dur[m][order]
    for g in range(4):
        sl=slice(64*g,64*g+64)
        print('  XCD0 gen %d: cycles mean %.3fM (min %.3f max %.3f), clock mean %.0f, dur %.0f..%.0f' % (g,c[sl].mean()/1e6,c[sl].min()/1e6,c[sl].max()/1e6,k[sl].mean(),du[sl].min(),du[sl].max()))
    # diag vs offdiag, correlate cycles with position in patch
    diag=ti==tj
    print('  diag tiles cycles %.3fM, off-diag %.3fM' % (cyc[diag].mean()/1e6,cyc[~diag].mean()/1e6))
