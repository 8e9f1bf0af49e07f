"""CPU statistics (numpy, no GPU): records per tile of the sparse MSDA backward for candidate tile shapes, on the committed in-model
sampling locations and on uniform ones -- how many tiles exceed the 640-record cap (they are cut into shares) and what the halo copies
cost.  python scripts/tile_stats_cpu.py"""
import numpy as np, sys
z=np.load('/root/repo/tests/golden/inmodel_decoder_locations.npz')
loc=z['loc'].astype(np.float32); print(loc.shape, list(z.keys()))
shapes=[(100,167),(50,84),(25,42),(13,21)]
B,Q,M,L,P,_=loc.shape
def stats(loc,name):
  print("==",name)
  for l,(H,W) in enumerate(shapes):
    x=loc[:,:,:,l,:,0]*np.float32(W)-np.float32(0.5); y=loc[:,:,:,l,:,1]*np.float32(H)-np.float32(0.5)
    valid=(y>-1)&(x>-1)&(y<H)&(x<W)
    cy=np.floor(y).astype(int)+1; cx=np.floor(x).astype(int)+1
    for (TH,TW) in [(16,8),(8,8),(8,4),(4,8),(4,4)]:
      nty=(H+TH-1)//TH; ntx=(W+TW-1)//TW
      cnt=np.zeros((B,M,nty,ntx),int)
      hy=np.minimum(cy,H-1); hx=np.minimum(cx,W-1); yA=np.maximum(cy-1,0); xA=np.maximum(cx-1,0)
      tyA,tyB,txA,txB=yA//TH,hy//TH,xA//TW,hx//TW
      bb=np.arange(B)[:,None,None,None]*np.ones_like(cy); mm=np.arange(M)[None,None,:,None]*np.ones_like(cy)
      def add(ty,tx,mask):
        np.add.at(cnt,(bb[mask],mm[mask],ty[mask],tx[mask]),1)
      add(tyB,txB,valid); add(tyB,txA,valid&(txA!=txB)); add(tyA,txB,valid&(tyA!=tyB)); add(tyA,txA,valid&(tyA!=tyB)&(txA!=txB))
      tot=cnt.sum(); nt=cnt.size
      c=cnt.reshape(-1)
      print("L%d %2dx%-2d tiles %5d records %7d (x%.2f of %d valid) max %5d  >640: %4d  >1024: %4d shares@640 %4d  mean %.0f p99 %d"%(l,TH,TW,nt,tot,tot/max(valid.sum(),1),valid.sum(),c.max(),(c>640).sum(),(c>1024).sum(),np.ceil(c[c>640]/640).sum(),c.mean(),np.percentile(c,99)))
stats(loc,"inmodel")
rng=np.random.default_rng(0)
stats(rng.random(loc.shape,dtype=np.float32),"uniform")
