import numpy as np, sys
sys.path.insert(0,'.')
from sdrterm_amd.engine import Engine
from tests.golden import prng
fs,q,n=1024000,64,32768
for center in (15700,0):
  eng=Engine(fs,q,'fm','h',row_freqs=[float(center)],apply_shift=bool(center),omega_out=5000)
  loud=np.frombuffer(prng.raw_bytes('h',n,seed=3,amp=8000),dtype=np.uint8)
  zero=np.zeros(n*4,dtype=np.uint8)
  for name,parts in (('Z',[zero]),('ZZ',[zero,zero]),('LZL',[loud,zero,loud]),('ZL',[zero,loud]),('LLLZ',[loud,loud,loud,zero])):
    raw=np.concatenate(parts).tobytes()
    out,y=eng.process_debug(raw)
    nch=len(parts); m=n//q
    y=y.reshape(1,nch,m); out=out.reshape(1,nch,m)
    zi=[i for i,p in enumerate(parts) if p is zero]
    print(center,name,eng.stage_a,[ (float(np.abs(y[0,i]).max()), float(np.abs(out[0,i]).max())) for i in zi])
    out2=eng.process(raw).reshape(1,nch,m)
    print('   process():',[float(np.abs(out2[0,i]).max()) for i in zi])
