"""Periods of 256 / 512 / 1024 frames, batches and single periods mixed, per-call error against oracle.RefCompat: the probe that located the
Q8 regime's late start for longer calls (round 3)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"tests"))
import oracle as oracle_mod
from cuda_audio_amd.engine import Convolution
from cuda_audio_amd.synth import make_input, make_ir
from helpers import BASE, apply_params
for period, pd, seq in [(1024, 256, [5,1,5,1,1,4]), (512, 256, [5,1,5,1,1,4]), (1024, 0, [5,1,5]), (1024, 256, [1,1,5,5]), (1024, 256, [5,5,1,1]), (256, 256, [5,1,5,1,1,4])]:
    n_ref=4096; pm=period//256
    ncalls=sum(seq)
    x=make_input(ncalls*period, seed=3)
    irs=[make_ir(3072, seed=11, norm=0.05), make_ir(2000, seed=12, norm=0.05)]
    p0,p1=dict(BASE, predelay=pd), dict(BASE, select=1, level=0.9)
    ref=oracle_mod.RefCompat(n_ref, True)
    c=Convolution("mix", n_ref, max_batch=8*pm, period=period, stream_threshold=8)
    for i,ir in enumerate(irs): ref.prepare(i,ir); c.prepare(i,ir)
    apply_params(ref,p0,p1,True); apply_params(c,p0,p1,False)
    q=0; out=[]
    for n in seq:
        s=slice(q*period,(q+n)*period)
        want=ref.process(x[0,s],x[1,s],block=period)
        if n==1: got=np.stack(c.onProcess(x[0,s],x[1,s]))
        else: got=c.process(x[0,s],x[1,s])
        per=[float(np.sqrt(np.mean((got[:,k*period:(k+1)*period]-want[:,k*period:(k+1)*period])**2))) for k in range(n)]
        out.append((n, ["%.1e"%e for e in per]))
        q+=n
    c.close()
    print(period, pd, seq, out)
