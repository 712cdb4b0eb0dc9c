import sys, time; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from pyaudiorestoration_amd import pipeline, _dev, _lib, fourier
rng=np.random.default_rng(4)
n1,sr,n_fft,hop,tiles=322531,44100,512,32,256
marks=[]
for k in range(tiles):
    for t in np.sort(rng.uniform(0.2, n1/sr-0.2, 32)):
        w=rng.uniform(0.004,0.02)
        marks.append((k*n1/sr+t-w/2,500.0,k*n1/sr+t+w/2,9000.0,0.5))
geo=np.array([pipeline.marker_geometry(m,sr,hop,n_fft) for m in marks],dtype=np.int64)
n=n1*tiles
x=torch.randn(n,1,device='cuda'); out=torch.empty_like(x)
plan=pipeline.heal_segments(geo,(n+256)//hop+1,n+256,n,n_fft,hop)
g=None
for rep in range(3):
    g,_=pipeline.heal_dropouts_dev(x,n,1,0,geo,n_fft,hop,out,0,True,g,plan)
torch.cuda.synchronize()
import cProfile, pstats
pr=cProfile.Profile(); pr.enable()
for rep in range(5):
    g,_=pipeline.heal_dropouts_dev(x,n,1,0,geo,n_fft,hop,out,0,True,g,plan)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
