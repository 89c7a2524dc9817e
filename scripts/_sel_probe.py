import sys; sys.path.insert(0,'/root/repo')
import numpy as np, qingdai_amd as qa
g=qa.SphericalGrid(721,1440); d=g._ops()
r=np.random.default_rng(0); x=np.exp(r.normal(-12,3,(721,1440)))
for _ in range(3): print(d.op_median_positive(x,1e-6), float(np.median(x)))
