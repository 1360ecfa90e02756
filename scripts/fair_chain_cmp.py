import numpy as np, sys
a=np.load(sys.argv[1]); b=np.load(sys.argv[2])
for k in a.files:
    x,y=a[k],b[k]
    d=np.abs(x-y)
    if x.ndim==3:
        print(k, [f"{float(d[i].max()):.2e}" for i in range(x.shape[0])])
    else:
        print(k, f"{float(d.max()):.3e}", "max|ref|", f"{float(np.abs(x).max()):.3e}")
