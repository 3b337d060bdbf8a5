import numpy as np
a=np.load('/root/repo/gpurun_out/dbg_a.npz'); b=np.load('/root/repo/gpurun_out/dbg_b.npz')
for k in a.files:
    d=np.abs(a[k]-b[k]); bad=np.argwhere(d>1e-3)
    print(k, 'max', d.max(), 'nbad', len(bad), 'of', d.size)
    if len(bad):
        import collections
        for ax,name in enumerate('NDHWC'):
            print('  ',name, sorted(collections.Counter(bad[:,ax]).items())[:24])
