import sys; sys.path.insert(0,'.')
from chunkyclplugin_amd.renderer import RendererInstance
i=RendererInstance.get(0)
for curve in (0,2):
    w=0
    for part in range(4):
        b,x=i.selftest_gamma_scan(curve, part<<30, 1<<30); w=max(w,x)
        print(curve, part, b, x)
    print('curve',curve,'worst',w, 'x1024=',w*1024)
