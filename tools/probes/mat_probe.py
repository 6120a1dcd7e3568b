import time, numpy as np, torch
from PIL import Image
dev="cuda"
fr = (torch.rand(17,512,512,3,device=dev)*255).to(torch.uint8)
pinned = torch.empty(fr.shape, dtype=torch.uint8, pin_memory=True)
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter()-t0)/n*1e3
def d2h_pinned():
    pinned.copy_(fr, non_blocking=True); torch.cuda.current_stream().synchronize()
print("D2H into pinned           %.2f ms" % t(d2h_pinned))
print("D2H .cpu() pageable        %.2f ms" % t(lambda: fr.cpu()))
hp = pinned.numpy(); hc = fr.cpu().numpy()
print("fromarray x17 from pinned  %.2f ms" % t(lambda: [Image.fromarray(a,"RGB").im for a in hp]))
print("fromarray x17 from pageable %.2f ms" % t(lambda: [Image.fromarray(a,"RGB").im for a in hc]))
print("numpy copy of pinned       %.2f ms" % t(lambda: hp.copy()))
print("numpy copy of pageable     %.2f ms" % t(lambda: hc.copy()))
big = np.empty((17,512,512,4),dtype=np.uint8)
def rgbx():
    big[...,:3] = hc
print("RGB->RGBX numpy            %.2f ms" % t(rgbx))
print("frombuffer RGBX zero-copy x17 %.2f ms" % t(lambda: [Image.frombuffer("RGBX",(512,512),memoryview(a),"raw","RGBX",0,1) for a in big]))
import os; print("cpus", os.cpu_count())
