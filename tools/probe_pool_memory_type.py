import ctypes, os, sys, torch
ROOT=os.getcwd()
sys.path[:0]=[ROOT, os.path.join(ROOT,"3d-semantic-segmentation_amd")]
import voxproj_host
L=ctypes.CDLL(os.path.join(ROOT,"tools","libprobe_rows.so"))
L.probe_rows_store.argtypes=[ctypes.c_void_p,ctypes.c_longlong,ctypes.c_longlong,ctypes.c_int,ctypes.c_int,ctypes.c_int,ctypes.c_ulonglong,ctypes.c_void_p,ctypes.c_void_p,ctypes.c_longlong,ctypes.c_int,ctypes.c_void_p]
dev=torch.device("cuda",0)
ROWS=32*548*968
sink=torch.zeros(4,device=dev); st=torch.cuda.current_stream().cuda_stream
dst=torch.zeros(200001*512,device=dev)
W,I=63000,272
def run(pool,mode):
    best=None
    for rep in range(4):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(); rc=L.probe_rows_store(pool.data_ptr(),0,ROWS,2048,W,I,7+rep,sink.data_ptr(),dst.data_ptr(),200001,mode,st); assert rc==0; e1.record(); e1.synchronize()
        t=e0.elapsed_time(e1)
        if rep: best=t if best is None else min(best,t)
    return best
pools=[("default (torch)", torch.empty(ROWS*512,dtype=torch.float32,device=dev).normal_())]
for fl,name in ((3,"uncached"),(1,"fine-grained"),(4,"contiguous")):
    try:
        t,_=voxproj_host.resident_empty((ROWS*512,), torch.float32, dev, fallback=False, flags=fl); t.copy_(pools[0][1]); pools.append((name,t))
    except Exception as e:
        print(name, "failed:", e)
for name,p in pools:
    print(f"{name:18s} read-only {run(p,0):.3f} ms | + one plain row store per wave {run(p,1):.3f} | + write-through store {run(p,9):.3f}", flush=True)
    print(f"{'':18s} streaming read {voxproj_host.stream_read_gbs(p):.0f} GB/s")
