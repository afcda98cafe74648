import ctypes, os, sys, torch
ROOT="/root/repo" if os.path.exists("/root/repo/tools") else os.getcwd()
L=ctypes.CDLL(os.path.join(ROOT,"tools","libprobe_rows.so"))
L.probe_rows_policy.argtypes=[ctypes.c_void_p,ctypes.c_longlong,ctypes.c_int,ctypes.c_int,ctypes.c_ulonglong,ctypes.c_void_p,ctypes.c_int,ctypes.c_void_p]
L.probe_rows.argtypes=[ctypes.c_void_p,ctypes.c_longlong,ctypes.c_longlong,ctypes.c_int,ctypes.c_int,ctypes.c_int,ctypes.c_ulonglong,ctypes.c_void_p,ctypes.c_void_p]
dev=torch.device("cuda",0)
ROWS=32*548*968
pool=torch.empty(ROWS*512,dtype=torch.float32,device=dev).normal_()
sink=torch.zeros(4,device=dev); st=torch.cuda.current_stream().cuda_stream
W,I=63000,272
def t(f):
    best=None
    for rep in range(4):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(); rc=f(rep); assert rc==0,rc; e1.record(); e1.synchronize()
        if rep: best=min(best,e0.elapsed_time(e1)) if best else e0.elapsed_time(e1)
    return best
print("global_load nt (the gather's):", round(t(lambda r: L.probe_rows(pool.data_ptr(),0,ROWS,2048,W,I,5+r,sink.data_ptr(),st)),3))
names={0:"plain",1:"sc0",2:"nt",3:"sc0 nt",16:"sc1",17:"sc0 sc1",18:"sc1 nt",19:"sc0 sc1 nt"}
for aux in (0,2,1,16,18,17,3,19):
    print(f"buffer_load {names[aux]:10s}:", round(t(lambda r: L.probe_rows_policy(pool.data_ptr(),ROWS,W,I,5+r,sink.data_ptr(),aux,st)),3), flush=True)
