import sys, os
sys.path[:0]=['/root/repo/audio-forge_amd','/root/repo/oracle','/root/repo/tests']
import numpy as np, ctypes as C
import signals as S, af_oracle_py as oracle
import mic_eq_mi
L=oracle.lib()
x=(S.kat_signal(40)*np.float32(3.0)+np.float32(0.2)).astype(np.float32); x[100]=np.nan; x[2000]=np.inf
class Pre(C.Structure): _fields_=[("dc_x1",C.c_float),("dc_y1",C.c_float),("hp",C.c_byte*256)]
pre=Pre(); L.afo_prefilter_init.argtypes=[C.c_void_p,C.c_double]; L.afo_prefilter_process_block.argtypes=[C.c_void_p,C.POINTER(C.c_float),C.c_size_t,C.c_int]
L.afo_sanitize_and_clamp.argtypes=[C.POINTER(C.c_float),C.c_size_t]; L.afo_sanitize_and_clamp.restype=C.c_uint64
L.afo_prefilter_init(C.byref(pre),48000.0); want=x.copy(); fp=want.ctypes.data_as(C.POINTER(C.c_float)); L.afo_sanitize_and_clamp(fp,want.size); L.afo_prefilter_process_block(C.byref(pre),fp,want.size,1)
for var in ("ring-8x4","ring-16x4","ring-16x2","lane"):
    os.environ["AF_KERNEL_VARIANT"]=var
    for rep in range(3):
        eng=mic_eq_mi.Engine(48000.0,1); eng.set_input_clamp_enabled(1); eng.set_prefilter_enabled(1,1); eng.set_limiter_enabled(0); eng.set_eq_enabled(0)
        got=eng.process(x.reshape(1,-1))[0]; eng.close()
        bad=np.nonzero(got!=want)[0]
        print(var, rep, bad.size, bad[:5], (got[bad[:3]], want[bad[:3]]) if bad.size else "")
