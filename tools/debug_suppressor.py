"""Stage-by-stage comparison of the GPU suppressor against the CPU restatement (development aid)."""
import ctypes as C
import sys

sys.path[:0] = ["oracle", "tests", "audio-forge_amd"]
import numpy as np

import af_oracle_py as o
import mic_eq_mi
import signals as S
from mic_eq_mi import _lib

L = o.lib()


class Dbg(C.Structure):
    _fields_ = [("Ex", C.c_float * 22), ("Ep", C.c_float * 22), ("Exp", C.c_float * 22), ("features", C.c_float * 42),
                ("gains", C.c_float * 22), ("X", C.c_float * 962), ("P", C.c_float * 962), ("pitch_index", C.c_int),
                ("silence", C.c_int), ("pitch_gain", C.c_float)]


n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 6
x = S.kat_signal(n_frames)
# oracle frame by frame (raw protocol)
dbg = Dbg.in_dll(L, "afo_rnn_last")
state = C.create_string_buffer(1 << 18)
L.afo_rnn_state_init.argtypes = [C.c_void_p]
L.afo_rnn_weights_synthetic.argtypes = [C.c_void_p, C.c_uint64]
L.afo_rnn_process_frame.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
L.afo_rnn_process_frame.restype = C.c_float
wts = C.create_string_buffer(1 << 17)
L.afo_rnn_weights_synthetic(wts, C.c_uint64(0x5EED))
L.afo_rnn_state_init(state)
ref = []
out_ref = np.zeros_like(x)
for f in range(n_frames):
    fin = (np.clip(x[f * 480:(f + 1) * 480], -1, 1) * np.float32(32768.0)).astype(np.float32)
    fout = np.zeros(480, dtype=np.float32)
    L.afo_rnn_process_frame(wts, state, fout.ctypes.data_as(C.POINTER(C.c_float)), fin.ctypes.data_as(C.POINTER(C.c_float)))
    out_ref[f * 480:(f + 1) * 480] = fout / np.float32(32768.0)
    ref.append({k: np.array(getattr(dbg, k)) if hasattr(getattr(dbg, k), "__len__") else getattr(dbg, k)
                for k, _ in Dbg._fields_})

eng = mic_eq_mi.Engine(48000.0, 1)
eng.set_eq_enabled(0); eng.set_limiter_enabled(0); eng.set_suppressor_enabled(1); eng.suppressor_set_raw_protocol(1)
eng.set_control_block_samples(480)
got = eng.process(x.reshape(1, -1))[0]
rec = np.zeros(160, dtype=np.float32); X = np.zeros(962, dtype=np.float32); P = np.zeros(962, dtype=np.float32)
fp = C.POINTER(C.c_float)
for f in range(n_frames):
    _lib.check(eng._lib.af_suppressor_debug_read(eng._h, f, 0, rec.ctypes.data_as(fp), X.ctypes.data_as(fp), P.ctypes.data_as(fp)))
    r = ref[f]
    Ex, Ep, Exp, feat = rec[0:22], rec[22:44], rec[44:66], rec[66:108]
    graw, g = rec[110:132], rec[132:154]
    sil, pitch = rec[154:156].view(np.int32)
    rel = lambda a, b: float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))
    print(f"frame {f}: X {rel(X, r['X']):.2e} Ex {rel(Ex, r['Ex']):.2e} pitch {pitch}/{r['pitch_index']} sil {sil}/{r['silence']} "
          f"P {rel(P, r['P']):.2e} Ep {rel(Ep, r['Ep']):.2e} Exp {rel(Exp, r['Exp']):.2e} feat {float(np.max(np.abs(feat - r['features']))):.2e} "
          f"g {float(np.max(np.abs(g - r['gains']))):.2e} out {float(np.max(np.abs(got[f*480:(f+1)*480] - out_ref[f*480:(f+1)*480])) ):.2e}")
eng.close()
