"""GPU: the device's stream priority range and the priorities of torch's default stream and of the
library's side stream."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from speech2text_amd import _native as N
hip = ctypes.CDLL("libamdhip64.so")
lo, hi = ctypes.c_int(), ctypes.c_int()
print("hipDeviceGetStreamPriorityRange rc", hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi)),
      "least", lo.value, "greatest", hi.value)
torch.zeros(1, device="cuda")
p = ctypes.c_int()
cur = torch.cuda.current_stream().cuda_stream
print("torch current stream", cur, "priority rc", hip.hipStreamGetPriority(ctypes.c_void_p(cur), ctypes.byref(p)), p.value)
side = N.lib().s2t_side_stream()
print("side stream priority rc", hip.hipStreamGetPriority(ctypes.c_void_p(side), ctypes.byref(p)), p.value)
s_hi = torch.cuda.Stream(priority=-1)
print("torch Stream(priority=-1) rc", hip.hipStreamGetPriority(ctypes.c_void_p(s_hi.cuda_stream), ctypes.byref(p)), p.value)
