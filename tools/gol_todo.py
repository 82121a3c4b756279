#!/usr/bin/env python3
"""Diagnostic: which streams does the wave-per-frame Exp-Golomb decoder leave to the lane-per-frame kernel?  (frad_debug_golomb_decode_wave)"""
import os, sys, ctypes, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frad_python_amd import core, _lib
dev = torch.device("cuda:0")
N, C = 2048, 2
g = torch.Generator(device=dev).manual_seed(1)
dll = _lib.load().dll
for kind, F in (("noise", 1500), ("tonal", 1500)):
    q = (torch.randn((F, N, C), generator=g, device=dev) * 6).round().to(torch.int32)
    if kind == "tonal":
        q = (torch.randn((F, N, C), generator=g, device=dev) * 40).round().to(torch.int32); q[:, 300:, :] = 0
    tq = torch.randint(0, 30, (F, 27, C), generator=g, device=dev, dtype=torch.int32)
    flat, offs = core.p1_golomb_encode_batch(q, tq)
    qo = torch.zeros_like(q); to = torch.zeros_like(tq); todo = torch.full((F,), -1, dtype=torch.int32, device=dev)
    rc = dll.frad_debug_golomb_decode_wave(ctypes.c_void_p(flat.data_ptr()), ctypes.c_void_p(offs.data_ptr()), ctypes.c_int64(F), ctypes.c_int32(N), ctypes.c_int32(C),
                                           ctypes.c_void_p(qo.data_ptr()), ctypes.c_void_p(to.data_ptr()), ctypes.c_void_p(todo.data_ptr()), ctypes.c_int32(0),
                                           ctypes.c_void_p(int(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    t = todo.cpu().numpy()
    print(json.dumps({"kind": kind, "rc": rc, "frames": F, "left_tq": int((t & 1).sum()), "left_q": int(((t >> 1) & 1).sum()),
                      "first_left": [int(i) for i in np.nonzero(t)[0][:8]], "ok_where_taken": bool(torch.equal(qo[todo == 0], q[todo == 0]))}))
