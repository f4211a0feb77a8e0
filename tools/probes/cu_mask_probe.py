#!/usr/bin/env python3
"""Does hipExtStreamCreateWithCUMask confine a stream's kernels to the masked CUs on this stack?  (the placement
step for config 5 needs it.)  A bandwidth-light, CU-bound torch kernel (fp32 matmul) is timed on an unmasked stream and on streams
masked to 224 and to 32 of the 256 CUs; then the 224-CU and the 32-CU stream run TOGETHER.  One JSON line."""
import ctypes
import json
import time

import torch

hip = ctypes.CDLL('libamdhip64.so')
dev = torch.device('cuda:0')
torch.zeros(1, device=dev)
n_cu = torch.cuda.get_device_properties(0).multi_processor_count


def masked_stream(cus):
    words = (n_cu + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for c in cus:
        mask[c // 32] |= 1 << (c % 32)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(words), mask)
    if rc != 0:
        raise RuntimeError(f'hipExtStreamCreateWithCUMask -> {rc}')
    return torch.cuda.ExternalStream(st.value, device=dev)


a = torch.randn(4096, 4096, device=dev)
b = torch.randn(4096, 4096, device=dev)


def run(streams, reps=20):
    for s in streams:
        with torch.cuda.stream(s):
            torch.matmul(a, b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for s in streams:
            with torch.cuda.stream(s):
                torch.matmul(a, b)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


out = {'n_cu': n_cu}
try:
    full = torch.cuda.Stream(dev)
    # CU numbering of the mask: bit i = CU i of the agent's flat list; take whole ranges
    s224 = masked_stream(range(0, 224))
    s32 = masked_stream(range(224, 256))
    out['ms_matmul_unmasked'] = round(run([full]), 3)
    out['ms_matmul_224'] = round(run([s224]), 3)
    out['ms_matmul_32'] = round(run([s32]), 3)
    out['ms_both_together_one_matmul_each'] = round(run([s224, s32]), 3)
except Exception as e:      # noqa: BLE001
    out['error'] = repr(e)
print(json.dumps(out))
