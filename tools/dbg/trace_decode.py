import sys, ctypes
sys.path[:0] = [".", "kosmos-x_amd"]
import torch
from kosmosx import _hip
from kosmosx.model import KosmosLanguage
m = KosmosLanguage(vocab_size=32002, dim=2048, _seed=0).eval().cuda()
m.precision = "bf16"
tok = torch.randint(0, 32002, (1, 300)).cuda()
lib = _hip.load()
buf = torch.zeros(64 * 4096, dtype=torch.int64, device="cuda")
with torch.no_grad():
    st = {"max_len": 512}
    m(tok[:, :114], incremental_state=st)
    for t in range(114, 140): m(tok[:, :t + 1], incremental_state=st)
    torch.cuda.synchronize()
    lib.kx_debug_trace.argtypes = [ctypes.c_void_p]
    lib.kx_debug_trace(buf.data_ptr())
    for t in range(140, 143): m(tok[:, :t + 1], incremental_state=st)
    torch.cuda.synchronize()
    lib.kx_debug_trace(None)
b = buf.cpu().view(-1, 64)[:, :48].reshape(-1, 3, 2, 8)
names = ["qkv", "out", "fc1", "fc2"]
# third step: launches 2*97 .. 3*97
base = 2 * 97
prev_end = None
for li in range(2, 5):       # layers 2..4
    for ki in range(4):
        r = b[base + li * 4 + ki]
        t0 = int(r[:, :, 0][r[:, :, 0] > 0].min())
        line = f"L{li} {names[ki]:4s}"
        if prev_end is not None: line += f" gap_from_prev_end {(t0 - prev_end) * 10:6d} ns |"
        for wg in range(3):
            for wv in range(2):
                ts = [int(x) for x in r[wg, wv, :6]]
                line += f" wg{wg}w{wv}:" + ",".join(f"{(x - t0) * 10 if x else -1:5d}" for x in ts)
        print(line)
        prev_end = int(r[:, :, 5].max())
