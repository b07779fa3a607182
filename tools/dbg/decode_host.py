import sys, time
sys.path[:0] = [".", "kosmos-x_amd"]
import torch
from kosmosx import _hip
from kosmosx.model import KosmosLanguage
m = KosmosLanguage(vocab_size=32002, dim=2048, _seed=0).eval().cuda()
m.precision = "bf16"
tok = torch.randint(0, 32002, (1, 300)).cuda()
lib = _hip.load()
orig = lib.kx_decoder_decode_step
acc = [0.0, 0]
def timed(*a):
    t0 = time.perf_counter(); r = orig(*a); acc[0] += time.perf_counter() - t0; acc[1] += 1; return r
with torch.no_grad():
    st = {"max_len": 512}
    m(tok[:, :114], incremental_state=st)
    for t in range(114, 130): m(tok[:, :t + 1], incremental_state=st)
    torch.cuda.synchronize()
    lib.kx_decoder_decode_step = timed
    t0 = time.perf_counter()
    for t in range(130, 194): m(tok[:, :t + 1], incremental_state=st)
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
print(f"per step: total {tot/64*1e3:.3f} ms, host issue {host/64*1e3:.3f} ms, inside kx_decoder_decode_step {acc[0]/acc[1]*1e3:.3f} ms")
