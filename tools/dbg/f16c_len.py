import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "kosmos-x_amd"); sys.path.insert(0, "tests")
from kosmosx.model import KosmosLanguage
from kosmosx import _hip as H
from helpers import rel_err
m = KosmosLanguage(vocab_size=32002, dim=2048, _seed=3, _perturb=0.05).eval().to("cuda")
g = torch.Generator().manual_seed(0)
for T in (114, 256, 300, 512, 1024, 2046):
    tok = torch.randint(0, 32002, (1, T), generator=g).cuda()
    m.precision = "fp32"; ref = m(tok)
    for tile in (0, 128):
        H.load().kx_set_tuning(1, tile)
        m.precision = "f16c"; out = m(tok)
        m.precision = "bf16x3"; o3 = m(tok)
        print(f"T={T} tile={tile}: f16c vs fp32-mode {rel_err(out, ref):.3e}   bf16x3 {rel_err(o3, ref):.3e}", flush=True)
    H.load().kx_set_tuning(1, 0)
